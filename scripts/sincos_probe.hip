// Probe (gfx950): absolute error of v_sin_f32 / v_cos_f32 (argument in revolutions) against float64 sin / cos of pi * z for the
// keypoint encoding's argument range, next to the Cody-Waite + minimax form the kernels use (kpn_sincos, kpn_device.h); also the
// doubled angles (2 s c, 1 - 2 s^2, twice), which amplify the error of the base pair.
//   hipcc --offload-arch=gfx950 -O3 scripts/sincos_probe.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__device__ inline void poly_sincos(float y, float& s, float& c) {
    const float k = rintf(y * 0.636619772367581343f);
    float r = fmaf(k, -1.5703125f, y);
    r = fmaf(k, -4.837512969970703125e-4f, r);
    r = fmaf(k, -7.54978995489188216e-8f, r);
    const float r2 = r * r;
    const float sp = fmaf(r2 * r, fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float cp = fmaf(r2 * r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), fmaf(r2, -0.5f, 1.0f));
    const int q = (int)k;
    const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}
__global__ void k(int n, float lo, float hi, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float z = lo + (hi - lo) * ((float)i + 0.5f) / (float)n;
    float hs, hc;
    const float rev = z * 0.5f;
    asm volatile("v_sin_f32 %0, %2\n\tv_cos_f32 %1, %2\n\ts_nop 1" : "=&v"(hs), "=&v"(hc) : "v"(rev));
    float ps, pc;
    poly_sincos(z * 3.14159274101257324f, ps, pc);
    out[i * 5 + 0] = z; out[i * 5 + 1] = hs; out[i * 5 + 2] = hc; out[i * 5 + 3] = ps; out[i * 5 + 4] = pc;
}
int main() {
    const int n = 1 << 22;
    float* d; hipMalloc(&d, (size_t)n * 5 * 4);
    for (float range : {0.5f, 2.0f, 8.0f}) {
        k<<<n / 256, 256>>>(n, -range, range, d);
        std::vector<float> h((size_t)n * 5);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        double e[2][3] = {{0, 0, 0}, {0, 0, 0}};   // [hardware, polynomial][frequency 1, 2, 4]: max over sin and cos
        for (int i = 0; i < n; ++i) {
            const double z = h[(size_t)i * 5];
            for (int m = 0; m < 2; ++m) {
                float s = h[(size_t)i * 5 + 1 + 2 * m], c = h[(size_t)i * 5 + 2 + 2 * m];
                for (int f = 0; f < 3; ++f) {
                    const double a = M_PI * z * (1 << f);
                    e[m][f] = fmax(e[m][f], fmax(fabs(s - sin(a)), fabs(c - cos(a))));
                    const float s2 = 2.0f * s * c, c2 = 1.0f - 2.0f * s * s;
                    s = s2; c = c2;
                }
            }
        }
        printf("z in [-%g, %g]: max abs error of (sin, cos)(pi z), (2 pi z), (4 pi z):  v_sin/v_cos_f32 %.3g %.3g %.3g   polynomial %.3g %.3g %.3g\n",
               range, range, e[0][0], e[0][1], e[0][2], e[1][0], e[1][1], e[1][2]);
    }
    return 0;
}
