// Feasibility benchmark for a split-bf16 k_geo_rows: a chain of 128 -> 128 softplus layers in the
// transposed, register-chained formulation, weights streamed from global memory (L2) as three pre-split bf16 pieces,
// activations split on the fly (hi/mid/lo), 6 products per 16-deep K-step on v_mfma_f32_32x32x16_bf16.
// Reports fp32-equivalent TFLOP/s (2*128*128 flop per point per layer) next to the fp32-MFMA kernel's 117.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) f32x4* gptr4;

__device__ __forceinline__ float softplus100(float x) {
    const float sp = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(x * 144.269504088896341f)) * 6.93147180559945309e-3f;
    return (x * 100.0f > 20.0f) ? x : sp;
}
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 a = (__bf16)x[i];
        const float r1 = x[i] - (float)a;
        const __bf16 b = (__bf16)r1;
        const float r2 = r1 - (float)b;
        h[i] = a; m[i] = b; l[i] = (__bf16)r2;
    }
}
// weights: [layer][kstep 8][piece 3][ob 4][lane 64] x 16 B
template <int NPROD>
__global__ __launch_bounds__(256, 2) void k_layers(const float* __restrict__ w, int layers, int tiles_per_wave, float* out) {
    const int lane = threadIdx.x & 63;
    f32x16 cur[4];
    for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) cur[b][r] = 0.01f * (lane + r + b);
    for (int t = 0; t < tiles_per_wave; ++t)
        for (int L = 0; L < layers; ++L) {
            f32x16 acc[4];
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][r] = 0.001f;
            const float* wl = w + (size_t)L * 8 * 3 * 4 * 64 * 4;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                float x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = softplus100(cur[ks / 2][(ks % 2) * 8 + i]);
                bf16x8 xh, xm, xl;
                split3(x, xh, xm, xl);
                const float* gp = wl + (size_t)ks * 3 * 4 * 64 * 4;
                asm volatile("" : "+s"(gp));
                bf16x8 wh[4], wm[4], wlo[4];
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    const f32x4 a = ((gptr4)gp)[(0 * 4 + ob) * 64 + lane];
                    const f32x4 b = ((gptr4)gp)[(1 * 4 + ob) * 64 + lane];
                    const f32x4 c = ((gptr4)gp)[(2 * 4 + ob) * 64 + lane];
                    wh[ob] = __builtin_bit_cast(bf16x8, a); wm[ob] = __builtin_bit_cast(bf16x8, b); wlo[ob] = __builtin_bit_cast(bf16x8, c);
                }
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ob], xh, acc[ob], 0, 0, 0);
                    acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ob], xm, acc[ob], 0, 0, 0);
                    acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[ob], xh, acc[ob], 0, 0, 0);
                    if constexpr (NPROD >= 6) {
                        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[ob], xm, acc[ob], 0, 0, 0);
                        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ob], xl, acc[ob], 0, 0, 0);
                        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[ob], xh, acc[ob], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) cur[b] = acc[b];
        }
    float s = 0;
    for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) s += cur[b][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NPROD>
void run(const float* w, float* d, int layers) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 512, tiles = 400;
    k_layers<NPROD><<<blocks, 256>>>(w, layers, 4, d);
    hipDeviceSynchronize();
    hipEventRecord(e0); k_layers<NPROD><<<blocks, 256>>>(w, layers, tiles, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * tiles * layers * 32 * 2.0 * 128 * 128;
    printf("%d products per K-step, %d layers: %.1f fp32-equivalent TFLOP/s\n", NPROD, layers, flop / ms / 1e9);
}
int main() {
    const int layers = 4;
    const size_t n = (size_t)layers * 8 * 3 * 4 * 64 * 4;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) { unsigned short lo = 0x3c00 + (i % 97), hi = 0x3b80 + (i % 89); unsigned v = ((unsigned)hi << 16) | lo; h[i] = *(float*)&v; }
    float *w, *d; hipMalloc(&w, n * 4); hipMalloc(&d, 512 * 256 * 4);
    hipMemcpy(w, h.data(), n * 4, hipMemcpyHostToDevice);
    run<3>(w, d, layers); run<6>(w, d, layers);
    return 0;
}
