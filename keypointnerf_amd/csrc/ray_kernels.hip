// ray_kernels.hip — scene preparation, ray set-up, hierarchical sampler and compositor (gfx950).
//
// These are the HBM-/latency-bound stages around the field evaluation:
//   k_scene_table, k_pack_rgbm, k_nchw_to_nhwc : kpn_scene_prepare
//   k_ray_bbox, k_make_rays                    : reference src/model.py:1019-1043, 1178-1237
//   k_coarse_z                                 : src/model.py:1045-1055 (uniform=True)
//   k_rgba2out                                 : src/model.py:1150-1176, one wavefront per ray,
//                                                transmittance by a 64-lane exclusive product scan
//   k_importance / k_fine_samples_w            : src/model.py:1110-1148 (+ sort(cat) :1076)
#include "kpn_device.h"

// ---------------------------------------------------------------------------------------------
// per-view table: KRT rows, extrinsic rows, camera centre = inverse(KRT)[:3,3] (model.py:823-824),
// keypoints in the camera frame (spatial.py:85).  One thread per view; double Gauss-Jordan.
// The running max |value| of everything kpn_scene_prepare copies (scene flags[0], zeroed by k_scene_table, which runs first on
// the stream): as int bit patterns |x| orders like the floats, and a NaN sits above +inf.  One atomic per wavefront, and only while
// the wave's maximum is above what is already there (a few per launch).  Every thread of the wave must call it.
__device__ __forceinline__ void kpn_note_absmax(float* __restrict__ flags, float a) {
    int u = __float_as_int(a) & 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(u, o); u = w > u ? w : u; }
    int* f = reinterpret_cast<int*>(flags);
    if ((threadIdx.x & 63) == 0 && u > *reinterpret_cast<volatile int*>(f)) atomicMax(f, u);
}
__device__ __forceinline__ float kpn_absmax2(float a, float b) {   // max(|a|, |b|) that keeps a NaN (fmaxf would drop it)
    const int x = __float_as_int(a) & 0x7fffffff, y = __float_as_int(b) & 0x7fffffff;
    return __int_as_float(x > y ? x : y);
}

__global__ void k_scene_table(int V, const float* __restrict__ KRT, const float* __restrict__ extrin,
                              const float* __restrict__ kpt3d, float* __restrict__ table, float* __restrict__ flags) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < KPN_SCENE_FLAG_FLOATS) flags[v] = 0.0f;
    if (v >= V) return;
    float* tb = table + (size_t)v * KPN_TBL_STRIDE;
    const float* M = KRT + v * 16;
    const float* E = extrin + v * 16;
    for (int i = 0; i < 12; ++i) { tb[KPN_TBL_KRT + i] = M[i]; tb[KPN_TBL_EXT + i] = E[i]; }
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = (double)M[i * 4 + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
        if (p != c) for (int j = 0; j < 8; ++j) { double t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
        const double d = a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] /= d;
        for (int r = 0; r < 4; ++r) if (r != c) {
            const double f = a[r][c];
            for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 3; ++i) tb[KPN_TBL_CPOS + i] = (float)a[i][7];
    tb[KPN_TBL_CPOS + 3] = 0.0f;
    for (int k = 0; k < KPN_NKPT; ++k)
        for (int i = 0; i < 3; ++i)
            tb[KPN_TBL_KCAM + k * 3 + i] =
                KADD(kpn_dot3(kpt3d[k * 3 + 0], kpt3d[k * 3 + 1], kpt3d[k * 3 + 2], E[i * 4 + 0], E[i * 4 + 1], E[i * 4 + 2]),
                     E[i * 4 + 3]);
    for (int i = KPN_TBL_KCAM + KPN_NKPT * 3; i < KPN_TBL_STRIDE; ++i) tb[i] = 0.0f;
}

// (V,3,H,W) image + (V,H,W) mask bytes -> (V,H,W,4) [r,g,b,fg]
__global__ void k_pack_rgbm(int64_t npix_total, int64_t HW, const float* __restrict__ img,
                            const uint8_t* __restrict__ mask, float* __restrict__ out, float* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float amax = 0.0f;
    if (i < npix_total) {
        const int64_t v = i / HW, p = i - v * HW;
        float4 o;
        o.x = img[(v * 3 + 0) * HW + p];
        o.y = img[(v * 3 + 1) * HW + p];
        o.z = img[(v * 3 + 2) * HW + p];
        o.w = mask ? (mask[i] ? 1.0f : 0.0f) : 1.0f;
        reinterpret_cast<float4*>(out)[i] = o;
        amax = kpn_absmax2(kpn_absmax2(o.x, o.y), o.z);
    }
    kpn_note_absmax(flags, amax);
}

// (V,C,h,w) -> (V,h,w,C); one thread per output element (writes coalesced)
__global__ void k_nchw_to_nhwc(int64_t total, int C, int64_t hw, const float* __restrict__ in, float* __restrict__ out,
                               float* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float x = 0.0f;
    if (i < total) {
        const int c = (int)(i % C);
        const int64_t vp = i / C;
        const int64_t v = vp / hw, p = vp - v * hw;
        x = in[(v * C + c) * hw + p];
        out[i] = x;
    }
    kpn_note_absmax(flags, x);
}

// ---------------------------------------------------------------------------------------------
// ray_bbox_intersection, model.py:1178-1237
__device__ __forceinline__ void kpn_ray_aabb(const float* __restrict__ bounds, float ox, float oy, float oz, float dx,
                                             float dy, float dz, float& near_o, float& far_o, int& hit) {
    float bmin[3], bmax[3], o[3] = {ox, oy, oz}, d[3] = {dx, dy, dz};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        bmin[i] = KADD(bounds[i], -0.01f);
        bmax[i] = KADD(bounds[3 + i], 0.01f);
        if (fabsf(d[i]) < 1e-5f) d[i] = 1e-5f;  // model.py:1198 (sign dropped, as in the reference)
    }
    int cnt = 0;
    float pint[2][3];
    const float eps = 1e-6f;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int ax = s % 3;
        const float bound = (s < 3) ? bmin[ax] : bmax[ax];
        const float t = KSUB(bound, o[ax]) / d[ax];
        const float p0 = KADD(KMUL(t, d[0]), o[0]), p1 = KADD(KMUL(t, d[1]), o[1]), p2 = KADD(KMUL(t, d[2]), o[2]);
        const int inside = (p0 >= KSUB(bmin[0], eps)) && (p0 <= KADD(bmax[0], eps)) && (p1 >= KSUB(bmin[1], eps)) &&
                           (p1 <= KADD(bmax[1], eps)) && (p2 >= KSUB(bmin[2], eps)) && (p2 <= KADD(bmax[2], eps));
        if (inside) {
            if (cnt < 2) { pint[cnt][0] = p0; pint[cnt][1] = p1; pint[cnt][2] = p2; }
            ++cnt;
        }
    }
    if (cnt == 2) {
        const float nr = sqrtf(kpn_dot3(d[0], d[1], d[2], d[0], d[1], d[2]));
        float dd[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float e0 = KSUB(pint[k][0], o[0]), e1 = KSUB(pint[k][1], o[1]), e2 = KSUB(pint[k][2], o[2]);
            dd[k] = sqrtf(kpn_dot3(e0, e1, e2, e0, e1, e2)) / nr;
        }
        near_o = fminf(dd[0], dd[1]); far_o = fmaxf(dd[0], dd[1]); hit = 1;
    } else {
        near_o = 1.0f; far_o = 1.0f; hit = 0;
    }
}

__global__ void k_ray_bbox(int64_t R, const float* __restrict__ bounds, const float* __restrict__ orig,
                           const float* __restrict__ dirs, float* __restrict__ near_o, float* __restrict__ far_o,
                           uint8_t* __restrict__ hit_o) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float n, f; int h;
    kpn_ray_aabb(bounds, orig[0], orig[1], orig[2], dirs[r * 3 + 0], dirs[r * 3 + 1], dirs[r * 3 + 2], n, f, h);
    near_o[r] = n; far_o[r] = f; hit_o[r] = (uint8_t)h;
}

// model.py:1026-1043 for the pixel grid px = x0+ix*step, py = y0+iy*stepy
__global__ void k_make_rays(const float* __restrict__ K, const float* __restrict__ RT, float znear, float zfar,
                            const float* __restrict__ bounds, int x0, int y0, int step, int stepy, int nx, int ny,
                            const int* __restrict__ pix, float* __restrict__ dirs, float* __restrict__ cam_pos,
                            float* __restrict__ near_o, float* __restrict__ far_o) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t R = (int64_t)nx * ny;
    // inverse of K[:3,:3] by cofactors in double (the reference calls th.inverse, model.py:1031)
    const double a = K[0], b = K[1], c = K[2], d = K[4], e = K[5], f = K[6], g = K[8], h = K[9], i = K[10];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    float iK[9];
    iK[0] = (float)((e * i - f * h) / det); iK[1] = (float)((c * h - b * i) / det); iK[2] = (float)((b * f - c * e) / det);
    iK[3] = (float)((f * g - d * i) / det); iK[4] = (float)((a * i - c * g) / det); iK[5] = (float)((c * d - a * f) / det);
    iK[6] = (float)((d * h - e * g) / det); iK[7] = (float)((b * g - a * h) / det); iK[8] = (float)((a * e - b * d) / det);
    float cp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)  // cam_pos = -t^T R, model.py:1036
        cp[k] = -kpn_dot3(RT[0 * 4 + 3], RT[1 * 4 + 3], RT[2 * 4 + 3], RT[0 * 4 + k], RT[1 * 4 + k], RT[2 * 4 + k]);
    if (r == 0) { cam_pos[0] = cp[0]; cam_pos[1] = cp[1]; cam_pos[2] = cp[2]; }
    if (r >= R) return;
    const int iy = (int)(r / nx), ix = (int)(r - (int64_t)iy * nx);
    // eval: strided grid (model.py:1019-1022); train: explicit patch pixels (x,y) (model.py:1008-1017)
    const float gx = pix ? (float)pix[r * 2 + 0] : (float)(x0 + ix * step);
    const float gy = pix ? (float)pix[r * 2 + 1] : (float)(y0 + iy * stepy), gz = 1.0f;
    float cr[3], cn[3], cf[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        cr[k] = kpn_dot3(gx, gy, gz, iK[k * 3 + 0], iK[k * 3 + 1], iK[k * 3 + 2]);
        cn[k] = kpn_dot3(KMUL(znear, gx), KMUL(znear, gy), KMUL(znear, gz), iK[k * 3 + 0], iK[k * 3 + 1], iK[k * 3 + 2]);
        cf[k] = kpn_dot3(KMUL(zfar, gx), KMUL(zfar, gy), KMUL(zfar, gz), iK[k * 3 + 0], iK[k * 3 + 1], iK[k * 3 + 2]);
    }
    float nr_ = sqrtf(kpn_dot3(cn[0], cn[1], cn[2], cn[0], cn[1], cn[2]));
    float fr_ = sqrtf(kpn_dot3(cf[0], cf[1], cf[2], cf[0], cf[1], cf[2]));
    float w[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) w[k] = kpn_dot3(cr[0], cr[1], cr[2], RT[0 * 4 + k], RT[1 * 4 + k], RT[2 * 4 + k]);
    const float nrm = fmaxf(sqrtf(kpn_dot3(w[0], w[1], w[2], w[0], w[1], w[2])), 1e-12f);
    const float d0 = w[0] / nrm, d1 = w[1] / nrm, d2 = w[2] / nrm;
    dirs[r * 3 + 0] = d0; dirs[r * 3 + 1] = d1; dirs[r * 3 + 2] = d2;
    float z1, z2; int hit;
    kpn_ray_aabb(bounds, cp[0], cp[1], cp[2], d0, d1, d2, z1, z2, hit);
    if (hit && z1 > nr_) nr_ = z1;  // model.py:1040-1043
    if (hit && z2 < fr_) fr_ = z2;
    near_o[r] = nr_; far_o[r] = fr_;
}

// torch.linspace(0,1,steps) in fp32 (ATen evaluates symmetrically from both ends)
__device__ __forceinline__ float kpn_linspace01(int i, int steps) {
    if (steps == 1) return 0.0f;
    const float step = 1.0f / (float)(steps - 1);
    const int half = steps / 2;
    return (i < half) ? KMUL(step, (float)i) : KSUB(1.0f, KMUL(step, (float)(steps - 1 - i)));
}

// z = near + (far-near)*linspace  (model.py:1045,1055)
// u != NULL: stratified jitter of the train branch (model.py:1049-1053): t = lower + u*(upper-lower) with
// lower = cat[t[:1], t_mid], upper = cat[t_mid, t[-1:]]
__global__ void k_coarse_z(int64_t R, int S, const float* __restrict__ near_i, const float* __restrict__ far_i,
                           const float* __restrict__ u, float* __restrict__ z) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * S) return;
    const int64_t r = i / S;
    const int s = (int)(i - r * S);
    float t = kpn_linspace01(s, S);
    if (u) {
        const float lo = (s == 0) ? t : KMUL(0.5f, KADD(t, kpn_linspace01(s - 1, S)));
        const float hi = (s == S - 1) ? t : KMUL(0.5f, KADD(kpn_linspace01(s + 1, S), t));
        t = KADD(lo, KMUL(u[i], KSUB(hi, lo)));
    }
    z[i] = KADD(near_i[r], KMUL(KSUB(far_i[r], near_i[r]), t));
}

// ---------------------------------------------------------------------------------------------
// rgba2out (model.py:1150-1176): one wavefront per ray.  Lane l owns the contiguous samples
// [l*per, (l+1)*per); the exclusive transmittance prod_{j<i}(1-c_j) is a 64-lane product scan of the
// per-lane products, the four weighted sums are butterfly reductions.
// A ray's loads (records, depths) are issued one ray ahead of its arithmetic: the kernel is latency-bound (a dependent
// chain of loads, a 6-step scan and 6-step reductions per ray), so the next ray's memory latency hides under them.
#define KPN_MAX_PER_LANE 8  // supports S <= 512
template <int PER>
struct kpn_ray_samples { float sig[PER], sd[PER], cr[PER], cg[PER], cb[PER], z[PER], dist[PER]; };
// the merged field records of a fine pass that re-used the coarse values (k_rgba2out reads them through `src` in place), written
// out sample by sample: kpn_render_stages.rgba_fine
__global__ __launch_bounds__(256) void k_merge_rgba(int64_t n, int S, int Sc, const float* __restrict__ rgba_c, const float* __restrict__ rgba_n,
                                                    const int16_t* __restrict__ src, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // sample index r * S + k
    if (i >= n * S) return;
    const int64_t r = i / S;
    const int id = src[i];
    const float* q = id < Sc ? rgba_c + (r * Sc + id) * 5 : rgba_n + (r * (S - Sc) + (id - Sc)) * 5;
#pragma unroll
    for (int c = 0; c < 5; ++c) out[i * 5 + c] = q[c];
}

template <int PER>
__global__ __launch_bounds__(256) void k_rgba2out(int64_t R, int S, const float* __restrict__ rgba,
                                                  const float* __restrict__ z, float* __restrict__ color,
                                                  float* __restrict__ depth, float* __restrict__ alpha,
                                                  float* __restrict__ contrib, float* __restrict__ sdf,
                                                  const int16_t* __restrict__ src, const float* __restrict__ rgba_new, int Sc) {
    // src != NULL: sample i of ray r is record src[r*S+i] of the ray's coarse records (rgba, Sc per ray) when < Sc, else of
    // its new-sample records (rgba_new, S - Sc per ray) — the merged list of the fine pass without materialising it
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int per = (S + 63) / 64;   // <= PER
    auto load = [&](int64_t r, kpn_ray_samples<PER>& o) {
        const float* zz = z + r * S;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = lane * per + k;
            o.sig[k] = 0.0f; o.sd[k] = 0.0f; o.cr[k] = 0.0f; o.cg[k] = 0.0f; o.cb[k] = 0.0f; o.z[k] = 0.0f; o.dist[k] = 0.0f;
            if (k < per && i < S) {
                const float* q;
                if (!src) q = rgba + (r * S + i) * 5;
                else {
                    const int id = src[r * S + i];
                    q = id < Sc ? rgba + (r * Sc + id) * 5 : rgba_new + (r * (S - Sc) + (id - Sc)) * 5;
                }
                o.sig[k] = q[0]; o.sd[k] = q[1]; o.cr[k] = q[2]; o.cg[k] = q[3]; o.cb[k] = q[4];
                o.z[k] = zz[i];
                o.dist[k] = (i + 1 < S) ? (zz[i + 1] - zz[i]) : 1e10f;  // :1166
            }
        }
    };
    kpn_ray_samples<PER> cur{}, nxt{};
    if (wave < R) load(wave, cur);
    for (int64_t r = wave; r < R; r += nwaves) {
        if (r + nwaves < R) load(r + nwaves, nxt);
        float c[PER];
        float tl = 1.0f;  // product of (1-c) over this lane's samples
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            c[k] = 0.0f;
            const int i = lane * per + k;
            if (k < per && i < S) {
                c[k] = 1.0f - expf(-cur.sig[k] * cur.dist[k]);  // :1167
                tl *= (1.0f - c[k]);
            }
        }
        // inclusive product scan across lanes, then shift to exclusive
        float incl = tl;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float o = __shfl_up(incl, d);
            if (lane >= d) incl *= o;
        }
        float T = __shfl_up(incl, 1);
        if (lane == 0) T = 1.0f;
        float s_r = 0.f, s_g = 0.f, s_b = 0.f, s_a = 0.f, s_s = 0.f, s_d = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = lane * per + k;
            if (k < per && i < S) {
                const float cw = c[k] * T;  // :1168-1169
                T *= (1.0f - c[k]);
                if (contrib) contrib[r * S + i] = cw;
                s_r += cur.cr[k] * cw; s_g += cur.cg[k] * cw; s_b += cur.cb[k] * cw;
                s_a += cw; s_s += cur.sd[k] * cw; s_d += cur.z[k] * cw;
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            s_r += __shfl_xor(s_r, m); s_g += __shfl_xor(s_g, m); s_b += __shfl_xor(s_b, m);
            s_a += __shfl_xor(s_a, m); s_s += __shfl_xor(s_s, m); s_d += __shfl_xor(s_d, m);
        }
        if (lane == 0) {
            color[r * 3 + 0] = s_r; color[r * 3 + 1] = s_g; color[r * 3 + 2] = s_b;
            alpha[r] = s_a;
            sdf[r] = s_s / (s_a + 1e-8f);    // :1173
            depth[r] = s_d / (s_a + 1e-8f);  // :1174
        }
        cur = nxt;
    }
}

// Backward of rgba2out: one thread per ray.  Forward sweep: transmittance T_i (parked in the output row) and the
// sums; reverse sweep with the division-free recurrence Q_{i-1} = g_i a_i + (1-a_i) Q_i, where g_i = dL/dc_i and
// dL/da_i = T_i (g_i - Q_i)  (the cumprod's 1/(1-a_i) never appears, so a_i == 1 is harmless).
__global__ void k_rgba2out_bwd(int64_t R, int S, const float* __restrict__ rgba, const float* __restrict__ z,
                               const float* __restrict__ d_color, const float* __restrict__ d_depth,
                               const float* __restrict__ d_alpha, const float* __restrict__ d_sdf, float* __restrict__ d_rgba) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float* q = rgba + r * S * 5;
    const float* zz = z + r * S;
    float* dq = d_rgba + r * S * 5;
    float T = 1.0f, A = 0.0f, Ssum = 0.0f, Dsum = 0.0f;
    for (int i = 0; i < S; ++i) {
        const float dist = (i + 1 < S) ? (zz[i + 1] - zz[i]) : 1e10f;
        const float a = 1.0f - expf(-q[i * 5 + 0] * dist);
        const float c = a * T;
        dq[i * 5 + 0] = T;
        T *= (1.0f - a);
        A += c; Ssum += q[i * 5 + 1] * c; Dsum += zz[i] * c;
    }
    const float dc0 = d_color ? d_color[r * 3 + 0] : 0.f, dc1 = d_color ? d_color[r * 3 + 1] : 0.f, dc2 = d_color ? d_color[r * 3 + 2] : 0.f;
    const float dA = d_alpha ? d_alpha[r] : 0.f, dS = d_sdf ? d_sdf[r] : 0.f, dD = d_depth ? d_depth[r] : 0.f;
    const float inv = 1.0f / (A + 1e-8f);
    const float gA = dA - (dS * Ssum + dD * Dsum) * inv * inv;  // sdf and depth also depend on every c_i through alpha
    float Q = 0.0f;
    for (int i = S - 1; i >= 0; --i) {
        const float dist = (i + 1 < S) ? (zz[i + 1] - zz[i]) : 1e10f;
        const float e = expf(-q[i * 5 + 0] * dist);  // 1 - a_i
        const float a = 1.0f - e;
        const float Ti = dq[i * 5 + 0];
        const float c = a * Ti;
        const float g = dc0 * q[i * 5 + 2] + dc1 * q[i * 5 + 3] + dc2 * q[i * 5 + 4] + gA + dS * q[i * 5 + 1] * inv + dD * zz[i] * inv;
        dq[i * 5 + 0] = Ti * (g - Q) * dist * e;     // d sigma_i = dL/da_i * da_i/dsigma_i
        dq[i * 5 + 1] = c * dS * inv;                // d sdf_i
        dq[i * 5 + 2] = c * dc0; dq[i * 5 + 3] = c * dc1; dq[i * 5 + 4] = c * dc2;
        Q = g * a + e * Q;
    }
}

// The same backward with one wavefront per ray (S <= 512; a training patch is ~1,000 rays, far too few threads for the form above:
// 16 workgroups on 256 CUs, 86 us per call).  Lane l owns the contiguous samples [l*per, (l+1)*per) as in k_rgba2out: the
// transmittance is the same 64-lane product scan, and the reverse recurrence, being affine in Q (Q_{i-1} = e_i Q_i + g_i a_i),
// is a 64-lane suffix scan of the per-lane maps (E, B): Q before the lane's first sample = E * (Q behind its last) + B.
template <int PER>
__global__ __launch_bounds__(256) void k_rgba2out_bwd_w(int64_t R, int S, const float* __restrict__ rgba, const float* __restrict__ z,
                                                        const float* __restrict__ d_color, const float* __restrict__ d_depth,
                                                        const float* __restrict__ d_alpha, const float* __restrict__ d_sdf,
                                                        float* __restrict__ d_rgba) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= R) return;   // a whole wavefront
    const int per = (S + 63) / 64;   // <= PER
    const float* zz = z + r * S;
    float sd[PER], cr[PER], cg[PER], cb[PER], zv[PER], dist[PER], e[PER], a[PER], Tk[PER];
    float tl = 1.0f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = lane * per + k;
        sd[k] = cr[k] = cg[k] = cb[k] = zv[k] = dist[k] = 0.0f;
        e[k] = 1.0f; a[k] = 0.0f;
        if (k < per && i < S) {
            const float* q = rgba + (r * S + i) * 5;
            sd[k] = q[1]; cr[k] = q[2]; cg[k] = q[3]; cb[k] = q[4];
            zv[k] = zz[i];
            dist[k] = (i + 1 < S) ? (zz[i + 1] - zz[i]) : 1e10f;
            e[k] = expf(-q[0] * dist[k]);   // 1 - a_i
            a[k] = 1.0f - e[k];
            tl *= (1.0f - a[k]);
        }
    }
    float incl = tl;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_up(incl, d);
        if (lane >= d) incl *= o;
    }
    float T = __shfl_up(incl, 1);
    if (lane == 0) T = 1.0f;
    float A = 0.0f, Ssum = 0.0f, Dsum = 0.0f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        Tk[k] = T;
        const float c = a[k] * T;
        T *= (1.0f - a[k]);
        A += c; Ssum += sd[k] * c; Dsum += zv[k] * c;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { A += __shfl_xor(A, m); Ssum += __shfl_xor(Ssum, m); Dsum += __shfl_xor(Dsum, m); }
    const float dc0 = d_color ? d_color[r * 3 + 0] : 0.f, dc1 = d_color ? d_color[r * 3 + 1] : 0.f, dc2 = d_color ? d_color[r * 3 + 2] : 0.f;
    const float dA = d_alpha ? d_alpha[r] : 0.f, dS = d_sdf ? d_sdf[r] : 0.f, dD = d_depth ? d_depth[r] : 0.f;
    const float inv = 1.0f / (A + 1e-8f);
    const float gA = dA - (dS * Ssum + dD * Dsum) * inv * inv;
    float g[PER];
    float E = 1.0f, B = 0.0f;   // this lane's map, last sample first
#pragma unroll
    for (int k = PER - 1; k >= 0; --k) {
        g[k] = dc0 * cr[k] + dc1 * cg[k] + dc2 * cb[k] + gA + dS * sd[k] * inv + dD * zv[k] * inv;
        B = g[k] * a[k] + e[k] * B;
        E *= e[k];
    }
    // inclusive suffix scan: (E, B) of lane l becomes the map of lanes l..63
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float Eo = __shfl_down(E, d), Bo = __shfl_down(B, d);
        if (lane + d < 64) { B = E * Bo + B; E *= Eo; }
    }
    float Q = __shfl_down(B, 1);   // the lanes behind this one, applied to Q = 0 behind the last sample
    if (lane == 63) Q = 0.0f;
    float* dq = d_rgba + r * S * 5;
#pragma unroll
    for (int k = PER - 1; k >= 0; --k) {
        const int i = lane * per + k;
        if (k < per && i < S) {
            const float c = a[k] * Tk[k];
            dq[i * 5 + 0] = Tk[k] * (g[k] - Q) * dist[k] * e[k];
            dq[i * 5 + 1] = c * dS * inv;
            dq[i * 5 + 2] = c * dc0; dq[i * 5 + 3] = c * dc1; dq[i * 5 + 4] = c * dc2;
        }
        Q = g[k] * a[k] + e[k] * Q;
    }
}

// ---------------------------------------------------------------------------------------------
// importance_sample (model.py:1110-1148): one thread per ray, sequential cdf (torch.cumsum order),
// searchsorted(right=True) by bisection in LDS.  contrib (R,Dm2), zin (R,Dm2+1), u (R,n)|NULL -> out (R,n)
#define KPN_IS_MAXD 129
__global__ __launch_bounds__(64) void k_importance(int64_t R, int Dm2, int n, const float* __restrict__ contrib,
                                                   const float* __restrict__ zin, const float* __restrict__ u,
                                                   float* __restrict__ out) {
    __shared__ float cdf_s[64][KPN_IS_MAXD];
    const int t = threadIdx.x;
    const int64_t r = (int64_t)blockIdx.x * 64 + t;
    if (r >= R) return;  // no barriers / cross-lane ops below
    const int C = Dm2 + 1;
    const float* c = contrib + r * Dm2;
    const float* zz = zin + r * C;
    float sum = 0.0f;
    for (int i = 0; i < Dm2; ++i) sum = KADD(sum, KADD(c[i], 1e-5f));  // :1120-1121
    float run = 0.0f;
    cdf_s[t][0] = 0.0f;
    for (int i = 0; i < Dm2; ++i) {                                     // :1122-1123
        run = KADD(run, KADD(c[i], 1e-5f) / sum);
        cdf_s[t][i + 1] = run;
    }
    for (int k = 0; k < n; ++k) {
        const float s = u ? u[r * n + k] : kpn_linspace01(k, n);
        int lo = 0, hi = C;  // first idx with cdf[idx] > s  (:1131 right=True)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf_s[t][mid] <= s) lo = mid + 1; else hi = mid;
        }
        const int ip = lo - 1 < 0 ? 0 : lo - 1;      // :1132
        const int in = lo > C - 1 ? C - 1 : lo;      // :1133
        const float num = KSUB(s, cdf_s[t][ip]);
        float den = KSUB(cdf_s[t][in], cdf_s[t][ip]);
        if (den < 1e-5f) den = 1.0f;                 // :1146
        out[r * n + k] = KADD(zz[ip], KMUL(num / den, KSUB(zz[in], zz[ip])));  // :1147
    }
}

// z_mid (:1074) and contrib[...,1:-1] (:1075) followed by importance sampling and z_fine = sort(cat[z, z_new]) (:1076),
// one wavefront per ray (ties between equal depths are ordered coarse-first, which changes nothing — equal depths on a
// ray are the same point).  Lane k draws sample k.  The two cumsums of the reference stay sequential (torch.cumsum order on the CPU, which the
// oracle is pinned to): with SMALL (Sc, Sf <= 64) lane i holds element i and the running value walks the lanes through
// v_readlane; otherwise lane 0 walks LDS.  The new samples are ordered by rank counting unless they already are
// (uniform u: nearly always); the merged order of two sorted lists is two binary searches per element, with rank
// counting over the whole list as the fall-back when the coarse depths are not sorted.  Sc, Sf <= 128.
// One workgroup = 4 wavefronts = 4 rays per iteration.
#ifndef KPN_SIMT_EMU
#define KPN_READLANE_F(v, i) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (i)))
#else
#define KPN_READLANE_F(v, i) __shfl((v), (i))
#endif
template <bool SMALL>
__global__ __launch_bounds__(256) void k_fine_samples_w(int64_t R, int Sc, int Sf, const float* __restrict__ zc,
                                                        const float* __restrict__ contrib, const float* __restrict__ u,
                                                        float* __restrict__ zf, float* __restrict__ znew,
                                                        int16_t* __restrict__ src) {
    constexpr int NE = SMALL ? 1 : 2;   // elements of one list per lane
    constexpr int W = 64 * NE;
    __shared__ float q_s[SMALL ? 1 : 4][SMALL ? 1 : W];   // (c_i + 1e-5), then / sum (LDS walk only)
    __shared__ float cdf_s[4][W + 1];
    __shared__ float zn_s[4][W];       // new samples as drawn
    __shared__ float ev_s[4][2 * W];   // all Sc + Sf depths: coarse, then new (sorted)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int Dm2 = Sc - 2, C = Sc - 1, Sfull = Sc + Sf;
    for (int64_t r = (int64_t)blockIdx.x * 4 + w; r < R; r += (int64_t)gridDim.x * 4) {
        const float* c = contrib + r * Sc + 1;
        const float* z = zc + r * Sc;
        for (int i = lane; i < Sc; i += 64) ev_s[w][i] = z[i];
        if constexpr (SMALL) {
            float q = lane < Dm2 ? KADD(c[lane], 1e-5f) : 0.0f;
            float sum = 0.0f;
            for (int i = 0; i < Dm2; ++i) sum = KADD(sum, KPN_READLANE_F(q, i));   // :1120-1121, sequential
            q = q / sum;
            float run = 0.0f, mine = 0.0f;
            for (int i = 0; i < Dm2; ++i) {                                        // :1122-1123, sequential
                run = KADD(run, KPN_READLANE_F(q, i));
                mine = lane == i + 1 ? run : mine;
            }
            if (lane < C) cdf_s[w][lane] = mine;
        } else {
            for (int i = lane; i < Dm2; i += 64) q_s[w][i] = KADD(c[i], 1e-5f);
            KPN_WAVE_SYNC();
            float sum = 0.0f;
            if (lane == 0)
                for (int i = 0; i < Dm2; ++i) sum = KADD(sum, q_s[w][i]);
            sum = __shfl(sum, 0);
            for (int i = lane; i < Dm2; i += 64) q_s[w][i] = q_s[w][i] / sum;
            KPN_WAVE_SYNC();
            if (lane == 0) {
                float run = 0.0f;
                cdf_s[w][0] = 0.0f;
                for (int i = 0; i < Dm2; ++i) { run = KADD(run, q_s[w][i]); cdf_s[w][i + 1] = run; }
            }
        }
        KPN_WAVE_SYNC();
        // inverse-CDF samples (:1125-1147)
        float v[NE];
        for (int e = 0; e < NE; ++e) {
            const int k = lane + 64 * e;
            v[e] = 0.0f;
            if (k < Sf) {
                const float sv = u ? u[r * Sf + k] : kpn_linspace01(k, Sf);
                int lo = 0, hi = C;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (cdf_s[w][mid] <= sv) lo = mid + 1; else hi = mid;
                }
                const int ip = lo - 1 < 0 ? 0 : lo - 1;
                const int in = lo > C - 1 ? C - 1 : lo;
                const float num = KSUB(sv, cdf_s[w][ip]);
                float den = KSUB(cdf_s[w][in], cdf_s[w][ip]);
                if (den < 1e-5f) den = 1.0f;
                const float zp = KMUL(0.5f, KADD(ev_s[w][ip + 1], ev_s[w][ip]));
                const float zq = KMUL(0.5f, KADD(ev_s[w][in + 1], ev_s[w][in]));
                v[e] = KADD(zp, KMUL(num / den, KSUB(zq, zp)));
                zn_s[w][k] = v[e];
            }
        }
        KPN_WAVE_SYNC();
        // order the new samples
        int rk[NE];
        int unsorted = 0;
        for (int e = 0; e < NE; ++e) {
            const int k = lane + 64 * e;
            rk[e] = k;
            unsorted |= (k + 1 < Sf) && !(v[e] <= zn_s[w][k + 1 < W ? k + 1 : k]);
        }
        if (__any(unsorted)) {
            for (int e = 0; e < NE; ++e) rk[e] = 0;
            for (int j = 0; j < Sf; ++j) {
                const float o = zn_s[w][j];
                for (int e = 0; e < NE; ++e) rk[e] += (o < v[e]) || (o == v[e] && j < lane + 64 * e);
            }
        }
        for (int e = 0; e < NE; ++e)
            if (lane + 64 * e < Sf) ev_s[w][Sc + rk[e]] = v[e];
        KPN_WAVE_SYNC();
        if (znew)
            for (int k = lane; k < Sf; k += 64) znew[r * Sf + k] = ev_s[w][Sc + k];
        // merged order
        int csorted = 1;
        for (int i = lane; i + 1 < Sc; i += 64) csorted &= ev_s[w][i] <= ev_s[w][i + 1];
        if (__all(csorted)) {
            // coarse i goes to i + #{new < it}, new k (sorted position) to k + #{coarse <= it}: a permutation of 0..Sfull-1
            for (int i = lane; i < Sc; i += 64) {
                const float a = ev_s[w][i];
                int lo = 0, hi = Sf;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (ev_s[w][Sc + mid] < a) lo = mid + 1; else hi = mid; }
                zf[r * Sfull + i + lo] = a;
                if (src) src[r * Sfull + i + lo] = (int16_t)i;
            }
            for (int k = lane; k < Sf; k += 64) {
                const float b = ev_s[w][Sc + k];
                int lo = 0, hi = Sc;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (ev_s[w][mid] <= b) lo = mid + 1; else hi = mid; }
                zf[r * Sfull + k + lo] = b;
                if (src) src[r * Sfull + k + lo] = (int16_t)(Sc + k);
            }
        } else {
            for (int id = lane; id < Sfull; id += 64) {
                const float a = ev_s[w][id];
                int rank = 0;
                for (int j = 0; j < Sfull; ++j) { const float o = ev_s[w][j]; rank += (o < a) || (o == a && j < id); }
                zf[r * Sfull + rank] = a;
                if (src) src[r * Sfull + rank] = (int16_t)id;
            }
        }
        KPN_WAVE_SYNC();
    }
}

// ---------------------------------------------------------------------------------------------
// Output side: clamp + quantise + CHW->HWC (model.py:427-430,496), MSE/PSNR (zju_evaluator.py:16-19)
__global__ void k_frame_to_rgb8(int HW, int bgr, const float* __restrict__ chw, uint8_t* __restrict__ hwc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = fminf(fmaxf(chw[(size_t)c * HW + i], 0.0f), 1.0f);
        hwc[(size_t)i * 3 + (bgr ? 2 - c : c)] = (uint8_t)(int)KMUL(v, 255.0f);  // numpy astype(uint8): truncation
    }
}

// grid-stride squared-error sum in fp64; the last workgroup to finish (ticket) folds the partials
__global__ __launch_bounds__(256) void k_mse_psnr(int64_t n, const float* __restrict__ a, const float* __restrict__ b,
                                                  double* __restrict__ partial, int* __restrict__ ticket,
                                                  double* __restrict__ out2) {
    __shared__ double red[256];
    __shared__ int last;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = KSUB(a[i], b[i]);
        acc += (double)KMUL(d, d);  // (pred-gt)**2 is an fp32 op in the reference
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = red[0];
        __threadfence();
        last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        double tot = 0.0;
        for (unsigned k = 0; k < gridDim.x; ++k) tot += ((volatile double*)partial)[k];
        const double mse = tot / (double)n;
        out2[0] = mse;
        out2[1] = -10.0 * log(mse) / log(10.0);
    }
}

// ---------------------------------------------------------------------------------------------
// pix_loss's L1 term (reference src/utils.py:164-168): loss = lambda * mean|src - tar|, and what autograd derives for it,
// d loss / d src = lambda * sign(src - tar) / n (sign(0) = 0, torch's abs backward) — the seed gradient of
// kpn_render_rays_train_backward for tex_fg (lambda_l1_c, coarse) and tex_fg_fine (lambda_l1, fine), src/utils.py:128-145.
// Deterministic: per-block fp64 partial sums, the last block adds them in block order.
__global__ __launch_bounds__(256) void k_pix_l1(int64_t n, float lambda, const float* __restrict__ src, const float* __restrict__ tar,
                                                double* __restrict__ partial, int* __restrict__ ticket, float* __restrict__ loss,
                                                float* __restrict__ d_src) {
    __shared__ double red[256];
    __shared__ int last;
    double acc = 0.0;
    const float gscale = lambda / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = KSUB(src[i], tar[i]);
        acc += (double)fabsf(d);
        if (d_src) d_src[i] = d > 0.0f ? gscale : (d < 0.0f ? -gscale : 0.0f);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = red[0];
        __threadfence();
        last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        double tot = 0.0;
        for (unsigned k = 0; k < gridDim.x; ++k) tot += ((volatile double*)partial)[k];
        loss[0] = lambda * (float)(tot / (double)n);
    }
}

// ---------------------------------------------------------------------------------------------
// SSIM as ZJUEvaluator._compute_ssim computes it (reference src/zju_evaluator.py:21-45): skimage 0.19's
// structural_similarity(pred, gt, multichannel=True) on the crop [y0, y0+h) x [x0, x0+w) of two float32 images —
// 7x7 uniform window (scipy.ndimage.uniform_filter: separable, fp64 accumulation, fp32 result per axis), sample
// covariance (49/48), data_range 2 (skimage's default for float images), K1 = 0.01, K2 = 0.03, mean over the interior
// (3-pixel border cropped) and the 3 channels.
// pass 1: vertical 7-tap means of x, y, xx, yy, xy per channel -> tmp[5][3][h-6][w]
__global__ __launch_bounds__(256) void k_ssim_vertical(const float* __restrict__ pred, const float* __restrict__ gt, int H, int W,
                                                       int x0, int y0, int w, int h, float* __restrict__ tmp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int hv = h - 6;
    if (i >= (int64_t)3 * hv * w) return;
    const int x = (int)(i % w), y = (int)((i / w) % hv), c = (int)(i / ((int64_t)w * hv));
    const float* a = pred + ((size_t)c * H + y0 + y) * W + x0 + x;
    const float* b = gt + ((size_t)c * H + y0 + y) * W + x0 + x;
    double s[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < 7; ++k) {
        const float u = a[(size_t)k * W], v = b[(size_t)k * W];
        s[0] += (double)u; s[1] += (double)v;
        s[2] += (double)KMUL(u, u); s[3] += (double)KMUL(v, v); s[4] += (double)KMUL(u, v);  // im1*im1 etc. are fp32 arrays
    }
    const size_t plane = (size_t)3 * hv * w;
    for (int q = 0; q < 5; ++q) tmp[q * plane + i] = (float)(s[q] / 7.0);
}
// pass 2: horizontal 7-tap means, the SSIM map, its sum in fp64 (last workgroup folds the partials)
__global__ __launch_bounds__(256) void k_ssim_map(const float* __restrict__ tmp, int w, int h, double* __restrict__ partial,
                                                  int* __restrict__ ticket, double* __restrict__ out) {
    __shared__ double red[256];
    __shared__ int last;
    const int hv = h - 6, wv = w - 6;
    const int64_t n = (int64_t)3 * hv * wv;
    const size_t plane = (size_t)3 * hv * w;
    const float C1 = KMUL(KMUL(0.01f, 2.0f), KMUL(0.01f, 2.0f)), C2 = KMUL(KMUL(0.03f, 2.0f), KMUL(0.03f, 2.0f));
    const float cov_norm = 49.0f / 48.0f;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % wv), y = (int)((i / wv) % hv), c = (int)(i / ((int64_t)wv * hv));
        float m[5];
        for (int q = 0; q < 5; ++q) {
            const float* t = tmp + q * plane + ((size_t)c * hv + y) * w + x;
            double s = 0.0;
            for (int k = 0; k < 7; ++k) s += (double)t[k];
            m[q] = (float)(s / 7.0);
        }
        const float ux = m[0], uy = m[1];
        const float vx = KMUL(cov_norm, KSUB(m[2], KMUL(ux, ux))), vy = KMUL(cov_norm, KSUB(m[3], KMUL(uy, uy)));
        const float vxy = KMUL(cov_norm, KSUB(m[4], KMUL(ux, uy)));
        const float A1 = KADD(KMUL(KMUL(2.0f, ux), uy), C1), A2 = KADD(KMUL(2.0f, vxy), C2);
        const float B1 = KADD(KADD(KMUL(ux, ux), KMUL(uy, uy)), C1), B2 = KADD(KADD(vx, vy), C2);
        acc += (double)(KMUL(A1, A2) / KMUL(B1, B2));
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = red[0];
        __threadfence();
        last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        double tot = 0.0;
        for (unsigned k = 0; k < gridDim.x; ++k) tot += ((volatile double*)partial)[k];
        out[0] = tot / (double)n;
    }
}
