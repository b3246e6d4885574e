// fuse_bwd_kernels.hip — reverse pass of the per-point kernel k_fuse_color: view pooling + MLPUNetFusion.layers2
// (reference src/utils.py:500-518, 612-647, 722-748, 577-587) and eval_func's packing (src/model.py:981-996).
//
//   k_fuse_bwd : per tile of 32 valid points —
//       recompute  pooled mean/var over views from the row scratch k_geo_rows wrote, layers2.0/.1 (k_fuse_color's
//                  arithmetic), dumping every Linear's input row-major for k_weight_grad;
//       reverse    d[sdf_raw, rad] (through eval_func's relu when mode = 1) -> layers2.2 (two rank-1 VALU updates)
//                  -> layers2.1^T, layers2.0^T on the matrix cores (BSEG_G2_*T, register-chained) -> d pooled;
//                  pooling reverse d x_v = pw_v (d mean' + 2 (x_v - mean) d var), d mean' = d mean - 2 d var
//                  sum_u pw_u (x_u - mean) (the weights sum to pwsum / (pwsum + 1e-6), not 1);
//       output     d x_view rows [row = (tile*V + v)*32 + p][64], the input of k_geo_rows_bwd.
// The colour head's reverse is k_color_bwd (below); it runs first and hands d lat to k_fuse_bwd.
#include "kpn_device.h"

// Plain stores here: k_color_bwd reads most of its own dumps back in its reverse half, and with streaming stores
// (-DKPN_DUMP_NT2, as in k_geo_rows_bwd) those reads miss: measured k_color_bwd 555 -> 592 us per call.
#if defined(KPN_DUMP_NT2) && !defined(KPN_SIMT_EMU)
typedef float kpn_row_nt4 __attribute__((ext_vector_type(4)));
#define KPN_ST_ROW4(p, v) do { const float4 v_ = (v); kpn_row_nt4 n_; n_[0] = v_.x; n_[1] = v_.y; n_[2] = v_.z; n_[3] = v_.w; __builtin_nontemporal_store(n_, reinterpret_cast<kpn_row_nt4*>(p)); } while (0)
#else
#define KPN_ST_ROW4(p, v) (*reinterpret_cast<float4*>(p) = (v))
#endif
// 4*NQ registers of one 32-feature block in the chained layout <-> row-major: regs 4q..4q+3 = features 8q+4h..+3
template <int NQ>
__device__ __forceinline__ void kpn_ld_chain(const float* __restrict__ base, int h, float (&x)[4 * NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float4 f = *reinterpret_cast<const float4*>(base + 8 * q + 4 * h);
        x[4 * q + 0] = f.x; x[4 * q + 1] = f.y; x[4 * q + 2] = f.z; x[4 * q + 3] = f.w;
    }
}
template <int NQ>
__device__ __forceinline__ void kpn_st_chain(float* __restrict__ base, int h, const float (&x)[4 * NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        KPN_ST_ROW4(base + 8 * q + 4 * h, make_float4(x[4 * q + 0], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]));
}

struct kpn_fuse_bwd_bufs {
    float* Xp;    // [points][128] pooled (mean64 | var64)
    float* Xh0;   // [points][64]  softplus(layers2.0)
    float* Xh1;   // [points][64]  softplus(layers2.1)
    float* D20;   // [points][64]  dA of layers2.0
    float* D21;   // [points][64]  dA of layers2.1
    float* D22;   // [points][2]   d [sdf_raw, rad]
    float* dxrows;  // [rows][64]  d x_view
    const float* Dcmp;  // [points][24] d lat written by k_color_bwd, or nullptr (geometry outputs only)
};

__global__ __launch_bounds__(256, 2) void k_fuse_bwd(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                                     const int* __restrict__ list, const int* __restrict__ count_ptr,
                                                     int* __restrict__ tickets, const float* __restrict__ xscr, int mode,
                                                     const float* __restrict__ d_out, kpn_fuse_bwd_bufs bufs) {
    const int lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    const int count = *count_ptr;
    const int ntiles = (count + KPN_TILE - 1) / KPN_TILE;
    const int V = sc.V;
    const uint32_t keep = sc.keep;
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(tickets, 1);
        t = __shfl(t, 0);
        if (t >= ntiles) break;
        const int ci_raw = t * KPN_TILE + p;
        const float live = ci_raw < count ? 1.0f : 0.0f;  // pad lanes recompute the last point with a zero upstream gradient
        const int ci = ci_raw < count ? ci_raw : count - 1;
        const int64_t n = list[ci];
        const size_t prow = (size_t)t * KPN_TILE + p;

        // ---- recompute: pooling (as k_fuse_color) ----
        const float4* rows = reinterpret_cast<const float4*>(xscr) + ((size_t)t * V * KPN_ROW_SLABS) * 64;
        float pooled[64];
        const float pwsum = kpn_pool_views(rows, V, keep, lane, p, pooled);
        // pooled[16b + r] = mean feature 32b + rowmap(r,h); pooled[32 + 16b + r] = var of the same feature
        {
            float* xp = bufs.Xp + prow * 128;
#pragma unroll
            for (int g = 0; g < 16; ++g)
                *reinterpret_cast<float4*>(xp + 64 * (g / 8) + 32 * ((g / 4) % 2) + 8 * (g % 4) + 4 * h) =
                    make_float4(pooled[4 * g + 0], pooled[4 * g + 1], pooled[4 * g + 2], pooled[4 * g + 3]);
        }
        // ---- recompute: layers2.0, layers2.1 (weights from L2; this kernel is not the hot one) ----
        kpn_f32x16 h0[2], h1[2], o2[1];
        kpn_load_bias<2>(wp + kpn_seg_boff(SEG_G2_0), h, h0);
        kpn_mfma_layer_regs<64, 2, 4, 0>(wp + kpn_seg_woff(SEG_G2_0), lane, pooled, h0);
        kpn_load_bias<2>(wp + kpn_seg_boff(SEG_G2_1), h, h1);
        float* xh0 = bufs.Xh0 + prow * 64;
        kpn_mfma_layer<32, 2, 4, 0>(wp + kpn_seg_woff(SEG_G2_1), lane, [&](auto gi, float (&x)[4]) {
            constexpr int g = decltype(gi)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = kpn_softplus100(h0[g / 4][(g % 4) * 4 + i]);
            *reinterpret_cast<float4*>(xh0 + 32 * (g / 4) + 8 * (g % 4) + 4 * h) = make_float4(x[0], x[1], x[2], x[3]);
        }, h1);
        kpn_load_bias<1>(wp + kpn_seg_boff(SEG_G2_2), h, o2);
        float* xh1 = bufs.Xh1 + prow * 64;
        kpn_mfma_layer<32, 1, 4, 0>(wp + kpn_seg_woff(SEG_G2_2), lane, [&](auto gi, float (&x)[4]) {
            constexpr int g = decltype(gi)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = kpn_softplus100(h1[g / 4][(g % 4) * 4 + i]);
            *reinterpret_cast<float4*>(xh1 + 32 * (g / 4) + 8 * (g % 4) + 4 * h) = make_float4(x[0], x[1], x[2], x[3]);
        }, o2);
        // rows 0 (sdf_raw), 1 (rad) live in regs 0,1 of the h = 0 lanes: every lane needs rad's sign
        float rad = __shfl(o2[0][1], p);

        // ---- upstream gradient of [sdf_raw, rad] ----
        const float* go = d_out + n * 5;
        float d_sdf, d_rad;
        if (mode == 1) {  // eval_func, mask = 1: out0 = relu(rad + noise), out1 = sdf_raw (model.py:981-996)
            if (ps.noise) rad = RADD(rad, RMUL(ps.noise[n], ps.noise_std));
            d_rad = rad > 0.0f ? go[0] : 0.0f;
            d_sdf = go[1];
        } else { d_sdf = go[0]; d_rad = go[1]; }
        d_sdf *= live; d_rad *= live;
        if (h == 0) { bufs.D22[prow * 2 + 0] = d_sdf; bufs.D22[prow * 2 + 1] = d_rad; }

        // ---- layers2 reverse ----
        kpn_f32x16 dh0[2], dpool[4];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) dh0[b][r] = 0.0f;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) dpool[b][r] = 0.0f;
        {
            const float* r_sdf = wp + kpn_brow_off(BROW_G2_2_SDF);
            const float* r_rad = wp + kpn_brow_off(BROW_G2_2_RAD);
            float* d21 = bufs.D21 + prow * 64;
            kpn_mfma_layer<32, 2, 4, 0>(wp + kpn_bseg_woff(BSEG_G2_1T), lane, [&](auto gi, float (&x)[4]) {
                constexpr int g = decltype(gi)::value;
                const int col = 32 * (g / 4) + 8 * (g % 4) + 4 * h;
                const float4 ws = *reinterpret_cast<const float4*>(r_sdf + (2 * (g / 4) + h) * 16 + 4 * (g % 4));
                const float4 wr = *reinterpret_cast<const float4*>(r_rad + (2 * (g / 4) + h) * 16 + 4 * (g % 4));
                const float4 s = *reinterpret_cast<const float4*>(xh1 + col);
                x[0] = (ws.x * d_sdf + wr.x * d_rad) * kpn_softplus100_grad_from_value(s.x);
                x[1] = (ws.y * d_sdf + wr.y * d_rad) * kpn_softplus100_grad_from_value(s.y);
                x[2] = (ws.z * d_sdf + wr.z * d_rad) * kpn_softplus100_grad_from_value(s.z);
                x[3] = (ws.w * d_sdf + wr.w * d_rad) * kpn_softplus100_grad_from_value(s.w);
                *reinterpret_cast<float4*>(d21 + col) = make_float4(x[0], x[1], x[2], x[3]);
            }, dh0);
            float* d20 = bufs.D20 + prow * 64;
            kpn_mfma_layer<32, 4, 4, 0>(wp + kpn_bseg_woff(BSEG_G2_0T), lane, [&](auto gi, float (&x)[4]) {
                constexpr int g = decltype(gi)::value;
                const int col = 32 * (g / 4) + 8 * (g % 4) + 4 * h;
                const float4 s = *reinterpret_cast<const float4*>(xh0 + col);
                x[0] = dh0[g / 4][(g % 4) * 4 + 0] * kpn_softplus100_grad_from_value(s.x);
                x[1] = dh0[g / 4][(g % 4) * 4 + 1] * kpn_softplus100_grad_from_value(s.y);
                x[2] = dh0[g / 4][(g % 4) * 4 + 2] * kpn_softplus100_grad_from_value(s.z);
                x[3] = dh0[g / 4][(g % 4) * 4 + 3] * kpn_softplus100_grad_from_value(s.w);
                *reinterpret_cast<float4*>(d20 + col) = make_float4(x[0], x[1], x[2], x[3]);
            }, dpool);
        }
        if (bufs.Dcmp) {  // ibr_compress_gfeat^T: the colour head's d lat joins d pooled
            float d12[12];
            kpn_ld_chain<3>(bufs.Dcmp + prow * 24, h, d12);
            kpn_mfma_layer_regs<12, 4, 4, 0>(wp + kpn_bseg_woff(BSEG_CMPT), lane, d12, dpool);
        }
        // ---- pooling reverse: dpool[b] (b < 2) = d mean, dpool[2 + b] = d var, same lane-register layout as pooled ----
        float sres[32];  // sum_u pw_u (x_u - mean)
#pragma unroll
        for (int i = 0; i < 32; ++i) sres[i] = 0.0f;
        for (int v = 0; v < V; ++v) {
            if (!((keep >> v) & 1u)) continue;
            const float4* src = rows + ((size_t)v * KPN_ROW_SLABS) * 64;
            const float pw = src[8 * 64 + p].w / RADD(pwsum, 1e-6f);
#pragma unroll
            for (int q4 = 0; q4 < 8; ++q4) {
                const float4 x = src[q4 * 64 + lane];
                sres[4 * q4 + 0] += pw * (x.x - pooled[4 * q4 + 0]); sres[4 * q4 + 1] += pw * (x.y - pooled[4 * q4 + 1]);
                sres[4 * q4 + 2] += pw * (x.z - pooled[4 * q4 + 2]); sres[4 * q4 + 3] += pw * (x.w - pooled[4 * q4 + 3]);
            }
        }
        for (int v = 0; v < V; ++v) {
            float* drow = bufs.dxrows + (((size_t)t * V + v) * KPN_TILE + p) * 64;
            const bool on = (keep >> v) & 1u;
            const float4* src = rows + ((size_t)v * KPN_ROW_SLABS) * 64;
            const float pw = on ? src[8 * 64 + p].w / RADD(pwsum, 1e-6f) : 0.0f;
#pragma unroll
            for (int q4 = 0; q4 < 8; ++q4) {
                const float4 x = src[q4 * 64 + lane];
                const float xe[4] = {x.x, x.y, x.z, x.w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * q4 + e, b = i / 16, r = i % 16;
                    const float dvar = dpool[2 + b][r];
                    const float dm = dpool[b][r] - 2.0f * dvar * sres[i];
                    o[e] = on ? pw * (dm + 2.0f * (xe[e] - pooled[i]) * dvar) : 0.0f;
                }
                // regs 4(q4%4)..+3 of block q4/4 = features 32(q4/4) + 8(q4%4) + 4h .. +3
                *reinterpret_cast<float4*>(drow + 32 * (q4 / 4) + 8 * (q4 % 4) + 4 * h) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

// =============================================================================================
// Colour head reverse: ibr_compress_gfeat + IBRRenderingHead (reference src/model.py:784-843, 1267-1302) + the
// feat_tex gather.  One tile = 32 valid points, V <= 3 source views.
//   forward   recomputed exactly as k_fuse_color (weights streamed from L2 instead of LDS), every Linear's input
//             dumped row-major per (point, view) row for k_weight_grad;
//   reverse   softmax blend -> out_layer -> vis_layer2 -> vis_layer1 -> base_layer per view (transposed segments
//             BSEG_*T, single-output layers as rank-1 VALU updates with the forward row vectors), the weighted
//             mean/var over views, the blend weights' dependence on ani_al, ray_encoder, d feat_tex (LDS-transposed
//             scatter), and d lat per point (Dcmp), which k_fuse_bwd turns into d pooled.
// Row strides of the dumps (floats); "x'" order = [lat24 | rgb3 | tex8] (kpn_common.h):
#ifndef KPN_COLOR_OCC
#define KPN_COLOR_OCC 1
#endif
#define KPN_LD_XDIR 36   // elu(ray_encoder.2) in x' order (+1 pad)
#define KPN_LD_XBL 108   // [mean'(35)+pad | var'(35)+pad | x'(35)+pad]
#define KPN_LD_XO0 40    // [x(32) | vis | ray_diff(4) | pad 3]
#define KPN_LD_DV11 36   // dA(vis_layer1.2): 33 + pad
struct kpn_color_bufs {
    float *Xrd, *Xe1, *Xdir, *Xbl, *Xb1, *Xa, *Xv10, *Xv11, *Xt33, *Xv20, *Xv21, *Xo0, *Xo1, *Xo2;
    float *Do2, *Do1, *Do0, *Dv21, *Dv20, *Dv11, *Dv10, *Dbl1, *Dbl0, *Dre1, *Dre0;
    float* Dcmp;  // [points][24] d lat
    float* dtex;  // V x th x tw x 8, accumulated
    float* dani;  // d ani_al, accumulated (the flat gradient's last entry)
};
// the 19 registers of an x'-ordered 35-vector: 16 chained (rows 0..31) + rows 32..34 in the h = 0 lanes
__device__ __forceinline__ void kpn_st_xprime(float* __restrict__ base, int h, const float (&x)[19]) {
    float c[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = x[r];
    kpn_st_chain<4>(base, h, c);
    if (h == 0) *reinterpret_cast<float4*>(base + 32) = make_float4(x[16], x[17], x[18], 0.0f);
}
__device__ __forceinline__ void kpn_ld_xprime(const float* __restrict__ base, int h, float (&x)[19]) {
    float c[16];
    kpn_ld_chain<4>(base, h, c);
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = c[r];
    const float4 f = *reinterpret_cast<const float4*>(base + 32);
    x[16] = h ? 0.0f : f.x; x[17] = h ? 0.0f : f.y; x[18] = h ? 0.0f : f.z;
}
__device__ __forceinline__ float kpn_elu_grad_from_out(float y) { return y > 0.0f ? 1.0f : y + 1.0f; }
// sum over the two halves of a point (lanes p and p + 32)
__device__ __forceinline__ float kpn_pair_sum(float x) { return x + __shfl_xor(x, 32); }

// VMAX = 3: the per-view scalars (dot, source colour, logit, ...) stay in registers; VMAX = KPN_MAXV: any view count, the
// small per-view arrays are indexed dynamically (private memory).
// [base, base + nfloats) of the packed buffer = the whole backward segments bseg0 .. bseg1-1, copied into LDS in the layout
// kpn_load_group<NQ, 1> reads (kpn_stage_lds_range, field_kernels.hip, for the transposed segments)
__device__ __forceinline__ void kpn_stage_lds_brange(const float* __restrict__ wp, float* __restrict__ wlds, int base, int nfloats,
                                                     int bseg0, int bseg1) {
    const float4* src = reinterpret_cast<const float4*>(wp + base);
    float4* dst = reinterpret_cast<float4*>(wlds);
    for (int i = threadIdx.x; i < nfloats / 4; i += blockDim.x) {
        int j = i;
#pragma unroll
        for (int seg = BSEG_CMPT; seg < BSEG_COUNT; ++seg) {
            if (seg < bseg0 || seg >= bseg1) continue;
            const int nq = kpn_bseg_shapes[seg].g * kpn_bseg_shapes[seg].nob / 4;
            if (nq == 1) continue;
            const int w0 = (kpn_bseg_woff(seg) - base) / 4, w1 = w0 + kpn_bseg_wfloats(seg) / 4;
            if (i >= w0 && i < w1) {
                const int r = i - w0, g = r / (64 * nq), e = r - g * 64 * nq;   // e = lane * nq + q
                j = w0 + g * 64 * nq + (e % nq) * 64 + e / nq;
            }
        }
        dst[j] = src[i];
    }
}
// The colour head is fourteen small layers per view in each direction, every one waiting for its weights: fetched from L2 a layer
// at a time (one wave per SIMD: 482 registers) the kernel ran at a fifth of its arithmetic.  Everything the per-view part reads —
// the forward segments from ray_encoder.0 on with the scalars and row vectors (64 KB) and the transposed segments from
// ray_encoder.2^T on (78 KB) — is therefore staged in LDS once per persistent workgroup (KPN_COLOR_WLDS, the default);
// ibr_compress_gfeat and its transpose (once per tile) stay in L2.
#ifndef KPN_COLOR_WLDS
#define KPN_COLOR_WLDS 1
#endif
constexpr int kpn_color_fbase() { return kpn_seg_woff(SEG_RE_0); }
constexpr int kpn_color_ffloats() { return kpn_fwd_floats() - kpn_color_fbase(); }
constexpr int kpn_color_bbase() { return kpn_bseg_woff(BSEG_RE_1T); }
constexpr int kpn_color_bfloats() { return kpn_bseg_woff(BSEG_COUNT) - kpn_color_bbase(); }
template <int VMAX>
__global__ __launch_bounds__(256, KPN_COLOR_OCC) void k_color_bwd(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                                      const int* __restrict__ list, const int* __restrict__ count_ptr,
                                                      int* __restrict__ tickets, const float* __restrict__ xscr,
                                                      const float* __restrict__ d_out, kpn_color_bufs B) {
    const int lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    const int count = *count_ptr;
    const int ntiles = (count + KPN_TILE - 1) / KPN_TILE;
    const int V = sc.V;  // <= VMAX (the launcher picks the instantiation)
    const uint32_t keep = sc.keep;
#if KPN_COLOR_WLDS
    __shared__ __attribute__((aligned(16))) float wl_f[kpn_color_ffloats()];
    __shared__ __attribute__((aligned(16))) float wl_b[kpn_color_bfloats()];
    kpn_stage_lds_range(wp, wl_f, kpn_color_fbase(), kpn_color_ffloats(), SEG_RE_0, SEG_COUNT);
    kpn_stage_lds_brange(wp, wl_b, kpn_color_bbase(), kpn_color_bfloats(), BSEG_RE_1T, BSEG_COUNT);
    __syncthreads();
    const float* wf = wl_f - kpn_color_fbase();   // biased: the packed-buffer offsets index the LDS copies directly
    const float* wb = wl_b - kpn_color_bbase();
#else
    const float* wf = wp;
    const float* wb = wp;
#endif
    const float ani = wf[kpn_scalar_off() + 0];  // |ani_al|
    __shared__ __attribute__((aligned(16))) float scat_s[4][KPN_TILE][8];
    __shared__ __attribute__((aligned(16))) int4 tap_o[4][KPN_TILE];
    __shared__ __attribute__((aligned(16))) float4 tap_w[4][KPN_TILE];
    const int w4 = threadIdx.x >> 6;
    float dani_acc = 0.0f;                     // this wave's share of d ani_al (every lane holds the tile sums)
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(tickets, 1);
        t = __shfl(t, 0);
        if (t >= ntiles) break;
        const int ci_raw = t * KPN_TILE + p;
        const float live = ci_raw < count ? 1.0f : 0.0f;
        const int ci = ci_raw < count ? ci_raw : count - 1;
        const int64_t n = list[ci];
        const size_t prow = (size_t)t * KPN_TILE + p;
        auto hrow = [&](int v) { return ((size_t)t * V + v) * KPN_TILE + p; };

        // ---------------- forward: pooling -> lat (as k_fuse_color) ----------------
        const float4* rows = reinterpret_cast<const float4*>(xscr) + ((size_t)t * V * KPN_ROW_SLABS) * 64;
        float lat0[16];
        {
            float pooled[64];
            kpn_pool_views(rows, V, keep, lane, p, pooled);
            kpn_f32x16 acc[1];
            kpn_load_bias<1>(wp + kpn_seg_boff(SEG_CMP), h, acc);
            kpn_mfma_layer_regs<64, 1, 4, 0>(wp + kpn_seg_woff(SEG_CMP), lane, pooled, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) lat0[r] = acc[0][r];
        }
        // ---------------- forward: per-view x' = [lat|rgb|tex] + ray_encoder(ray_diff) ----------------
        float dotv[VMAX], rgbv[VMAX][3], logit[VMAX];
        float emin = 3.0e38f, esum = 0.0f;
        for (int v = 0; v < V; ++v) {
            kpn_view_gather g;
            kpn_gather_view(reinterpret_cast<const float4*>(xscr) + ((size_t)(t * V + v) * KPN_ROW_SLABS + 8) * 64, lane, h, g);
            dotv[v] = g.rd[3];
            rgbv[v][0] = g.rgb[0]; rgbv[v][1] = g.rgb[1]; rgbv[v][2] = g.rgb[2];
            emin = fminf(emin, kpn_fast_exp(RMUL(ani, RSUB(g.rd[3], 1.0f))));
            if (h == 0) *reinterpret_cast<float4*>(B.Xrd + hrow(v) * 4) = make_float4(g.rd[0], g.rd[1], g.rd[2], g.rd[3]);
            if (!((keep >> v) & 1u)) continue;
            const float in4[4] = {h ? g.rd[1] : g.rd[0], h ? g.rd[3] : g.rd[2], 0.0f, 0.0f};
            kpn_f32x16 a1[1];
            kpn_load_bias<1>(wf + kpn_seg_boff(SEG_RE_0), h, a1);
            kpn_mfma_layer_regs<4, 1, 4, KPN_COLOR_WLDS>(wf + kpn_seg_woff(SEG_RE_0), lane, in4, a1);
            float in8[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) in8[r] = kpn_elu(a1[0][r]);
            kpn_st_chain<2>(B.Xe1 + hrow(v) * 16, h, in8);
            kpn_f32x16 a2[2];
            kpn_load_bias<2>(wf + kpn_seg_boff(SEG_RE_1), h, a2);
            kpn_mfma_layer_regs<8, 2, 4, KPN_COLOR_WLDS>(wf + kpn_seg_woff(SEG_RE_1), lane, in8, a2);
            float dir[19], xq[19];
#pragma unroll
            for (int r = 0; r < 16; ++r) { dir[r] = kpn_elu(a2[0][r]); xq[r] = dir[r] + lat0[r]; }
#pragma unroll
            for (int i = 0; i < 3; ++i) { dir[16 + i] = kpn_elu(a2[1][i]); xq[16 + i] = dir[16 + i] + g.fadd[4 + i]; }
            xq[12] += g.fadd[0]; xq[13] += g.fadd[1]; xq[14] += g.fadd[2]; xq[15] += g.fadd[3];
            kpn_st_xprime(B.Xdir + hrow(v) * KPN_LD_XDIR, h, dir);
            kpn_st_xprime(B.Xbl + hrow(v) * KPN_LD_XBL + 72, h, xq);
        }
        for (int v = 0; v < V; ++v)
            if ((keep >> v) & 1u) esum = RADD(esum, RSUB(kpn_fast_exp(RMUL(ani, RSUB(dotv[v], 1.0f))), emin));
        auto blend_w = [&](int v) { return RSUB(kpn_fast_exp(RMUL(ani, RSUB(dotv[v], 1.0f))), emin) / RADD(esum, 1e-8f); };
        // fused mean / var over views of x' (utils.py:91-95)
        float mv[40];
#pragma unroll
        for (int i = 0; i < 40; ++i) mv[i] = 0.0f;
        for (int pass = 0; pass < 2; ++pass)
            for (int v = 0; v < V; ++v) {
                if (!((keep >> v) & 1u)) continue;
                float xq[19];
                kpn_ld_xprime(B.Xbl + hrow(v) * KPN_LD_XBL + 72, h, xq);
                const float wv = blend_w(v);
#pragma unroll
                for (int i = 0; i < 19; ++i) {
                    if (pass == 0) mv[i] = RADD(mv[i], RMUL(xq[i], wv));
                    else { const float d = RSUB(xq[i], mv[i]); mv[20 + i] = RADD(mv[20 + i], RMUL(wv, RMUL(d, d))); }
                }
            }
        kpn_f32x16 base[2];
        kpn_load_bias<2>(wf + kpn_seg_boff(SEG_BL_0A), h, base);
        kpn_mfma_layer_regs<40, 2, 4, KPN_COLOR_WLDS>(wf + kpn_seg_woff(SEG_BL_0A), lane, mv, base);
        float meanq[19], varq[19];
#pragma unroll
        for (int i = 0; i < 19; ++i) { meanq[i] = mv[i]; varq[i] = mv[20 + i]; }
        // ---------------- forward: the head per view, dumping every Linear's input ----------------
        for (int v = 0; v < V; ++v) {
            logit[v] = -1.0e9f;  // masked_fill (model.py:1300)
            if (!((keep >> v) & 1u)) continue;
            const size_t hr = hrow(v);
            const float wv = blend_w(v);
            float xq[19], xin[20];
            kpn_ld_xprime(B.Xbl + hr * KPN_LD_XBL + 72, h, xq);
            kpn_st_xprime(B.Xbl + hr * KPN_LD_XBL + 0, h, meanq);
            kpn_st_xprime(B.Xbl + hr * KPN_LD_XBL + 36, h, varq);
#pragma unroll
            for (int i = 0; i < 19; ++i) xin[i] = xq[i];
            xin[19] = 0.0f;
            kpn_f32x16 a[2] = {base[0], base[1]};
            kpn_mfma_layer_regs<20, 2, 4, KPN_COLOR_WLDS>(wf + kpn_seg_woff(SEG_BL_0B), lane, xin, a);
            kpn_f32x16 xa[1];
            kpn_load_bias<1>(wf + kpn_seg_boff(SEG_BL_1), h, xa);
            float* xb1 = B.Xb1 + hr * 64;
            kpn_mfma_layer<32, 1, 4, KPN_COLOR_WLDS>(wf + kpn_seg_woff(SEG_BL_1), lane, [&](auto gi, float (&x)[4]) {
                constexpr int gq = decltype(gi)::value;
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = kpn_elu(a[gq / 4][(gq % 4) * 4 + i]);
                *reinterpret_cast<float4*>(xb1 + 32 * (gq / 4) + 8 * (gq % 4) + 4 * h) = make_float4(x[0], x[1], x[2], x[3]);
            }, xa);
            float x[16], tin[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { x[r] = kpn_elu(xa[0][r]); tin[r] = x[r] * wv; }
            kpn_st_chain<4>(B.Xa + hr * 32, h, x);
            kpn_st_chain<4>(B.Xv10 + hr * 32, h, tin);
            kpn_f32x16 va[1], vb[1];
            kpn_load_bias<1>(wf + kpn_seg_boff(SEG_V1_0), h, va);
            kpn_mfma_layer_regs<16, 1, 4, KPN_COLOR_WLDS>(wf + kpn_seg_woff(SEG_V1_0), lane, tin, va);
#pragma unroll
            for (int r = 0; r < 16; ++r) tin[r] = kpn_elu(va[0][r]);
            kpn_st_chain<4>(B.Xv11 + hr * 32, h, tin);
            kpn_load_bias<1>(wf + kpn_seg_boff(SEG_V1_1), h, vb);
            kpn_mfma_layer_regs<16, 1, 4, KPN_COLOR_WLDS>(wf + kpn_seg_woff(SEG_V1_1), lane, tin, vb);
            const float visr = kpn_elu(kpn_row_dot(wf + kpn_row_off(ROW_V1_VIS), h, tin));
            const float sv = kpn_sigmoid(visr);
            if (h == 0) { B.Xt33[hr * 2 + 0] = visr; B.Xt33[hr * 2 + 1] = 0.0f; }
#pragma unroll
            for (int r = 0; r < 16; ++r) { x[r] = x[r] + kpn_elu(vb[0][r]); tin[r] = x[r] * sv; }
            kpn_st_chain<4>(B.Xv20 + hr * 32, h, tin);
            kpn_load_bias<1>(wf + kpn_seg_boff(SEG_V2_0), h, va);
            kpn_mfma_layer_regs<16, 1, 4, KPN_COLOR_WLDS>(wf + kpn_seg_woff(SEG_V2_0), lane, tin, va);
#pragma unroll
            for (int r = 0; r < 16; ++r) tin[r] = kpn_elu(va[0][r]);
            kpn_st_chain<4>(B.Xv21 + hr * 32, h, tin);
            const float vis = kpn_sigmoid(kpn_row_dot(wf + kpn_row_off(ROW_V2_1), h, tin));
            const float4 rd = *reinterpret_cast<const float4*>(B.Xrd + hr * 4);
            float oin[20];
#pragma unroll
            for (int r = 0; r < 16; ++r) oin[r] = x[r];
            oin[16] = h ? rd.x : vis;
            oin[17] = h ? rd.z : rd.y;
            oin[18] = h ? 0.0f : rd.w;
            oin[19] = 0.0f;
            kpn_st_chain<4>(B.Xo0 + hr * KPN_LD_XO0, h, x);
            if (h == 0) {
                *reinterpret_cast<float4*>(B.Xo0 + hr * KPN_LD_XO0 + 32) = make_float4(vis, rd.x, rd.y, rd.z);
                *reinterpret_cast<float4*>(B.Xo0 + hr * KPN_LD_XO0 + 36) = make_float4(rd.w, 0.0f, 0.0f, 0.0f);
            }
            kpn_load_bias<1>(wf + kpn_seg_boff(SEG_O_0), h, va);
            kpn_mfma_layer_regs<20, 1, 4, KPN_COLOR_WLDS>(wf + kpn_seg_woff(SEG_O_0), lane, oin, va);
            float o8[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) o8[r] = kpn_elu(va[0][r]);
            kpn_st_chain<2>(B.Xo1 + hr * 16, h, o8);
            kpn_load_bias<1>(wf + kpn_seg_boff(SEG_O_1), h, va);
            kpn_mfma_layer_regs<8, 1, 4, KPN_COLOR_WLDS>(wf + kpn_seg_woff(SEG_O_1), lane, o8, va);
#pragma unroll
            for (int r = 0; r < 16; ++r) tin[r] = kpn_elu(va[0][r]);
            {
                float o4[4] = {tin[0], tin[1], tin[2], tin[3]};
                kpn_st_chain<1>(B.Xo2 + hr * 8, h, o4);
            }
            logit[v] = kpn_row_dot(wf + kpn_row_off(ROW_O_2), h, tin);
        }
        // ---------------- reverse: softmax blend of the source colours (model.py:1301) ----------------
        const float* go = d_out + n * 5;
        const float dr = go[2] * live, dg = go[3] * live, db = go[4] * live;
        float lmax = -3.0e38f;
        for (int v = 0; v < V; ++v) lmax = fmaxf(lmax, logit[v]);
        float den = 0.0f, sm[VMAX], rdot[VMAX], rtot = 0.0f;
        for (int v = 0; v < V; ++v) { sm[v] = ((keep >> v) & 1u) ? kpn_fast_exp(logit[v] - lmax) : 0.0f; den += sm[v]; }
        for (int v = 0; v < V; ++v) {
            sm[v] /= den;
            rdot[v] = rgbv[v][0] * dr + rgbv[v][1] * dg + rgbv[v][2] * db;
            rtot += sm[v] * rdot[v];
        }
        // ---------------- reverse: the head per view ----------------
        kpn_f32x16 dasum[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) dasum[b][r] = 0.0f;
        float dwv[VMAX];
#pragma unroll
        for (int v = 0; v < VMAX; ++v) dwv[v] = 0.0f;
        for (int v = 0; v < V; ++v) {
            if (!((keep >> v) & 1u)) continue;
            const size_t hr = hrow(v);
            const float wv = blend_w(v);
            const float dl = sm[v] * (rdot[v] - rtot);
            if (h == 0) { B.Do2[hr * 2 + 0] = dl; B.Do2[hr * 2 + 1] = 0.0f; }
            // out_layer.4 (1 x 8): rank-1; then out_layer.2^T, out_layer.0^T
            float d8[4];
            {
                float o4[4];
                kpn_ld_chain<1>(B.Xo2 + hr * 8, h, o4);
                const float* rw = wf + kpn_row_off(ROW_O_2) + h * 16;
#pragma unroll
                for (int r = 0; r < 4; ++r) d8[r] = rw[r] * dl * kpn_elu_grad_from_out(o4[r]);
                kpn_st_chain<1>(B.Do1 + hr * 8, h, d8);
            }
            kpn_f32x16 d16[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) d16[0][r] = 0.0f;
            kpn_mfma_layer_regs<4, 1, 4, KPN_COLOR_WLDS>(wb + kpn_bseg_woff(BSEG_O_1T), lane, d8, d16);
            float d16p[8];
            {
                float o8[8];
                kpn_ld_chain<2>(B.Xo1 + hr * 16, h, o8);
#pragma unroll
                for (int r = 0; r < 8; ++r) d16p[r] = d16[0][r] * kpn_elu_grad_from_out(o8[r]);
                kpn_st_chain<2>(B.Do0 + hr * 16, h, d16p);
            }
            kpn_f32x16 doin[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) doin[b][r] = 0.0f;
            kpn_mfma_layer_regs<8, 2, 4, KPN_COLOR_WLDS>(wb + kpn_bseg_woff(BSEG_O_0T), lane, d16p, doin);
            float dxb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) dxb[r] = doin[0][r];
            const float dvis = __shfl(doin[1][0], p);  // row 32 = block 1 row 0: reg 0 of the h = 0 lane
            // vis_layer2: vis = sigmoid(W t3 + b)
            float xb[16], x1[16];
            kpn_ld_chain<4>(B.Xo0 + hr * KPN_LD_XO0, h, xb);
            kpn_ld_chain<4>(B.Xa + hr * 32, h, x1);
            const float vis = B.Xo0[hr * KPN_LD_XO0 + 32];
            const float ds1 = dvis * vis * (1.0f - vis);
            if (h == 0) { B.Dv21[hr * 2 + 0] = ds1; B.Dv21[hr * 2 + 1] = 0.0f; }
            float dt[16];
            {
                float t3[16];
                kpn_ld_chain<4>(B.Xv21 + hr * 32, h, t3);
                const float* rw = wf + kpn_row_off(ROW_V2_1) + h * 16;
#pragma unroll
                for (int r = 0; r < 16; ++r) dt[r] = rw[r] * ds1 * kpn_elu_grad_from_out(t3[r]);
                kpn_st_chain<4>(B.Dv20 + hr * 32, h, dt);
            }
            kpn_f32x16 dx2[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) dx2[0][r] = 0.0f;
            kpn_mfma_layer_regs<16, 1, 4, KPN_COLOR_WLDS>(wb + kpn_bseg_woff(BSEG_V2_0T), lane, dt, dx2);
            const float visr = B.Xt33[hr * 2 + 0];
            const float sv = kpn_sigmoid(visr);
            float dsv = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { dxb[r] += dx2[0][r] * sv; dsv += dx2[0][r] * xb[r]; }
            dsv = kpn_pair_sum(dsv);
            // vis_layer1: [res(32) | vis] = elu(W t2 + b)
            const float dvr = dsv * sv * (1.0f - sv) * kpn_elu_grad_from_out(visr);
            float d33[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) d33[r] = dxb[r] * kpn_elu_grad_from_out(xb[r] - x1[r]);  // elu(vb) = x2 - x
            kpn_st_chain<4>(B.Dv11 + hr * KPN_LD_DV11, h, d33);
            if (h == 0) *reinterpret_cast<float4*>(B.Dv11 + hr * KPN_LD_DV11 + 32) = make_float4(dvr, 0.0f, 0.0f, 0.0f);
            kpn_f32x16 dt2[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) dt2[0][r] = 0.0f;
            kpn_mfma_layer_regs<16, 1, 4, KPN_COLOR_WLDS>(wb + kpn_bseg_woff(BSEG_V1_1T), lane, d33, dt2);
            {
                float t2[16];
                kpn_ld_chain<4>(B.Xv11 + hr * 32, h, t2);
                const float* rw = wf + kpn_row_off(ROW_V1_VIS) + h * 16;
#pragma unroll
                for (int r = 0; r < 16; ++r) dt[r] = (dt2[0][r] + rw[r] * dvr) * kpn_elu_grad_from_out(t2[r]);
                kpn_st_chain<4>(B.Dv10 + hr * 32, h, dt);
            }
            kpn_f32x16 dx1[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) dx1[0][r] = 0.0f;
            kpn_mfma_layer_regs<16, 1, 4, KPN_COLOR_WLDS>(wb + kpn_bseg_woff(BSEG_V1_0T), lane, dt, dx1);
            float dxa[16], dw_part = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dxa[r] = (dxb[r] + dx1[0][r] * wv) * kpn_elu_grad_from_out(x1[r]);
                dw_part += dx1[0][r] * x1[r];
            }
            dwv[v] += kpn_pair_sum(dw_part);
            kpn_st_chain<4>(B.Dbl1 + hr * 32, h, dxa);
            // base_layer.2^T, then dA of base_layer.0
            kpn_f32x16 dh[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) dh[b][r] = 0.0f;
            kpn_mfma_layer_regs<16, 2, 4, KPN_COLOR_WLDS>(wb + kpn_bseg_woff(BSEG_BL_1T), lane, dxa, dh);
            float da[32];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float hv[16], dd[16];
                kpn_ld_chain<4>(B.Xb1 + hr * 64 + 32 * b, h, hv);
#pragma unroll
                for (int r = 0; r < 16; ++r) { dd[r] = dh[b][r] * kpn_elu_grad_from_out(hv[r]); da[16 * b + r] = dd[r]; dasum[b][r] += dd[r]; }
                kpn_st_chain<4>(B.Dbl0 + hr * 64 + 32 * b, h, dd);
            }
            // d x' of this view through base_layer.0's x columns; parked in Dre1 until the mean/var reverse
            kpn_f32x16 dxq[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) dxq[b][r] = 0.0f;
            kpn_mfma_layer_regs<32, 2, 4, KPN_COLOR_WLDS>(wb + kpn_bseg_woff(BSEG_BL_0BT), lane, da, dxq);
            float dq[19];
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[r] = dxq[0][r];
            dq[16] = dxq[1][0]; dq[17] = dxq[1][1]; dq[18] = dxq[1][2];
            kpn_st_xprime(B.Dre1 + hr * KPN_LD_XDIR, h, dq);
        }
        // ---------------- reverse: weighted mean / var over views (fused_mean_variance) ----------------
        float dmean[19], dvar[19];
        {
            kpn_f32x16 dmv[4];
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) dmv[b][r] = 0.0f;
            float das[32];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) das[16 * b + r] = dasum[b][r];
            kpn_mfma_layer_regs<32, 4, 4, KPN_COLOR_WLDS>(wb + kpn_bseg_woff(BSEG_BL_0AT), lane, das, dmv);
#pragma unroll
            for (int r = 0; r < 16; ++r) { dmean[r] = dmv[0][r]; dvar[r] = dmv[2][r]; }
#pragma unroll
            for (int i = 0; i < 3; ++i) { dmean[16 + i] = h ? 0.0f : dmv[1][i]; dvar[16 + i] = h ? 0.0f : dmv[3][i]; }
        }
        float sres[19];
#pragma unroll
        for (int i = 0; i < 19; ++i) sres[i] = 0.0f;
        for (int v = 0; v < V; ++v) {
            if (!((keep >> v) & 1u)) continue;
            float xq[19];
            kpn_ld_xprime(B.Xbl + hrow(v) * KPN_LD_XBL + 72, h, xq);
            const float wv = blend_w(v);
#pragma unroll
            for (int i = 0; i < 19; ++i) sres[i] += wv * (xq[i] - meanq[i]);
        }
        float dlat[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dlat[r] = 0.0f;
        // per view: finish d x', ray_encoder reverse, d feat_tex scatter
        float P[3], D[3];
        kpn_get_point(ps, n, P, D);
        for (int v = 0; v < V; ++v) {
            if (!((keep >> v) & 1u)) continue;
            const size_t hr = hrow(v);
            const float wv = blend_w(v);
            float xq[19], dq[19], dir[19];
            kpn_ld_xprime(B.Xbl + hr * KPN_LD_XBL + 72, h, xq);
            kpn_ld_xprime(B.Dre1 + hr * KPN_LD_XDIR, h, dq);
            kpn_ld_xprime(B.Xdir + hr * KPN_LD_XDIR, h, dir);
            float dw_part = 0.0f;
#pragma unroll
            for (int i = 0; i < 19; ++i) {
                const float dm = dmean[i] - 2.0f * dvar[i] * sres[i];
                const float d = xq[i] - meanq[i];
                dq[i] += wv * (dm + 2.0f * d * dvar[i]);
                dw_part += xq[i] * dm + d * d * dvar[i];
            }
            dwv[v] += kpn_pair_sum(dw_part);
            // x' rows 0..23 = lat (regs r < 12 of both halves)
#pragma unroll
            for (int r = 0; r < 12; ++r) dlat[r] += dq[r];
            // d tex: x' rows 27..34 = tex 0..7: row 27 = reg 15 (h=0); rows 28..31 = regs 12..15 (h=1); rows 32..34 = block 1 (h=0)
            {
                const float* tb = sc.table + (size_t)v * KPN_TBL_STRIDE;
                // strict arithmetic, like the forward's gather record (kpn_row_record_b<true>): the gradient goes back through the very
                // texels the forward sampled (an advisor finding of round 5: the contracting flavour can land one texel off at an
                // integer boundary)
                float Ps[3], Ds[3];
                kpn_get_point<true>(ps, n, Ps, Ds);
                const kpn_proj q = kpn_project<true>(tb, Ps[0], Ps[1], Ps[2], sc);
                const kpn_taps tt = kpn_make_taps<true>(q.xn, q.yn, sc.th, sc.tw);
                float* sg = scat_s[w4][p];
                if (h == 0) {
                    sg[0] = dq[15]; sg[5] = dq[16]; sg[6] = dq[17]; sg[7] = dq[18];
                    tap_o[w4][p] = make_int4(tt.o00, tt.o01, tt.o10, tt.o11);
                    tap_w[w4][p] = make_float4(tt.w00 * live, tt.w01 * live, tt.w10 * live, tt.w11 * live);
                } else {
                    sg[1] = dq[12]; sg[2] = dq[13]; sg[3] = dq[14]; sg[4] = dq[15];
                }
                KPN_WAVE_SYNC();
                const int npt = min(KPN_TILE, count - t * KPN_TILE);
                kpn_scatter_rle8(B.dtex + (size_t)v * sc.th * sc.tw * 8, scat_s[w4][0], 8, tap_o[w4], tap_w[w4], npt, lane);
                KPN_WAVE_SYNC();
            }
            // ray_encoder reverse: dA(ray_encoder.2) = d x' * elu'(dir), in x' order
#pragma unroll
            for (int i = 0; i < 19; ++i) dq[i] *= kpn_elu_grad_from_out(dir[i]);
            if (h) { dq[16] = 0.0f; dq[17] = 0.0f; dq[18] = 0.0f; }
            kpn_st_xprime(B.Dre1 + hr * KPN_LD_XDIR, h, dq);
            float dq20[20];
#pragma unroll
            for (int i = 0; i < 19; ++i) dq20[i] = dq[i];
            dq20[19] = 0.0f;
            kpn_f32x16 de[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) de[0][r] = 0.0f;
            kpn_mfma_layer_regs<20, 1, 4, KPN_COLOR_WLDS>(wb + kpn_bseg_woff(BSEG_RE_1T), lane, dq20, de);
            float e1[8], de1[8];
            kpn_ld_chain<2>(B.Xe1 + hr * 16, h, e1);
#pragma unroll
            for (int r = 0; r < 8; ++r) de1[r] = de[0][r] * kpn_elu_grad_from_out(e1[r]);
            kpn_st_chain<2>(B.Dre0 + hr * 16, h, de1);
        }
        // d lat -> Dcmp [point][24] (x' rows 0..23 = chained regs r < 12)
        {
            float d12[12];
#pragma unroll
            for (int r = 0; r < 12; ++r) d12[r] = dlat[r];
            kpn_st_chain<3>(B.Dcmp + prow * 24, h, d12);
        }
        // ---------------- reverse: blend weights -> |ani_al| (model.py:1287-1289) ----------------
        {
            const float S = RADD(esum, 1e-8f);
            float ev[VMAX], dot_du = 0.0f;
            int imin = 0;
            for (int v = 0; v < V; ++v) {
                ev[v] = kpn_fast_exp(RMUL(ani, RSUB(dotv[v], 1.0f)));
                if (ev[v] < ev[imin]) imin = v;
                if ((keep >> v) & 1u) dot_du += dwv[v] * (ev[v] - emin);
            }
            // d e_v first (the argmin view collects -sum of the others: a difference of nearly equal numbers when one
            // kept view carries almost all the weight), then the common factor — the reference's association
            float de[VMAX], demin = 0.0f;
#pragma unroll
            for (int v = 0; v < VMAX; ++v) de[v] = 0.0f;
            for (int v = 0; v < V; ++v) {
                if (!((keep >> v) & 1u)) continue;
                const float du = dwv[v] / S - dot_du / (S * S);
                de[v] = du;
                demin -= du;
            }
            float dabs = 0.0f;
            for (int v = 0; v < V; ++v) {
                const float dev = de[v] + (v == imin ? demin : 0.0f);
                dabs += dev * ev[v] * (dotv[v] - 1.0f);
            }
            dabs = (h == 0) ? dabs * live * wf[kpn_scalar_off() + 3] : 0.0f;  // d|a|/da = sign(ani_al)
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) dabs += __shfl_xor(dabs, m);
            dani_acc += dabs;                  // one atomic per wave at the end of the kernel, not one per tile on one address
        }
    }
    if (lane == 0 && dani_acc != 0.0f) kpn_atomic_add(B.dani, dani_acc);
}
