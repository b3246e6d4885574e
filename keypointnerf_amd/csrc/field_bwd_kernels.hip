// field_bwd_kernels.hip — reverse pass of the per-(point,view) geometry rows, i.e. of k_geo_rows:
// MLPUNet.layers1 232->128->128->(+8)120->64 (reference src/utils.py:691-716) and the bilinear feature
// gathers that feed it (feat_sample of feat_geo[0], feat_geo[1]; src/model.py:763-765, src/utils.py:74-89).
// This is the dominant ~70 % of the field evaluation's arithmetic; the reference gets it from autograd.
//
//   k_geo_rows_bwd : per row tile (32 rows of one view) —
//       F phase  recompute layers1.0 .. layers1.2 exactly as k_geo_rows does (activation checkpointing:
//                nothing is kept from the forward pass), writing each layer's INPUT row-major
//                (X0 [enc168|geo64], X1, X2 [128|hd8], X3) for the weight-gradient kernel;
//       B phase  dX = W^T dY on the matrix cores with the transposed segments (kpn_common.h BSEG_*),
//                chained through registers like the forward pass; softplus' = 1 - exp(-100 softplus) is
//                taken from the X dumps; each dA (= dY of a Linear) is written row-major;
//       scatter  d(geo0 64 ch) and d(geo1 8 ch) go back to the channels-last maps through the 4
//                bilinear taps with hardware float atomics.
//   k_weight_grad  : dW[o][f] += sum_rows dY[row][o] X[row][f], db[o] += sum_rows dY[row][o] — the
//                reduction over rows is the K dimension of v_mfma_f32_32x32x2_f32; row-major dumps are
//                exactly its A/B operand layout (lane = feature, k = row parity), no transposes.
// The keypoint encoding has no upstream parameters (points and keypoints are inputs), so dX0's first
// 168 entries are never formed.
#include "kpn_device.h"

struct kpn_bwd_bufs {
    float* X0;  // [rows][232]: cols 14j+7h+t = PE block t of keypoint 12h+j; cols 168..231 geometry channels
    float* X1;  // [rows][128]
    float* X2;  // [rows][136]
    float* X3;  // [rows][128]  (cols >= 120 unused)
    float* D0;  // dA0 [rows][128]
    float* D1;  // dA1 [rows][128]
    float* D2;  // dA2 [rows][128] (cols >= 120 are 0)
    float* D3;  // dY3 [rows][64]
    float* dgeo0;  // V x g0h x g0w x 64, accumulated
    float* dgeo1;  // V x g1h x g1w x 8, accumulated
};
// The dumps are written once and read once by a later kernel (k_weight_grad; 2.5 GB per training iteration from this kernel alone):
// streaming stores (global_store ... nt) keep them from displacing the weight streams and the feature maps in L2.  Measured:
// backward 4.98 -> 4.80 ms; without any dump stores (KPN_ABLATE_DUMP, scripts/experiments/ablation_switches.patch) 4.48 ms.  -DKPN_DUMP_NT=0 restores plain stores.
#ifndef KPN_DUMP_NT
#define KPN_DUMP_NT 1
#endif
#if KPN_DUMP_NT && !defined(KPN_SIMT_EMU)
typedef float kpn_nt4 __attribute__((ext_vector_type(4)));
#define KPN_ST4(p, v) do { const float4 v_ = (v); kpn_nt4 n_; n_[0] = v_.x; n_[1] = v_.y; n_[2] = v_.z; n_[3] = v_.w; __builtin_nontemporal_store(n_, reinterpret_cast<kpn_nt4*>(p)); } while (0)
#else
#define KPN_ST4(p, v) (*reinterpret_cast<float4*>(p) = (v))
#endif
#define KPN_ST1(p, v) (*(p) = (v))
#define KPN_STAGE_LD 132  // 128 floats + pad: rows start on different banks, float4-aligned
#define KPN_SCAT_LD KPN_STAGE_LD
#ifndef KPN_BWD_OCC
#define KPN_BWD_OCC 2
#endif
#ifndef KPN_BWD_LOCKSTEP
#define KPN_BWD_LOCKSTEP 1
#endif
#ifndef KPN_SIMT_EMU
#define KPN_BLOCK_MEET() __builtin_amdgcn_s_barrier()   // execution only: no memory is exchanged at these points
#else
#define KPN_BLOCK_MEET() __syncthreads()
#endif
#define KPN_LDX0 232
#define KPN_LDX2 136

// The nine Linear layers of the kernel below run on v_mfma_f32_32x32x16_bf16 with three bf16 pieces per operand (kpn_blayer, the
// BH_* streams: 2.7x less matrix time than the fp32 pipe, fp32-class results in fp32's exponent range — gradients included);
// -DKPN_BWD_F32 builds the fp32-MFMA form (kpn_mfma_layer on the fp32 streams) for A/B measurements.
template <int BH, int KS, int NOB, int G, class InFn>
__device__ __forceinline__ void kpn_bwd_layer(const float* __restrict__ wp, int segoff, int lane, InFn&& fn, kpn_f32x16 (&acc)[NOB]) {
    static_assert(KS == kpn_bh_shape(BH).ks && NOB == kpn_bh_shape(BH).nob && G == kpn_bh_shape(BH).g, "segment shape");
#ifdef KPN_BWD_F32
    kpn_mfma_layer<KS, NOB, G>(wp + segoff, lane, fn, acc);
#else
    (void)segoff;
    kpn_blayer<BH>(wp, lane, fn, acc);
#endif
}
// d softplus100(a) / da from the softplus value itself: sigmoid(100 a) = 1 - exp(-100 softplus(a)); above the
// threshold (softplus == a, 100 a > 20) this is 1 to within 2e-9, the reference's exact 1
__device__ __forceinline__ float kpn_softplus100_grad_from_value(float sp) { return 1.0f - kpn_exp2(sp * -144.269504088896341f); }

#ifdef KPN_BWD_TIMING   // debug builds: cycles (s_memtime) per phase of the tiles of one wave, summed (scripts/bwd_timing.py)
__device__ unsigned long long kpn_bwd_cycles[16];
#define KPN_BWD_STAMP(i) do { const unsigned long long now_ = clock64(); if (blockIdx.x == 3 && threadIdx.x == 64) atomicAdd(&kpn_bwd_cycles[i], now_ - stamp_); stamp_ = now_; } while (0)
#else
#define KPN_BWD_STAMP(i) ((void)0)
#endif
// dx: upstream gradient d loss / d x_view: [N][V][64] indexed by the ORIGINAL point index, or (dx_compact) by
// row = (tile*V + v)*32 + p as k_fuse_bwd writes it
__global__ __launch_bounds__(256, KPN_BWD_OCC) void k_geo_rows_bwd(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                                         const int* __restrict__ list, const int* __restrict__ count_ptr,
                                                         int* __restrict__ tickets, const float* __restrict__ dx,
                                                         int dx_compact, kpn_bwd_bufs bufs) {
    const int lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    const int count = *count_ptr;
    const int ntiles = (count + KPN_TILE - 1) / KPN_TILE;
    const int nwork = ntiles * sc.V;
    __shared__ __attribute__((aligned(16))) float bias_s[3][128];
    // Per wave: 32 rows x up to 128 features (+ pad).  Every dump below is staged here and written out with the LANES ALONG THE
    // FEATURES (one instruction = two whole 512-B rows): from the registers a lane holds ONE row, and a store instruction issued
    // from that layout touches 64 different cache lines with 4 or 16 bytes each — measured (PMC, profiles/r03_j_train_pmc_counters.txt)
    // 7.1 KB written per row for 4.3 KB of dumps and a kernel at 27 % matrix-pipe busy.  The same buffer is the transposition
    // buffer of the feature-map scatter at the end of a tile.
    __shared__ __attribute__((aligned(16))) float stage_s[4][KPN_TILE][KPN_STAGE_LD];
    __shared__ __attribute__((aligned(16))) int4 tap_o[4][2][KPN_TILE];
    __shared__ __attribute__((aligned(16))) float4 tap_w[4][2][KPN_TILE];
    {
        const int segs[3] = {SEG_G1_0A, SEG_G1_1, SEG_G1_2};
        for (int i = threadIdx.x; i < 3 * 128; i += blockDim.x) bias_s[i >> 7][i & 127] = wp[kpn_seg_boff(segs[i >> 7]) + (i & 127)];
    }
    __syncthreads();

#ifdef KPN_BWD_TIMING
    unsigned long long stamp_ = clock64();
#endif
#if KPN_BWD_LOCKSTEP
    __shared__ int ticket_s;
#endif
    for (;;) {
        KPN_BWD_STAMP(11);
#if KPN_BWD_LOCKSTEP
        // The four waves of a workgroup take four consecutive work items and meet at a barrier before every layer: they then ask for
        // the same weight lines within a few hundred cycles of each other and three of the four requests are served by the CU's L1
        // (or merged with the miss in flight) instead of by L2 — every wave streams 708 KB of weights per work item.  A wave whose
        // item lies beyond the end redoes the last one with nothing live and nothing stored (at most three per workgroup).
        __syncthreads();
        if (threadIdx.x == 0) ticket_s = atomicAdd(tickets, 4);
        __syncthreads();
        const int wbase = ticket_s;
        if (wbase >= nwork) break;
        const bool item_ok = wbase + (int)(threadIdx.x >> 6) < nwork;
        const int wi = item_ok ? wbase + (int)(threadIdx.x >> 6) : nwork - 1;
#define KPN_BWD_MEET() KPN_BLOCK_MEET()
#else
        int wi = 0;
        if (lane == 0) wi = atomicAdd(tickets, 1);
        wi = __shfl(wi, 0);
        if (wi >= nwork) break;
        const bool item_ok = true;
#define KPN_BWD_MEET() ((void)0)
#endif
        const int t = wi / sc.V, v = wi - t * sc.V;
        int ci = t * KPN_TILE + p;
        // pad lanes recompute the last point with a zero upstream gradient; so do views switched off by the
        // train-time dropout (their pooling weight is 0, model.py:748)
        const float live = (item_ok && ci < count && ((sc.keep >> v) & 1u)) ? 1.0f : 0.0f;
        if (ci >= count) ci = count - 1;
        const int64_t n = list[ci];
        const size_t row = (size_t)wi * KPN_TILE + p;
        float P[3], D[3];
        kpn_get_point(ps, n, P, D);
        const float* tb = sc.table + (size_t)v * KPN_TBL_STRIDE;
        const kpn_proj q = kpn_project(tb, P[0], P[1], P[2], sc);
        const kpn_taps tp0 = kpn_make_taps(q.xn, q.yn, sc.g0h, sc.g0w);
        const kpn_taps tp1 = kpn_make_taps(q.xn, q.yn, sc.g1h, sc.g1w);
        float* const x1row = bufs.X1 + row * 128;
        float* const x2row = bufs.X2 + row * KPN_LDX2;
        float* const x3row = bufs.X3 + row * 128;
        float* const stg = stage_s[threadIdx.x >> 6][0];         // this wave's staging tile
        float* const srow = stg + p * KPN_STAGE_LD;              // this lane's row of it
        const size_t row0 = (size_t)wi * KPN_TILE;               // first row of the tile in the dumps
        // write columns [0, ncols) of the staged tile to dump[(row0 + r) * ld + col0 + c]; ncols a multiple of 4
        auto flush = [&](float* __restrict__ dump, int ld, int col0, int ncols) {
            KPN_WAVE_SYNC();
            const int q = ncols >> 2;                             // float4 per row
            for (int i = lane; item_ok && i < KPN_TILE * q; i += 64) {
                const int r = i / q, c = (i - r * q) << 2;
                KPN_ST4(dump + (row0 + r) * ld + col0 + c, *reinterpret_cast<const float4*>(stg + r * KPN_STAGE_LD + c));
            }
            KPN_WAVE_SYNC();
        };

        KPN_BWD_STAMP(0);
        // The B phase re-reads its rows (upstream gradient, X3, X2, X1) one chained group per call; fetched where it is used, every
        // group exposed a whole L2 round trip (two per chunk of the layer): a two-deep ring fetches group g + 2 when g is consumed.
        float4 pf[2];
        auto row4 = [&](const float* rowp, int g) { return *reinterpret_cast<const float4*>(rowp + 32 * (g / 4) + 8 * (g % 4) + 4 * h); };
        // ================= F phase (same arithmetic as k_geo_rows) =================
        kpn_f32x16 a0[4];
        {
            const float* E = tb + KPN_TBL_EXT;
            const float cx = RADD(kpn_dot3(P[0], P[1], P[2], E[0], E[1], E[2]), E[3]);
            const float cy = RADD(kpn_dot3(P[0], P[1], P[2], E[4], E[5], E[6]), E[7]);
            const float cz = RADD(kpn_dot3(P[0], P[1], P[2], E[8], E[9], E[10]), E[11]);
            const float* kc = tb + KPN_TBL_KCAM + (12 * h) * 3;
            kpn_load_bias<4>(bias_s[0], h, a0);
            KPN_BWD_MEET();
            kpn_bwd_layer<BH_G1_0A, 84, 4, 7>(wp, kpn_seg_woff(SEG_G1_0A), lane, [&](auto gi, float (&x)[7]) {
                constexpr int j = decltype(gi)::value;
                const float dx_ = RSUB(cx, kc[j * 3 + 0]), dy = RSUB(cy, kc[j * 3 + 1]), dz = RSUB(cz, kc[j * 3 + 2]);
                const float d2 = RADD(RADD(RMUL(dx_, dx_), RMUL(dy, dy)), RMUL(dz, dz));
                const float w = kpn_fast_exp(-d2 / sc.two_sigma2);
                float s1, c1;
                kpn_sincos_pi(dz, s1, c1);
                const float s2 = 2.0f * s1 * c1, c2 = 1.0f - 2.0f * s1 * s1;
                const float s4 = 2.0f * s2 * c2, c4 = 1.0f - 2.0f * s2 * s2;
                x[0] = dz * w;
                x[1] = s1 * w; x[2] = c1 * w;
                x[3] = s2 * w; x[4] = c2 * w;
                x[5] = s4 * w; x[6] = c4 * w;
                // X0's encoding columns are keypoint-major: column 14 j + 7 h + t (kpn_grad_col, cmap 1): the values of six
                // keypoint steps (both halves) are 84 adjacent columns = one staged flush
                float* d = srow + (j % 6) * 14 + 7 * h;
#pragma unroll
                for (int i = 0; i < 7; ++i) d[i] = x[i];
                if constexpr (j == 5 || j == 11) flush(bufs.X0, KPN_LDX0, 84 * (j / 6), 84);
            }, a0);
            KPN_BWD_STAMP(1);
            const float* g0 = sc.geo0 + (size_t)v * sc.g0h * sc.g0w * 64;
            KPN_BWD_MEET();
            kpn_bwd_layer<BH_G1_0B, 32, 4, 4>(wp, kpn_seg_woff(SEG_G1_0B), lane, [&](auto gi, float (&x)[4]) {
                constexpr int g = decltype(gi)::value;
                const float4 f = kpn_tap4(g0, 64, 32 * h + 4 * g, tp0);
                x[0] = f.x; x[1] = f.y; x[2] = f.z; x[3] = f.w;
                *reinterpret_cast<float4*>(srow + 32 * h + 4 * g) = f;
            }, a0);
            flush(bufs.X0, KPN_LDX0, 168, 64);
        }
        KPN_BWD_STAMP(2);
        // chained group g of a 128-vector = features 32(g/4) + 8(g%4) + 4h .. +3 of this lane's row
        kpn_f32x16 a1[4];
        kpn_load_bias<4>(bias_s[1], h, a1);
        KPN_BWD_MEET();
        kpn_bwd_layer<BH_G1_1, 64, 4, 4>(wp, kpn_seg_woff(SEG_G1_1), lane, [&](auto gi, float (&x)[4]) {
            constexpr int g = decltype(gi)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = kpn_softplus100(a0[g / 4][(g % 4) * 4 + i]);
            *reinterpret_cast<float4*>(srow + 32 * (g / 4) + 8 * (g % 4) + 4 * h) = make_float4(x[0], x[1], x[2], x[3]);
        }, a1);
        flush(bufs.X1, 128, 0, 128);
        KPN_BWD_STAMP(3);
        kpn_f32x16 a2[4];
        {
            const float4 f = kpn_tap4(sc.geo1 + (size_t)v * sc.g1h * sc.g1w * 8, 8, 4 * h, tp1);
            if (item_ok) KPN_ST4(x2row + 128 + 4 * h, f);   // (8 of 136 columns: left as a per-row store)
            kpn_load_bias<4>(bias_s[2], h, a2);
            KPN_BWD_MEET();
            kpn_bwd_layer<BH_G1_2, 68, 4, 4>(wp, kpn_seg_woff(SEG_G1_2), lane, [&](auto gi, float (&x)[4]) {
                constexpr int g = decltype(gi)::value;
                if constexpr (g < 16) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) x[i] = kpn_softplus100(a1[g / 4][(g % 4) * 4 + i]);
                    *reinterpret_cast<float4*>(srow + 32 * (g / 4) + 8 * (g % 4) + 4 * h) = make_float4(x[0], x[1], x[2], x[3]);
                } else {
                    x[0] = f.x; x[1] = f.y; x[2] = f.z; x[3] = f.w;
                }
            }, a2);
            flush(bufs.X2, KPN_LDX2, 0, 128);
        }
        KPN_BWD_STAMP(4);
        // X3 = softplus(a2), the input of layers1.3 (its forward product itself is not needed here)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            float4 o;
            o.x = kpn_softplus100(a2[g / 4][(g % 4) * 4 + 0]); o.y = kpn_softplus100(a2[g / 4][(g % 4) * 4 + 1]);
            o.z = kpn_softplus100(a2[g / 4][(g % 4) * 4 + 2]); o.w = kpn_softplus100(a2[g / 4][(g % 4) * 4 + 3]);
            *reinterpret_cast<float4*>(srow + 32 * (g / 4) + 8 * (g % 4) + 4 * h) = o;
        }
        flush(bufs.X3, 128, 0, 128);
        KPN_BWD_STAMP(5);

        // ================= B phase =================
        // dX3 = W3^T dY3
        kpn_f32x16 d3[4];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) d3[ob][r] = 0.0f;
        {
            const float* grow = dx_compact ? dx + row * 64 : dx + ((size_t)n * sc.V + v) * 64;
            pf[0] = row4(grow, 0); pf[1] = row4(grow, 1);
            KPN_BWD_MEET();
            kpn_bwd_layer<BH_G1_3T, 32, 4, 4>(wp, kpn_bseg_woff(BSEG_G1_3T), lane, [&](auto gi, float (&x)[4]) {
                constexpr int g = decltype(gi)::value;
                const int col = 32 * (g / 4) + 8 * (g % 4) + 4 * h;
                const float4 f = pf[g & 1];
                if constexpr (g + 2 < 8) pf[g & 1] = row4(grow, g + 2);
                x[0] = f.x * live; x[1] = f.y * live; x[2] = f.z * live; x[3] = f.w * live;
                *reinterpret_cast<float4*>(srow + col) = make_float4(x[0], x[1], x[2], x[3]);
            }, d3);
            flush(bufs.D3, 64, 0, 64);
        }
        KPN_BWD_STAMP(6);
        // dA2 = dX3 * softplus'(a2);  [dX2 | d hd] = W2^T dA2
        kpn_f32x16 d2[5];
#pragma unroll
        for (int ob = 0; ob < 5; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) d2[ob][r] = 0.0f;
        {
            pf[0] = row4(x3row, 0); pf[1] = row4(x3row, 1);
            KPN_BWD_MEET();
            kpn_bwd_layer<BH_G1_2T, 64, 5, 4>(wp, kpn_bseg_woff(BSEG_G1_2T), lane, [&](auto gi, float (&x)[4]) {
                constexpr int g = decltype(gi)::value;
                const int col = 32 * (g / 4) + 8 * (g % 4) + 4 * h;
                const float4 s = pf[g & 1];
                if constexpr (g + 2 < 16) pf[g & 1] = row4(x3row, g + 2);
                x[0] = d3[g / 4][(g % 4) * 4 + 0] * kpn_softplus100_grad_from_value(s.x);
                x[1] = d3[g / 4][(g % 4) * 4 + 1] * kpn_softplus100_grad_from_value(s.y);
                x[2] = d3[g / 4][(g % 4) * 4 + 2] * kpn_softplus100_grad_from_value(s.z);
                x[3] = d3[g / 4][(g % 4) * 4 + 3] * kpn_softplus100_grad_from_value(s.w);
                *reinterpret_cast<float4*>(srow + col) = make_float4(x[0], x[1], x[2], x[3]);
            }, d2);
            flush(bufs.D2, 128, 0, 128);
        }
        KPN_BWD_STAMP(7);
        // the 8 hd channels (rows 128..135 of dX2 = block 4 regs 0..3: channels 4h..4h+3) go back to feat_geo[1]
        const float4 dhd = make_float4(d2[4][0], d2[4][1], d2[4][2], d2[4][3]);
        // dA1 = dX2 * softplus'(a1);  dX1 = W1^T dA1
        kpn_f32x16 d1[4];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) d1[ob][r] = 0.0f;
        {
            pf[0] = row4(x2row, 0); pf[1] = row4(x2row, 1);
            KPN_BWD_MEET();
            kpn_bwd_layer<BH_G1_1T, 64, 4, 4>(wp, kpn_bseg_woff(BSEG_G1_1T), lane, [&](auto gi, float (&x)[4]) {
                constexpr int g = decltype(gi)::value;
                const int col = 32 * (g / 4) + 8 * (g % 4) + 4 * h;
                const float4 s = pf[g & 1];
                if constexpr (g + 2 < 16) pf[g & 1] = row4(x2row, g + 2);
                x[0] = d2[g / 4][(g % 4) * 4 + 0] * kpn_softplus100_grad_from_value(s.x);
                x[1] = d2[g / 4][(g % 4) * 4 + 1] * kpn_softplus100_grad_from_value(s.y);
                x[2] = d2[g / 4][(g % 4) * 4 + 2] * kpn_softplus100_grad_from_value(s.z);
                x[3] = d2[g / 4][(g % 4) * 4 + 3] * kpn_softplus100_grad_from_value(s.w);
                *reinterpret_cast<float4*>(srow + col) = make_float4(x[0], x[1], x[2], x[3]);
            }, d1);
            flush(bufs.D1, 128, 0, 128);
        }
        KPN_BWD_STAMP(8);
        // dA0 = dX1 * softplus'(a0);  d geo0 = W0[:,168:232]^T dA0
        kpn_f32x16 dg[2];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) dg[ob][r] = 0.0f;
        {
            pf[0] = row4(x1row, 0); pf[1] = row4(x1row, 1);
            KPN_BWD_MEET();
            kpn_bwd_layer<BH_G1_0T, 64, 2, 4>(wp, kpn_bseg_woff(BSEG_G1_0T), lane, [&](auto gi, float (&x)[4]) {
                constexpr int g = decltype(gi)::value;
                const int col = 32 * (g / 4) + 8 * (g % 4) + 4 * h;
                const float4 s = pf[g & 1];
                if constexpr (g + 2 < 16) pf[g & 1] = row4(x1row, g + 2);
                x[0] = d1[g / 4][(g % 4) * 4 + 0] * kpn_softplus100_grad_from_value(s.x);
                x[1] = d1[g / 4][(g % 4) * 4 + 1] * kpn_softplus100_grad_from_value(s.y);
                x[2] = d1[g / 4][(g % 4) * 4 + 2] * kpn_softplus100_grad_from_value(s.z);
                x[3] = d1[g / 4][(g % 4) * 4 + 3] * kpn_softplus100_grad_from_value(s.w);
                *reinterpret_cast<float4*>(srow + col) = make_float4(x[0], x[1], x[2], x[3]);
            }, dg);
            flush(bufs.D0, 128, 0, 128);
        }
        KPN_BWD_STAMP(9);
        // Scatter through the bilinear taps.  A lane holds 32 channels of ONE row; issued from this layout an atomic
        // instruction would touch 32 different texels (cache lines).  The tile is transposed through LDS instead:
        // lane = channel, so that one instruction adds 64 consecutive floats (two lines) of one texel.
        {
            const int w4 = threadIdx.x >> 6;
            float* sg = stg;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
                    *reinterpret_cast<float4*>(sg + p * KPN_SCAT_LD + 32 * b + 8 * qd + 4 * h) =
                        make_float4(dg[b][4 * qd + 0], dg[b][4 * qd + 1], dg[b][4 * qd + 2], dg[b][4 * qd + 3]);
            *reinterpret_cast<float4*>(sg + p * KPN_SCAT_LD + 64 + 4 * h) = dhd;
            if (h == 1) {   // (the 64-channel map's taps stay in registers: kpn_scatter_rle64_regs)
                tap_o[w4][1][p] = make_int4(tp1.o00, tp1.o01, tp1.o10, tp1.o11);
                tap_w[w4][1][p] = make_float4(tp1.w00 * live, tp1.w01 * live, tp1.w10 * live, tp1.w11 * live);
            }
            KPN_WAVE_SYNC();
            const int npt = min(KPN_TILE, count - t * KPN_TILE);
            // one atomic per run of equal texel offsets and tap (kpn_scatter_rle*, kpn_device.h), not one per point
            {
                const int o0[4] = {tp0.o00, tp0.o01, tp0.o10, tp0.o11};
                const float w0[4] = {tp0.w00 * live, tp0.w01 * live, tp0.w10 * live, tp0.w11 * live};
                kpn_scatter_rle64_regs(bufs.dgeo0 + (size_t)v * sc.g0h * sc.g0w * 64, sg, KPN_SCAT_LD, o0, w0, lane);
            }
            kpn_scatter_rle8(bufs.dgeo1 + (size_t)v * sc.g1h * sc.g1w * 8, sg + 64, KPN_SCAT_LD, tap_o[w4][1], tap_w[w4][1], npt, lane);
            KPN_WAVE_SYNC();  // the next tile overwrites the exchange buffers
        }
        KPN_BWD_STAMP(10);
    }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient  dW[o][colmap(c)] = sum_row dY[row][o] * X[row][c],  db[o] = sum_row dY[row][o].
// The sum over rows is the K dimension of the MFMA: lane l supplies row 2s + (l>>5) of K-step s.  A lane
// loads MV consecutive features of dY (16 B for MV = 4) and 2 consecutive columns of X per K-step and feeds
// MV x 2 tiles: tile (a, b) holds output feature MV*i + a (A lane i) x column c0 + 2*j + b (B lane j) — the
// strided labelling makes the vector loads the operands of several MFMAs at once (8 MFMAs per 24 bytes).
// One launch serves several layers ("jobs", blockIdx.y): the small layers of the colour head would each fill a
// fraction of the chip on their own.  grid (row workers, jobs); a workgroup = one row slice of one job, its waves =
// the job's 64-column groups (dY is fetched from HBM once).  Every wave writes its partial tile block in raw
// register order; k_weight_grad_reduce sums the workers in a fixed order (no atomics: bit-reproducible for a given row order,
// i.e. for a given valid list — scripts/soak_backward.py).
//   partial: [column groups][workers][MV*2*16][64 lanes];  dbp: [workers][MV][64 lanes]
#define KPN_WGRAD_MAX_JOBS 20
struct kpn_wgrad_job {
    const float* dY; const float* X;   // row-major dumps
    float* partial; float* dbp;
    float* dW; float* dB;              // plain (out, in) layout + bias, accumulated by the reduce
    int ldy, M, ldx, Kc;               // Kc: columns of X read (even)
    int Kt, in_dim, cmap, omap;        // reduce: real columns, row stride of dW, column / row maps
    int mv, which;                     // MV of the job (reduce), rows[which] = row count
    int olab;                          // labelling of the partial tiles' output features: 0 = MV i + a (k_weight_grad_f32), 1 = 32 a + i
    int V; uint32_t keep;              // which == 0: row r belongs to view (r / 32) % V; rows of views whose keep bit is 0 (train-time
                                       // view dropout) count as zeros — their dumps are never written
};
struct kpn_wgrad_jobs { kpn_wgrad_job j[KPN_WGRAD_MAX_JOBS]; int n; };

template <int MV>
__global__ __launch_bounds__(256) void k_weight_grad_f32(kpn_wgrad_jobs jobs, const int64_t* __restrict__ rows_ptr) {
    const kpn_wgrad_job& J = jobs.j[blockIdx.y];
    const int z = threadIdx.x >> 6;
    if (64 * z >= J.Kc) return;  // this job has fewer column groups than the launch's widest
    const float* __restrict__ dY = J.dY;
    const float* __restrict__ X = J.X;
    const int ldy = J.ldy, M = J.M, ldx = J.ldx, Kc = J.Kc;
    const int64_t rows = rows_ptr[J.which];
    const int lane = threadIdx.x & 63, i = lane & 31, kk = lane >> 5;
    const int worker = blockIdx.x, nworkers = gridDim.x;
    const int c0 = z * 64;
    const int64_t npairs = rows / 2;
    int64_t per = (npairs + nworkers - 1) / nworkers;
    per = (per + 3) / 4 * 4;  // whole groups of 4 K-steps
    const int64_t pbeg = (int64_t)worker * per;
    const int64_t pend = pbeg + per < npairs ? pbeg + per : npairs;
    kpn_f32x16 acc[MV][2];
#pragma unroll
    for (int a = 0; a < MV; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float bs[MV];
#pragma unroll
    for (int a = 0; a < MV; ++a) bs[a] = 0.0f;
    const bool cok = c0 + 2 * i < Kc;  // Kc is even: both columns of the pair are in or out
    constexpr int U = 4;               // K-steps per software-pipeline stage
    float ya[2][U][MV];
    float2 xb[2][U];
    auto fetch = [&](int64_t pr, float (&y)[U][MV], float2 (&x)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = 2 * (pr + u) + kk;
            const bool in = pr + u < pend && ((J.keep >> ((uint32_t)(r >> 5) % (uint32_t)J.V)) & 1u);
            const bool yin = in && MV * i < M;  // M is a multiple of MV
            if constexpr (MV == 4) {
                float4 y4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (yin) y4 = *reinterpret_cast<const float4*>(dY + r * ldy + 4 * i);
                y[u][0] = y4.x; y[u][1] = y4.y; y[u][2] = y4.z; y[u][3] = y4.w;
            } else if constexpr (MV == 2) {
                float2 y2 = make_float2(0.f, 0.f);
                if (yin) y2 = *reinterpret_cast<const float2*>(dY + r * ldy + 2 * i);
                y[u][0] = y2.x; y[u][1] = y2.y;
            } else {
                y[u][0] = yin ? dY[r * ldy + i] : 0.0f;
            }
            x[u] = make_float2(0.f, 0.f);
            if (in && cok) x[u] = *reinterpret_cast<const float2*>(X + r * ldx + c0 + 2 * i);
        }
    };
    auto consume = [&](const float (&y)[U][MV], const float2 (&x)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int a = 0; a < MV; ++a) {
                bs[a] += y[u][a];
                acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(y[u][a], x[u].x, acc[a][0], 0, 0, 0);
                acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(y[u][a], x[u].y, acc[a][1], 0, 0, 0);
            }
    };
    if (pbeg < pend) {
        fetch(pbeg, ya[0], xb[0]);
        for (int64_t pr = pbeg; pr < pend; pr += 2 * U) {
            fetch(pr + U, ya[1], xb[1]);   // (reads nothing past pend)
            consume(ya[0], xb[0]);
            fetch(pr + 2 * U, ya[0], xb[0]);
            consume(ya[1], xb[1]);
        }
    }
    float* dst = J.partial + ((size_t)z * nworkers + worker) * (MV * 2 * 16 * 64) + lane;
#pragma unroll
    for (int a = 0; a < MV; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[((a * 2 + b) * 16 + r) * 64] = acc[a][b][r];
    if (z == 0) {
#pragma unroll
        for (int a = 0; a < MV; ++a) J.dbp[((size_t)worker * MV + a) * 64 + lane] = bs[a];
    }
}

// The default form (round 3): the same sum on v_mfma_f32_32x32x16_bf16 with every operand as three bf16 pieces and six products
// (fp32-class, fp32's exponent range — gradients span too many decades for fp16 pieces): K = 16 rows per step, lane l supplies
// rows 8 (l >> 5) + e of a step.  2.7x less matrix time than the fp32 form (48 MFMAs of 32 cycles per 16 rows against 64 of 64).
//
// With the matrix time that small the kernel is bound by what feeds it, so the feeding is shared: a value of dY is split ONCE per
// workgroup (the fp32 form and the first bf16 form split it in every wave, i.e. up to four times) — the wave that owns output
// block a loads dY[16 rows][32 a + i], splits it and parks the three pieces in LDS in operand order; after one barrier every
// wave reads the MV blocks back (ds_read_b128) for its own 64 columns, whose X pieces never leave its registers.  Two LDS
// stages: a stage is rewritten two steps later, behind the next step's barrier.  Waves beyond the job's column groups own
// output blocks before the others do (they have nothing else to do).  Loads are unconditional: the dumps are whole tiles
// (rows % 32 == 0), a lane beyond M or Kc re-reads the last valid feature / column pair (its tile rows / columns are never
// read by the reduce), and the steps of a dropped view (never written) are skipped as a whole.  The barrier is s_barrier behind
// s_waitcnt lgkmcnt(0) only: __syncthreads() would also drain the global loads of the next step, which are the point.
//   labelling: tile (a, b) holds output feature 32 a + i (A lane i) x column c0 + 2 j + b (B lane j)   [olab = 1]
#ifndef KPN_SIMT_EMU
#define KPN_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define KPN_LDS_BARRIER() __syncthreads()
#endif
template <int MV>
__global__ __launch_bounds__(256, 2) void k_weight_grad(kpn_wgrad_jobs jobs, const int64_t* __restrict__ rows_ptr) {
    const kpn_wgrad_job& J = jobs.j[blockIdx.y];
    const int z = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int G = (J.Kc + 63) >> 6;            // column groups of this job: waves z < G multiply, every wave may split
    const bool mul = z < G;
    const int ldy = J.ldy, M = J.M, ldx = J.ldx, Kc = J.Kc;
    const int64_t rows = rows_ptr[J.which];
    const int lane = threadIdx.x & 63, i = lane & 31, kk = lane >> 5;
    const int worker = blockIdx.x, nworkers = gridDim.x;
    const int64_t nchunks = rows / 16;
    const int64_t per = (nchunks + nworkers - 1) / nworkers;
    const int64_t cbeg = (int64_t)worker * per;
    const int64_t cend = cbeg + per < nchunks ? cbeg + per : nchunks;
    __shared__ __attribute__((aligned(16))) kpn_u32x4 a_s[2][MV][3][64];
    bool own[MV];                              // wave-uniform: this wave splits output block a
    uint32_t fo4[MV];                          // byte offset of (row 8 kk, feature 32 a + i, clamped) in a step of dY
#pragma unroll
    for (int a = 0; a < MV; ++a) {
        own[a] = (nw > G ? G + a % (nw - G) : a % nw) == z;
        const int f = 32 * a + i;
        fo4[a] = 4u * (uint32_t)(8 * kk * ldy + (f < M ? f : M - 1));
    }
    // (row 8 kk, column pair 64 z + 2 i, clamped; Kc is even) in a step of X; row e of the eight adds e * ld * 4 on the scalar side
    const uint32_t xo4 = 4u * (uint32_t)(8 * kk * ldx + (mul ? min(64 * z + 2 * i, Kc - 2) : 0));
    const uint32_t ldy4 = 4u * (uint32_t)ldy, ldx4 = 4u * (uint32_t)ldx;
    kpn_f32x16 acc[MV][2];
#pragma unroll
    for (int a = 0; a < MV; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float bs[MV];
#pragma unroll
    for (int a = 0; a < MV; ++a) bs[a] = 0.0f;
    float ya[MV][8];
    float2 xb[8];
    // the 16 rows of a step lie in one 32-row (tile, view) block
    auto kept = [&](int64_t c) { return ((J.keep >> ((uint32_t)(c >> 1) % (uint32_t)J.V)) & 1u) != 0; };
    auto next_kept = [&](int64_t c) { while (c < cend && !kept(c)) ++c; return c; };
    auto fetch = [&](int64_t c) {
        const char* yb = reinterpret_cast<const char*>(J.dY + c * 16 * ldy);   // uniform
        const char* xp = reinterpret_cast<const char*>(J.X + c * 16 * ldx);
#pragma unroll
        for (int a = 0; a < MV; ++a)
            if (own[a]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) ya[a][e] = *reinterpret_cast<const float*>(yb + e * ldy4 + fo4[a]);
            }
        if (mul) {
#pragma unroll
            for (int e = 0; e < 8; ++e) xb[e] = *reinterpret_cast<const float2*>(xp + e * ldx4 + xo4);
        }
    };
    int64_t c = next_kept(cbeg);
    if (c < cend) fetch(c);
    int st = 0;
    while (c < cend) {
#pragma unroll
        for (int a = 0; a < MV; ++a)
            if (own[a]) {
                kpn_bf16x8 h, m, l;
                kpn_split_bf16x8<false>(ya[a], h, m, l);
#pragma unroll
                for (int e = 0; e < 8; ++e) bs[a] += ya[a][e];
                a_s[st][a][0][lane] = __builtin_bit_cast(kpn_u32x4, h);
                a_s[st][a][1][lane] = __builtin_bit_cast(kpn_u32x4, m);
                a_s[st][a][2][lane] = __builtin_bit_cast(kpn_u32x4, l);
            }
        kpn_bf16x8 bh[2], bm[2], bl[2];
        if (mul) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = b == 0 ? xb[e].x : xb[e].y;
                kpn_split_bf16x8<false>(v, bh[b], bm[b], bl[b]);
            }
        }
        const int64_t cn = next_kept(c + 1);
        if (cn < cend) fetch(cn);
        KPN_LDS_BARRIER();
        if (mul) {
#pragma unroll
            for (int a = 0; a < MV; ++a) {
                const kpn_bf16x8 ah = __builtin_bit_cast(kpn_bf16x8, a_s[st][a][0][lane]);
                const kpn_bf16x8 am = __builtin_bit_cast(kpn_bf16x8, a_s[st][a][1][lane]);
                const kpn_bf16x8 al = __builtin_bit_cast(kpn_bf16x8, a_s[st][a][2][lane]);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = KPN_MFMA16(ah, bh[b], acc[a][b]);
                    acc[a][b] = KPN_MFMA16(ah, bm[b], acc[a][b]);
                    acc[a][b] = KPN_MFMA16(am, bh[b], acc[a][b]);
                    acc[a][b] = KPN_MFMA16(am, bm[b], acc[a][b]);
                    acc[a][b] = KPN_MFMA16(ah, bl[b], acc[a][b]);
                    acc[a][b] = KPN_MFMA16(al, bh[b], acc[a][b]);
                }
            }
        }
        st ^= 1;
        c = cn;
    }
    if (mul) {
        float* dst = J.partial + ((size_t)z * nworkers + worker) * (MV * 2 * 16 * 64) + lane;
#pragma unroll
        for (int a = 0; a < MV; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[((a * 2 + b) * 16 + r) * 64] = acc[a][b][r];
    }
#pragma unroll
    for (int a = 0; a < MV; ++a)
        if (own[a]) J.dbp[((size_t)worker * MV + a) * 64 + lane] = bs[a];
}

// cmap: 0 = X columns are plain input features; 1 = X0 dump (first 168 columns in (keypoint, PE block) order);
//       2 = base_layer.0's [mean' | var' | x'] dump (3 x 36 columns, x' order).  omap: 1 = dY rows are in x' order.
__device__ __forceinline__ int kpn_grad_col(int cmap, int c) {
    if (cmap == 1) return c < 168 ? (c % 7) * 24 + 12 * ((c % 14) / 7) + c / 14 : c;   // X0 column 14 j + 7 h + t = PE block t of keypoint 12 h + j
    if (cmap == 2) { const int q = c % 36; return q < 35 ? (c / 36) * 35 + kpn_xprime_to_orig(q) : -1; }
    return c;
}
// Fixed-order sum of the workers' partial blocks, added to the plain-layout gradient (no atomics: the weight
// gradient is reproducible for a given row order).  grid (element groups of 32, column groups, jobs); a workgroup owns 32 consecutive
// elements; its 8 thread-octets stride over the workers and combine through LDS.
__global__ __launch_bounds__(256) void k_weight_grad_reduce(kpn_wgrad_jobs jobs, int nworkers) {
    const kpn_wgrad_job& J = jobs.j[blockIdx.z];
    const int MV = J.mv, M = J.M, Kc = J.Kt;
    const int TILE_E = MV * 2 * 16 * 64;
    __shared__ float red[8][32];
    const int el = threadIdx.x & 31, wl = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    const int z = blockIdx.y;
    if (64 * z >= J.Kc || blockIdx.x * 32 >= TILE_E + MV * 32) return;  // uniform per workgroup
    float s = 0.0f;
    if (e < TILE_E) {
        // 64 workers per thread: sixteen loads in flight, added in the fixed order (one load per add, as the loop it replaces,
        // exposed a whole HBM round trip per worker: 104 us per call however few workers there were)
        const float* src = J.partial + (size_t)z * nworkers * TILE_E + e;
        int w = wl;
        for (; w + 8 * 15 < nworkers; w += 8 * 16) {
            float a[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = src[(size_t)(w + 8 * k) * TILE_E];
#pragma unroll
            for (int k = 0; k < 16; ++k) s += a[k];
        }
        for (; w < nworkers; w += 8) s += src[(size_t)w * TILE_E];
    } else if (z == 0 && e < TILE_E + MV * 32) {
        const int q = e - TILE_E, a = q / 32, i = q % 32;
        for (int w = wl; w < nworkers; w += 8) s += J.dbp[((size_t)w * MV + a) * 64 + i] + J.dbp[((size_t)w * MV + a) * 64 + 32 + i];
    }
    red[wl][el] = s;
    __syncthreads();
    if (wl != 0) return;
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][el];
    if (e < TILE_E) {
        const int lane = e & 63, r = (e >> 6) & 15, ab = e >> 10, a = ab >> 1, b = ab & 1;
        int o = J.olab ? 32 * a + KPN_ROWMAP(r, lane >> 5) : MV * KPN_ROWMAP(r, lane >> 5) + a;
        const int c = z * 64 + 2 * (lane & 31) + b;
        if (o < M && c < Kc) {
            const int f = kpn_grad_col(J.cmap, c);
            if (J.omap) o = kpn_xprime_to_orig(o);
            if (f >= 0) J.dW[(size_t)o * J.in_dim + f] += s;
        }
    } else if (z == 0 && e < TILE_E + MV * 32) {
        const int q = e - TILE_E, a = q / 32, i = q % 32;
        const int o = J.olab ? 32 * a + i : MV * i + a;
        if (o < M) J.dB[J.omap ? kpn_xprime_to_orig(o) : o] += s;
    }
}
