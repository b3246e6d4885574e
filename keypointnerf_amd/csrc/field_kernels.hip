// field_kernels.hip — the field evaluation KeypointNeRF.query (reference src/model.py:690-843,
// 1239-1302; src/spatial.py:63-118; src/utils.py:476-748) as three gfx950 kernels:
//
//   k_mask_compact : per point — project into the V source views, validity + fg mask (AND over
//                    views, model.py:725-739), write the constant result of masked points and append
//                    valid points to a compact list (wave-aggregated atomic).
//   k_geo_rows     : [dominant, MFMA-bound] per (valid point, view) "row" — bilinear gather of the
//                    64+8 geometry channels, keypoint-relative encoding (168), MLPUNet layers1
//                    232->128->128->(+8)120->64 on v_mfma_f32_32x32x2_f32; writes the 64-vector.
//   k_fuse_color   : per valid point — view-weighted mean/var pooling, layers2 128->64->64->2,
//                    ibr_compress 128->24, IBR head over the V views, softmax blend of source colours.
//
// Work unit: a wavefront owns a tile of 32 valid points (the N dimension of the 32x32x2 MFMA); lane
// l = (p = l&31, h = l>>5) holds half of point p's features.  Activations never leave registers
// between layers (kpn_common.h).  Weights stream from the packed buffer (L2-resident, 0.4 MB).
#include "kpn_device.h"
#include "kpn_field_shared.h"

// ---------------------------------------------------------------------------------------------
// lean != 0 (the render passes): the colour of a masked point is never looked at — its density is exactly 0, so it
// contributes 0 * rgb to the composited ray — hence the projection loop stops at the first view that rejects the point
// and the masked result carries rgb = 0.  lean == 0 (kpn_query): the reference's full result, incl. the plain average of
// the sampled source colours.
// One workgroup handles ppt * 256 consecutive points (ppt <= KPN_MASK_PPT) and reserves its slice of the list with ONE
// atomic: 16.7 M points per pass would otherwise be 262 k atomics on the same address, which is what bounded the kernel.
// The list stays in ascending point order inside a workgroup.
#define KPN_MASK_PPT 8
__global__ __launch_bounds__(256) void k_mask_compact(kpn_scene_dev sc, kpn_points ps, int64_t N, int mode, int lean, int ppt,
                                                      const float* __restrict__ wscalars, float* __restrict__ out,
                                                      uint8_t* __restrict__ valid, int* __restrict__ list,
                                                      int* __restrict__ count) {
    __shared__ unsigned long long mask_s[KPN_MASK_PPT][4];
    __shared__ int off_s[KPN_MASK_PPT][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * 256 * ppt + threadIdx.x;
    for (int k = 0; k < ppt; ++k) {
        const int64_t n = n0 + (int64_t)k * 256;
        int is_valid = 0;
        if (n < N) {
            float P[3], D[3];
            kpn_get_point<true>(ps, n, P, D);
            int all_in = 1, all_fg = 1;
            float acc[3] = {0.f, 0.f, 0.f};
            const float pu = 1.0f / (float)sc.V;
            const size_t HW = (size_t)sc.H * sc.W;
            for (int v = 0; v < sc.V; ++v) {
                const float* tb = sc.table + (size_t)v * KPN_TBL_STRIDE;
                const kpn_proj q = kpn_project<true>(tb, P[0], P[1], P[2], sc);
                all_in &= q.in;
                if (lean && !all_in) break;
                const kpn_taps tp = kpn_make_taps<true>(q.xn, q.yn, sc.H, sc.W);
                const float4 s = kpn_tap4<true>(sc.rgbm + (size_t)v * HW * 4, 4, 0, tp);
                if (!sc.disable_fg_mask) all_fg &= (s.w > 0.1f);  // model.py:737-739
                if (lean && !all_fg) break;
                acc[0] = KADD(acc[0], KMUL(s.x, pu)); acc[1] = KADD(acc[1], KMUL(s.y, pu)); acc[2] = KADD(acc[2], KMUL(s.z, pu));
            }
            // a_v = in_v * all(fg) * all(in) * dropout_v (model.py:739,748); valid = sum_v a_v > 0 (utils.py:643-646)
            is_valid = all_in && all_fg && ((sc.keep & ((1u << sc.V) - 1u)) != 0u);
            if (valid) valid[n] = (uint8_t)is_valid;
            if (!is_valid && out) {
                // every view masked: pooled features are exactly 0 and the IBR softmax is uniform, so the
                // reference's result is a constant + the plain average of the sampled source colours
                float* o = out + n * 5;
                if (mode == 1) { o[0] = 0.0f; o[1] = 0.1f / sc.nml_scale; }  // eval_func, model.py:981-996
                else { o[0] = wscalars[1]; o[1] = wscalars[2]; }
                o[2] = lean ? 0.0f : acc[0]; o[3] = lean ? 0.0f : acc[1]; o[4] = lean ? 0.0f : acc[2];
            }
        }
        const unsigned long long m = __ballot(is_valid);
        if (lane == 0) mask_s[k][w] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int k = 0; k < ppt; ++k)
            for (int j = 0; j < 4; ++j) { off_s[k][j] = total; total += __popcll(mask_s[k][j]); }
        const int base = total ? atomicAdd(count, total) : 0;
        for (int k = 0; k < ppt; ++k)
            for (int j = 0; j < 4; ++j) off_s[k][j] += base;
    }
    __syncthreads();
    for (int k = 0; k < ppt; ++k) {
        const unsigned long long m = mask_s[k][w];
        if ((m >> lane) & 1ull) list[off_s[k][w] + __popcll(m & ((1ull << lane) - 1ull))] = (int)(n0 + (int64_t)k * 256);
    }
}

__global__ __launch_bounds__(256, 2) void k_geo_rows(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                                     const int* __restrict__ list, const int* __restrict__ count_ptr,
                                                     int* __restrict__ tickets, float* __restrict__ xscr, kpn_batch batch) {
    const int lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    if (!kpn_batch_gate(batch, sc, wp)) return;
    const int count = *count_ptr;
    const int ntiles = (count + KPN_TILE - 1) / KPN_TILE;
    int t0, t1;
    if (!kpn_batch_range(batch, ntiles, t0, t1)) return;
    const int nwork = (t1 - t0) * sc.V;
    // the four bias blocks (448 floats) sit in LDS for the lifetime of the persistent workgroup: a layer
    // starts with a ds_read instead of an exposed L2 round trip
    __shared__ __attribute__((aligned(16))) float bias_s[4][128];
    {
        const int segs[4] = {SEG_G1_0A, SEG_G1_1, SEG_G1_2, SEG_G1_3};
        for (int i = threadIdx.x; i < 4 * 128; i += blockDim.x) {
            const int sg = i >> 7, k = i & 127;
            bias_s[sg][k] = k < kpn_seg_bfloats(segs[sg]) ? wp[kpn_seg_boff(segs[sg]) + k] : 0.0f;
        }
    }
    __syncthreads();

    // work items are drawn from a device-wide ticket counter (zeroed by the launcher): waves on CUs that run
    // slower simply draw fewer tiles, which removes the tail of a static round-robin assignment
    (void)wave; (void)nwaves;
    for (;;) {
        int wi = 0;
        if (lane == 0) wi = atomicAdd(tickets + 0, 1);
        wi = __shfl(wi, 0);
        if (wi >= nwork) break;
        const int tr = wi / sc.V, v = wi - tr * sc.V;   // tile relative to the batch; the scratch slot is wi
        int ci = (t0 + tr) * KPN_TILE + p;
        if (ci >= count) ci = count - 1;  // pad lanes recompute the last point; their result is never read
        const int64_t n = list[ci];
        float P[3], D[3];
        kpn_get_point(ps, n, P, D);
        const float* tb = sc.table + (size_t)v * KPN_TBL_STRIDE;
        const kpn_proj q = kpn_project(tb, P[0], P[1], P[2], sc);
        if (!((sc.keep >> v) & 1u)) {
            // view switched off by the train-time dropout: its pooling / blend weights are 0, only the record
            // (the blend-weight minimum runs over ALL views, model.py:1288) is needed
            float4 rec0, rec1;
            kpn_row_record(sc, ps, n, tb, v, h, rec0, rec1);
            float4* dst = reinterpret_cast<float4*>(xscr) + ((size_t)wi * KPN_ROW_SLABS) * 64 + lane;
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[k * 64] = make_float4(0.f, 0.f, 0.f, 0.f);
            dst[8 * 64] = rec0;
            dst[9 * 64] = rec1;
            continue;
        }

        // Activations are applied lazily: layer L+1 takes softplus(acc_L[...]) group by group as its B
        // operands, so the transcendental work interleaves with the MFMAs instead of forming a
        // VALU-only phase between layers.
        // ---- layers1.0 : [168 keypoint encoding | 64 geometry channels] -> 128 ----
        kpn_f32x16 a0[4];
        {
            const float* E = tb + KPN_TBL_EXT;  // camera-space position, spatial.py:76
            const float cx = RADD(kpn_dot3(P[0], P[1], P[2], E[0], E[1], E[2]), E[3]);
            const float cy = RADD(kpn_dot3(P[0], P[1], P[2], E[4], E[5], E[6]), E[7]);
            const float cz = RADD(kpn_dot3(P[0], P[1], P[2], E[8], E[9], E[10]), E[11]);
            const float* kc = tb + KPN_TBL_KCAM + (12 * h) * 3;
            kpn_load_bias<4>(bias_s[0], h, a0);
            // group j = keypoint j + 12h: 7 encoding values (spatial.py:110-118), produced while the
            // previous keypoint's 28 MFMAs issue
            kpn_mfma_layer<84, 4, 7>(wp + kpn_seg_woff(SEG_G1_0A), lane, [&](auto gi, float (&x)[7]) {
                constexpr int j = decltype(gi)::value;
                const float dx = RSUB(cx, kc[j * 3 + 0]), dy = RSUB(cy, kc[j * 3 + 1]), dz = RSUB(cz, kc[j * 3 + 2]);
                const float d2 = RADD(RADD(RMUL(dx, dx), RMUL(dy, dy)), RMUL(dz, dz));
                const float w = kpn_fast_exp(-d2 / sc.two_sigma2);
                float s1, c1;
                kpn_sincos_pi(dz, s1, c1);
                // sin/cos(2y), sin/cos(4y): the reference's arguments are exactly 2y and 4y
                // (float32(2*pi) == 2*float32(pi)), so the double-angle identities apply to them
                const float s2 = 2.0f * s1 * c1, c2 = 1.0f - 2.0f * s1 * s1;
                const float s4 = 2.0f * s2 * c2, c4 = 1.0f - 2.0f * s2 * s2;
                x[0] = dz * w;
                x[1] = s1 * w; x[2] = c1 * w;
                x[3] = s2 * w; x[4] = c2 * w;
                x[5] = s4 * w; x[6] = c4 * w;
            }, a0);
            const kpn_taps tp = kpn_make_taps(q.xn, q.yn, sc.g0h, sc.g0w);  // model.py:763-765
            const float* g0 = sc.geo0 + (size_t)v * sc.g0h * sc.g0w * 64;
            kpn_mfma_layer<32, 4, 4>(wp + kpn_seg_woff(SEG_G1_0B), lane, [&](auto gi, float (&x)[4]) {
                constexpr int g = decltype(gi)::value;
                const float4 f = kpn_tap4(g0, 64, 32 * h + 4 * g, tp);
                x[0] = f.x; x[1] = f.y; x[2] = f.z; x[3] = f.w;
            }, a0);
        }
        // ---- layers1.1 : softplus(128) -> 128 ----
        kpn_f32x16 a1[4];
        kpn_load_bias<4>(bias_s[1], h, a1);
        kpn_mfma_layer<64, 4, 4>(wp + kpn_seg_woff(SEG_G1_1), lane, [&](auto gi, float (&x)[4]) {
            constexpr int g = decltype(gi)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = kpn_softplus100(a0[g / 4][(g % 4) * 4 + i]);
        }, a1);
        // ---- layers1.2 : [softplus(128) | 8 hd channels] -> 120 (skip connection, utils.py:706-713) ----
        kpn_f32x16 a2[4];
        {
            const kpn_taps tp = kpn_make_taps(q.xn, q.yn, sc.g1h, sc.g1w);
            const float4 f = kpn_tap4(sc.geo1 + (size_t)v * sc.g1h * sc.g1w * 8, 8, 4 * h, tp);
            kpn_load_bias<4>(bias_s[2], h, a2);
            kpn_mfma_layer<68, 4, 4>(wp + kpn_seg_woff(SEG_G1_2), lane, [&](auto gi, float (&x)[4]) {
                constexpr int g = decltype(gi)::value;
                if constexpr (g < 16) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) x[i] = kpn_softplus100(a1[g / 4][(g % 4) * 4 + i]);
                } else {
                    x[0] = f.x; x[1] = f.y; x[2] = f.z; x[3] = f.w;
                }
            }, a2);
        }
        // ---- layers1.3 : softplus(120) -> 64, linear.  The colour head's per-(point,view) gathers ride along
        //      (query_color, model.py:806-832): this kernel already has the projection and VALU/memory slack,
        //      k_fuse_color has neither.  They are produced inside this layer's operand callback so that their
        //      loads and arithmetic interleave with its MFMAs. ----
        {
            kpn_f32x16 acc[2];
            float4 rec0 = make_float4(0.f, 0.f, 0.f, 0.f), rec1 = rec0;
            kpn_load_bias<2>(bias_s[3], h, acc);
            kpn_mfma_layer<64, 2, 4>(wp + kpn_seg_woff(SEG_G1_3), lane, [&](auto gi, float (&x)[4]) {
                constexpr int g = decltype(gi)::value;
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = kpn_softplus100(a2[g / 4][(g % 4) * 4 + i]);
                if constexpr (g == 2) kpn_row_record(sc, ps, n, tb, v, h, rec0, rec1);
            }, acc);
            float4* dst = reinterpret_cast<float4*>(xscr) + ((size_t)wi * KPN_ROW_SLABS) * 64 + lane;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
                    dst[(b * 4 + qd) * 64] =
                        make_float4(acc[b][4 * qd + 0], acc[b][4 * qd + 1], acc[b][4 * qd + 2], acc[b][4 * qd + 3]);
            dst[8 * 64] = rec0;
            dst[9 * 64] = rec1;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Per-view inputs of the IBR head for this lane's point (query_color, model.py:806-832), in two steps:
// the gather (projection, bilinear taps, ray-direction difference: VALU + memory only) and the
// ray_encoder MLP (MFMA).  With V <= 3 the gathers of all views are issued up front.
struct kpn_view_gather {
    float pw;       // un-normalised boundary-smooth pooling weight (model.py:752-758)
    float rgb[3];   // sampled source colour
    float fadd[7];  // this lane's rgb/tex entries of x': regs 12..15 of block 0, regs 0..2 of block 1
    float rd[4];    // ray_diff = [direction(3), dot]
};
struct kpn_ibr_view {
    float xb0[16];  // x' rows rowmap(r,h): x' = [lat24 | rgb3 | tex8] + ray_encoder(ray_diff)   (block 0)
    float xb1[3];   // x' rows 32..34 (h == 0 lanes only)
};

// the gather record of (tile t, view v) written by k_geo_rows: own half + partner half (lane ^ 32)
__device__ __forceinline__ void kpn_gather_view(const float4* __restrict__ rec, int lane, int h, kpn_view_gather& o) {
    const float4 own0 = rec[lane], own1 = rec[64 + lane], oth0 = rec[lane ^ 32], oth1 = rec[64 + (lane ^ 32)];
    const float4 a0 = h ? oth0 : own0, a1 = h ? oth1 : own1;  // [r,g,b,pw] , [ray_diff(3), dot]
    const float4 t0 = h ? own0 : oth0, t1 = h ? own1 : oth1;  // texture channels 0..3, 4..7
    o.pw = a0.w;
    o.rgb[0] = a0.x; o.rgb[1] = a0.y; o.rgb[2] = a0.z;
    // x' order is [lat24 | rgb3 | tex8]: rows 24..27 live in regs 12..15 of the h=0 lanes, rows 28..31 in
    // regs 12..15 of h=1, rows 32..34 in block 1 regs 0..2 of h=0
    o.fadd[0] = h ? t0.y : a0.x; o.fadd[1] = h ? t0.z : a0.y; o.fadd[2] = h ? t0.w : a0.z; o.fadd[3] = h ? t1.x : t0.x;
    o.fadd[4] = h ? 0.0f : t1.y; o.fadd[5] = h ? 0.0f : t1.z; o.fadd[6] = h ? 0.0f : t1.w;
    o.rd[0] = a1.x; o.rd[1] = a1.y; o.rd[2] = a1.z; o.rd[3] = a1.w;
}

// The per-point kernel exists with two weight formats (template parameter F16 of its body):
//   false: fp32 streams on v_mfma_f32_32x32x2_f32 (k_fuse_color; the region [kpn_k2_base(), + kpn_k2_floats()) of the packed buffer)
//   true : two fp16 pieces per value, three products on v_mfma_f32_32x32x16_f16 (k_fuse_color_h; the kpn_cseg_* region) — the
//          k_geo_rows_f2 arithmetic: a quarter of the matrix time for the same fp32-class results
// kpn_fuse_w<F16> maps a segment / scalar / row vector to its offset in the packed buffer (the LDS pointer is biased by -base).
template <bool F16> struct kpn_fuse_w;
template <> struct kpn_fuse_w<false> {
    static constexpr int base = kpn_k2_base(), floats = kpn_k2_floats();
    static constexpr int woff(int seg) { return kpn_seg_woff(seg); }
    static constexpr int boff(int seg) { return kpn_seg_boff(seg); }
    static constexpr int scalars() { return kpn_scalar_off(); }
    static constexpr int row(int r) { return kpn_row_off(r); }
};
template <> struct kpn_fuse_w<true> {
    static constexpr int base = kpn_k2h_base(), floats = kpn_k2h_floats();
    static constexpr int woff(int seg) { return kpn_cseg_woff(seg); }
    static constexpr int boff(int seg) { return kpn_cseg_boff(seg); }
    static constexpr int scalars() { return kpn_k2h_tail_off(); }
    static constexpr int row(int r) { return kpn_k2h_tail_off() + (kpn_row_off(r) - kpn_scalar_off()); }
};
// one Linear layer of segment SEG from the LDS copy, either format; in_fn as kpn_mfma_layer with G = 4
template <bool F16, int SEG, int KS, int NOB, class InFn>
__device__ __forceinline__ void kpn_fuse_layer(const float* __restrict__ wl, int lane, InFn&& in_fn, kpn_f32x16 (&acc)[NOB]) {
    static_assert(KS == kpn_seg_shapes[SEG].ks && NOB == kpn_seg_shapes[SEG].nob, "segment shape");
    if constexpr (F16) kpn_hlayer<KS, NOB, SEG>(wl + kpn_fuse_w<true>::woff(SEG), lane, in_fn, acc);
    else kpn_mfma_layer<KS, NOB, 4, 1>(wl + kpn_fuse_w<false>::woff(SEG), lane, in_fn, acc);
}
template <bool F16, int SEG, int KS, int NOB, int NSRC>
__device__ __forceinline__ void kpn_fuse_layer_regs(const float* __restrict__ wl, int lane, const float (&src)[NSRC], kpn_f32x16 (&acc)[NOB]) {
    static_assert(KS == kpn_seg_shapes[SEG].ks && NOB == kpn_seg_shapes[SEG].nob, "segment shape");
    if constexpr (F16) kpn_hlayer_regs<KS, NOB, SEG>(wl + kpn_fuse_w<true>::woff(SEG), lane, src, acc);
    else kpn_mfma_layer_regs<KS, NOB, 4, 1>(wl + kpn_fuse_w<false>::woff(SEG), lane, src, acc);
}

// ray_encoder: Linear(4,16) ELU Linear(16,35) ELU (model.py:1246,1279), then x' = rgb_feat' + dir' (:1281-1284)
template <bool F16 = false>
__device__ __forceinline__ void kpn_encode_view(const float* __restrict__ wl, int lane, int h, const kpn_view_gather& g,
                                                const float (&lat0)[16], kpn_ibr_view& o) {
    using W = kpn_fuse_w<F16>;
    const float in4[4] = {h ? g.rd[1] : g.rd[0], h ? g.rd[3] : g.rd[2], 0.0f, 0.0f};
    kpn_f32x16 a1[1];
    kpn_load_bias<1>(wl + W::boff(SEG_RE_0), h, a1);
    kpn_fuse_layer_regs<F16, SEG_RE_0, 4, 1>(wl, lane, in4, a1);
    float in8[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) in8[r] = kpn_elu(a1[0][r]);
    kpn_f32x16 a2[2];
    kpn_load_bias<2>(wl + W::boff(SEG_RE_1), h, a2);
    kpn_fuse_layer_regs<F16, SEG_RE_1, 8, 2>(wl, lane, in8, a2);
#pragma unroll
    for (int r = 0; r < 16; ++r) o.xb0[r] = kpn_elu(a2[0][r]) + lat0[r];
    o.xb0[12] += g.fadd[0]; o.xb0[13] += g.fadd[1]; o.xb0[14] += g.fadd[2]; o.xb0[15] += g.fadd[3];
    o.xb1[0] = kpn_elu(a2[1][0]) + g.fadd[4];
    o.xb1[1] = kpn_elu(a2[1][1]) + g.fadd[5];
    o.xb1[2] = kpn_elu(a2[1][2]) + g.fadd[6];
}

// View pooling of the 64-vector (PoolModule / pool_ops, utils.py:612-647, 731-748): weighted mean and variance over the source
// views.  The un-normalised boundary-smooth weights (model.py:752-759; mask == 1 in every view for listed points) come with the
// gather records k_geo_rows wrote; dropped views weigh 0.  pooled[16b + r] = mean of feature 32b + rowmap(r,h),
// pooled[32 + 16b + r] = its variance (a lane's half of the point); PWSUM = the weights' sum (before the + 1e-6 of the normalisation).
// ONE pass over the rows (they are read from the row scratch: the reference's two passes cost a second fetch of every row —
// measured 0.4 ms per 512x512 frame), shifted by the first kept view's row x0 (d = x - x0):
//     s1 = sum pw d,  s2 = sum pw d^2,  mean = x0 sum pw + s1,  var = sum pw (x - mean)^2 = s2 - 2 m s1 + m^2 sum pw,  m = mean - x0.
// The shift keeps the subtraction benign: where the views agree (var << mean^2) d is small and so are all three terms; the result
// differs from the two-pass form by rounding only (200-scene sweep against the oracle: 14 rays above 1e-4 instead of 13, all
// ill-conditioned by the oracle's own probe).  A macro, not a function: through a helper function the forward kernel measured
// 1.7 ms per frame slower (register allocation), and the three users (k_fuse_color*, the split colour path, the backward kernels)
// must stay bit-identical.
#define KPN_POOL_VIEWS(ROWS, V_, KEEP, LANE, P_, PWSUM, POOLED)                                                              \
    do {                                                                                                                      \
        PWSUM = 0.0f;                                                                                                         \
        for (int v_ = 0; v_ < (V_); ++v_)                                                                                     \
            if (((KEEP) >> v_) & 1u) PWSUM = RADD(PWSUM, (ROWS)[((size_t)v_ * KPN_ROW_SLABS + 8) * 64 + (P_)].w);             \
        _Pragma("unroll") for (int i_ = 0; i_ < 64; ++i_) POOLED[i_] = 0.0f;                                                  \
        int v0_ = 0;                                                                                                          \
        while (v0_ < (V_) && !(((KEEP) >> v0_) & 1u)) ++v0_;                                                                  \
        if (v0_ < (V_)) {                                                                                                     \
            float x0_[32];                                                                                                    \
            const float4* src0_ = (ROWS) + ((size_t)v0_ * KPN_ROW_SLABS) * 64;                                                \
            float spw_ = src0_[8 * 64 + (P_)].w / RADD(PWSUM, 1e-6f);                                                         \
            _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                                                                \
                const float4 x_ = src0_[q_ * 64 + (LANE)];                                                                    \
                x0_[4 * q_ + 0] = x_.x; x0_[4 * q_ + 1] = x_.y; x0_[4 * q_ + 2] = x_.z; x0_[4 * q_ + 3] = x_.w;               \
            }                                                                                                                 \
            for (int v_ = v0_ + 1; v_ < (V_); ++v_) {                                                                         \
                if (!(((KEEP) >> v_) & 1u)) continue;                                                                         \
                const float4* src_ = (ROWS) + ((size_t)v_ * KPN_ROW_SLABS) * 64;                                              \
                const float pw_ = src_[8 * 64 + (P_)].w / RADD(PWSUM, 1e-6f);                                                 \
                spw_ = RADD(spw_, pw_);                                                                                       \
                _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                                                            \
                    const float4 x_ = src_[q_ * 64 + (LANE)];                                                                 \
                    const float xe_[4] = {x_.x, x_.y, x_.z, x_.w};                                                            \
                    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                                        \
                        const int i_ = 4 * q_ + e_;                                                                           \
                        const float d_ = RSUB(xe_[e_], x0_[i_]);                                                              \
                        POOLED[i_] = fmaf(pw_, d_, POOLED[i_]);                                                               \
                        POOLED[32 + i_] = fmaf(RMUL(pw_, d_), d_, POOLED[32 + i_]);                                           \
                    }                                                                                                         \
                }                                                                                                             \
            }                                                                                                                 \
            _Pragma("unroll") for (int i_ = 0; i_ < 32; ++i_) {                                                               \
                const float s1_ = POOLED[i_], s2_ = POOLED[32 + i_];                                                          \
                const float mean_ = fmaf(x0_[i_], spw_, s1_), m_ = RSUB(mean_, x0_[i_]);                                      \
                POOLED[i_] = mean_;                                                                                           \
                POOLED[32 + i_] = fmaxf(fmaf(m_, fmaf(m_, spw_, RMUL(-2.0f, s1_)), s2_), 0.0f);                               \
            }                                                                                                                 \
        }                                                                                                                     \
    } while (0)
__device__ __forceinline__ float kpn_pool_views(const float4* __restrict__ rows, int V, uint32_t keep, int lane, int p,
                                                float (&pooled)[64]) {
    float pwsum;
    KPN_POOL_VIEWS(rows, V, keep, lane, p, pwsum, pooled);
    return pwsum;
}

// Copy of the k_fuse_color region of the packed weights (segments SEG_G2_0.., scalars, row vectors) into LDS.  The W part
// of a segment whose lanes take more than one float4 per group (NOB = 2: G * NOB / 4 = 2) is re-laid from the global
// [group][lane][q] to [group][q][lane], the conflict-free order for ds_read_b128 (kpn_load_group<NQ, 1>); everything
// else (NQ = 1 streams, biases, scalars, row vectors) is copied as it is.
// [base, base + nfloats) of the packed buffer (whole segments seg0 .. seg1-1, plus what follows the last one)
__device__ __forceinline__ void kpn_stage_lds_range(const float* __restrict__ wp, float* __restrict__ wlds, int base, int nfloats,
                                                    int seg0, int seg1) {
    const float4* src = reinterpret_cast<const float4*>(wp + base);
    float4* dst = reinterpret_cast<float4*>(wlds);
    for (int i = threadIdx.x; i < nfloats / 4; i += blockDim.x) {
        int j = i;
#pragma unroll
        for (int seg = SEG_G2_0; seg < SEG_COUNT; ++seg) {
            if (seg < seg0 || seg >= seg1) continue;
            const int nq = kpn_seg_shapes[seg].g * kpn_seg_shapes[seg].nob / 4;
            if (nq == 1) continue;
            const int w0 = (kpn_seg_woff(seg) - base) / 4, w1 = w0 + kpn_seg_wfloats(seg) / 4;
            if (i >= w0 && i < w1) {
                const int r = i - w0, g = r / (64 * nq), e = r - g * 64 * nq;   // e = lane * nq + q
                j = w0 + g * 64 * nq + (e % nq) * 64 + e / nq;
            }
        }
        dst[j] = src[i];
    }
}
__device__ __forceinline__ void kpn_stage_lds_streams(const float* __restrict__ wp, float* __restrict__ wlds) {
    kpn_stage_lds_range(wp, wlds, kpn_k2_base(), kpn_k2_floats(), SEG_G2_0, SEG_COUNT);
}

// Any V <= KPN_MAXV: the per-view x' vectors are recomputed in each of the three passes over the views (two for the
// weighted mean / variance, one for the head).  A V <= 3 variant that kept them in registers was measured slower: this
// kernel is register-bound, and every spilled VGPR costs more than re-running the 624-MAC ray encoder.
#ifdef KPN_FUSE_TIMING   // debug builds: cycles (s_memtime) per phase of one wave's tiles, summed (scripts/fuse_timing.py)
__device__ unsigned long long kpn_fuse_cycles[8];
#define KPN_FUSE_STAMP(i) do { const unsigned long long now_ = clock64(); if (blockIdx.x == 3 && threadIdx.x == 64) atomicAdd(&kpn_fuse_cycles[i], now_ - fstamp_); fstamp_ = now_; } while (0)
#else
#define KPN_FUSE_STAMP(i) ((void)0)
#endif
// VFIX > 0: the number of source views is known at compile time and none of them is dropped (eval passes with the shipped V = 3,
// src/zju_dataset.py:45): the STATISTICS loops over the views (blend weights, gather -> ray_encoder -> x' -> Welford update) are
// unrolled, so that the scheduler sees the views' independent chains in one block and starts a view's gather loads and matrix
// chains under another view's arithmetic.  Measured on the MI355X (same box, profiles/r05_d_per_point_kernel_variants.txt): frame
// 21.25 -> 21.04 ms with 18 VGPRs spilled to scratch; the per-view HEADS unrolled as well spill 84 VGPRs at two waves per SIMD, and
// as ONE wave per SIMD with 512 registers (no spills) the frame is 21.8-22.0 ms: hipcc keeps every head's MFMA chain contiguous
// (12 dependent MFMAs in a row), it does not interleave the three heads, and the accumulators move to AGPRs (518 v_accvgpr_read).
// VFIX = 0: any V <= KPN_MAXV, any keep mask.
//
// kpn_density_counts (kpn_density_stats): [0] points a render pass's per-point kernels took density decisions for, [1] those found
// live (!(rad <= 0)) — one addition per wave at its exit.  A device global of the module, like the range guard's counter.
#ifndef KPN_SIMT_EMU
__device__ unsigned long long kpn_density_counts[2];
#else
static unsigned long long kpn_density_counts[2];
#endif
// PHASE (round 6) — density first, colour for the LIVE points only (reference src/model.py:981-996 eval_func, :1150-1176 rgba2out:
// a sample with relu(rad) == 0 composites with weight 1 - exp(-0 * delta) = 0 EXACTLY, so its colour never reaches the image):
//   0: the fused kernel — density and colour of every listed point, tile by tile (kpn_query, the train branch, the backward's
//      forward, the fp32-range kernels behind the range guard, and the render passes of a density that is live nearly everywhere);
//      its exact short path works per 32-point TILE;
//   1: pass A of a render pass (k_density_h): pooled vector -> layers2 -> [sdf, rad] -> the point's density record, + the
//      compress layer's 24 latent values written IN PLACE over the first four pooled slabs of the tile (this wave has just consumed
//      all sixteen; nobody else reads them).  A point with !(rad <= 0) (a NaN counts as live: it must reach the range guard) is
//      pushed on the wave's queue of live scratch slots (tile * 32 + point) in LDS, which is appended to the batch's LIVE list 256
//      slots at a time (one reservation); a dead point gets rgb = 0 and is done;
//   2: pass B (k_colour_h / k_colour_h3): tiles of 32 LIVE points, each lane addressing the scratch by the slot its list entry names
//      — the latent values, the gather records (k_row_records_live forms them for the listed points only) — V heads, blend, rgb.
// Per point the arithmetic is the fused kernel's instruction for instruction (a point's column of an MFMA depends on no other
// column), so the frame is bit-identical with density first on or off, whichever points end up sharing a tile.
// Measured on the MI355X (profiles/r06_a_density_first.txt): pass A is HBM-bound (it reads the 512-B pooled vector of every point:
// 0.53 ms per launch), pass B costs what the fused kernel's colour part costs (1.13 ms with every point live), so on a density that
// is live everywhere the pair is 0.16 ms per launch SLOWER than the fused kernel (1.50 ms), at 25 % dead points it is 1 % of a frame ahead, at 82 %
// it wins 3 % of the frame on top of the fused kernel's own short path.  Which of the two runs is therefore decided per render call
// from the dead fraction the previous calls measured (kpn_api.hip density_first_now).  A third form — one kernel, the colour part
// run on a per-wave LDS queue of live slots as soon as it holds 32 — was built and measured as well: 1.59 ms all-live (30 spilled
// VGPRs) and no better than the fused kernel at 82 % dead; not kept.
template <bool F16, int VFIX = 0, int PHASE = 0>
__device__ __forceinline__ void kpn_fuse_color_body(const kpn_scene_dev& sc, const kpn_points& ps, const float* __restrict__ wp,
                                                    const int* __restrict__ list, const int* __restrict__ count_ptr,
                                                    int* __restrict__ tickets, const float* __restrict__ xscr, int mode,
                                                    int* __restrict__ live, float* __restrict__ out, const kpn_batch& batch, int zero_skip) {
    static_assert(PHASE == 0 || F16, "density first exists for the two-fp16-piece kernels only");
    static_assert(PHASE >= 0 && PHASE <= 2, "PHASE");
    using W = kpn_fuse_w<F16>;
    const int lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    if (!kpn_batch_gate(batch, sc, wp)) return;
    const int count = *count_ptr;
    int t0, t1;
    if (!kpn_batch_range(batch, (count + KPN_TILE - 1) / KPN_TILE, t0, t1)) return;   // before the LDS staging
    if (batch.cond == KPN_RUN_IF_UNSAFE && batch.redone != nullptr && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(batch.redone, 1);
    // tickets: [1] the tile ticket of the fused kernel / pass A, [2] the number of slots on the batch's live list, [3] pass B's ticket
    const int nlive = PHASE == 2 ? tickets[2] : 0;
    if (PHASE == 2 && nlive == 0) return;   // before the LDS staging
    const int ntiles = PHASE == 2 ? (nlive + KPN_TILE - 1) / KPN_TILE : t1 - t0;
    int* const ticket = tickets + (PHASE == 2 ? 3 : 1);
    const int V = VFIX > 0 ? VFIX : sc.V;
    const kpn_tile_layout lay(batch.pool, V);   // ROWS or POOL layout of the scratch (kpn_field_shared.h)
    // all weights of this kernel live in LDS for the lifetime of the (persistent) workgroup;
    // wl is biased so that the packed-buffer offsets (kpn_seg_woff etc.) index it directly
    __shared__ __attribute__((aligned(16))) float wlds[W::floats];
    if constexpr (F16) {   // the fp16 region needs no re-layout: its streams are packed lane-contiguous
        const float4* src = reinterpret_cast<const float4*>(wp + W::base);
        float4* dst = reinterpret_cast<float4*>(wlds);
        for (int i = threadIdx.x; i < W::floats / 4; i += blockDim.x) dst[i] = src[i];
    } else {
        kpn_stage_lds_streams(wp, wlds);
    }
    __syncthreads();
    const float* wl = wlds - W::base;
    const float ani = wl[W::scalars() + 0];  // |ani_al|

    (void)wave; (void)nwaves;
#ifdef KPN_FUSE_TIMING
    unsigned long long fstamp_ = clock64();
#endif
    unsigned stat_listed = 0, stat_live = 0;
    const float4* const scr_base = reinterpret_cast<const float4*>(xscr);
    // pass A: the wave's queue of live scratch slots in LDS, appended to the batch's live list KPN_LIVEQ slots (or what is left) at a
    // time: one reservation per 256 slots instead of one per tile
    constexpr int KPN_LIVEQ = 256;
    __shared__ int liveq[PHASE == 1 ? 8 : 1][PHASE == 1 ? KPN_LIVEQ : 1];
    int* const myq = liveq[PHASE == 1 ? (threadIdx.x >> 6) & 7 : 0];
    int qn = 0;
    // One tile.  PH = 0: the fused tile (PHASE 0); 1: the density part of valid-list tile t; 2: the colour part of the 32 live slots
    // live_src[t * 32 ...] (nlive_src of them in all).
    auto tile = [&](auto phase_c, const int t, const int* __restrict__ live_src, const int nlive_src) __attribute__((always_inline)) {
        constexpr int PH = decltype(phase_c)::value;
        // this lane's point: entry ci_raw of the valid list (pass B: of the batch's live list, whose entry names the scratch slot
        // (tile t_scr, point p_scr) pass A found the point in); l_scr = the lane's own position in that tile's slabs
        const int ci_raw = (PH == 2 ? 0 : t0 * KPN_TILE) + t * KPN_TILE + p;
        const int ci_end = PH == 2 ? nlive_src : count;
        const int ci = ci_raw < ci_end ? ci_raw : ci_end - 1;
        int t_scr = t, l_scr = lane;
        int64_t n;
        if constexpr (PH == 2) {
            const int slot = live_src[ci];
            t_scr = slot >> 5;
            l_scr = (h << 5) | (slot & 31);
            n = list[(t0 + t_scr) * KPN_TILE + (slot & 31)];
        } else {
            n = list[ci];
        }
        const int p_scr = l_scr & 31;

        KPN_FUSE_STAMP(0);
        const float4* const scr = scr_base;
        const float4* rows = scr + lay.tile(t_scr) * 64;
        const uint32_t keep = VFIX > 0 ? 0xFFFFFFFFu : sc.keep;  // train-time view dropout (all ones in eval): weights of dropped views are 0
        float sdf_raw = 0.0f, rad = 0.0f;
        float lat0[16];
        kpn_u32x4 pooled_h[F16 ? 8 : 1], pooled_l[F16 ? 8 : 1];
        float pooled[PH == 2 ? 1 : 64];  // K-steps 0..31 = mean (block b, reg r), 32..63 = var
        if constexpr (PH == 2) {   // the compress layer's output of this point, left by pass A in its tile's first four slabs
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_) {
                const float4 x_ = rows[q_ * 64 + l_scr];
                lat0[4 * q_ + 0] = x_.x; lat0[4 * q_ + 1] = x_.y; lat0[4 * q_ + 2] = x_.z; lat0[4 * q_ + 3] = x_.w;
            }
        } else {
        // ---- pooled mean / var over views of the 64-vector ----
        float pwsum;
        if (lay.pool) {    // POOL layout: the rows kernel has pooled already (slabs 0..7 mean, 8..15 variance of this tile)
#pragma unroll
            for (int q_ = 0; q_ < 16; ++q_) {
                const float4 x_ = rows[q_ * 64 + lane];
                pooled[4 * q_ + 0] = x_.x; pooled[4 * q_ + 1] = x_.y; pooled[4 * q_ + 2] = x_.z; pooled[4 * q_ + 3] = x_.w;
            }
            pwsum = 0.0f;
        } else {
            KPN_POOL_VIEWS(rows, V, keep, lane, p, pwsum, pooled);
        }
        (void)pwsum;
        KPN_FUSE_STAMP(1);
        // ---- layers2: 128 -> 64 -> 64 -> 2 (utils.py:577-587), activations applied lazily ----
        {
            kpn_f32x16 h0[2], h1[2], o2[1];
            kpn_load_bias<2>(wl + W::boff(SEG_G2_0), h, h0);
            if constexpr (F16) {   // the pooled vector is split once for layers2.0 and the compress layer below
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float x8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) x8[i] = pooled[8 * c + i];
                    kpn_split_f16x8(x8, pooled_h[c], pooled_l[c]);
                }
                kpn_hlayer_presplit<8, 2, SEG_G2_0>(wl + W::woff(SEG_G2_0), lane, pooled_h, pooled_l, h0);
            } else {
                kpn_fuse_layer_regs<F16, SEG_G2_0, 64, 2>(wl, lane, pooled, h0);
            }
            kpn_load_bias<2>(wl + W::boff(SEG_G2_1), h, h1);
            kpn_fuse_layer<F16, SEG_G2_1, 32, 2>(wl, lane, [&](auto gi, float (&x)[4]) {
                constexpr int g = decltype(gi)::value;
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = F16 ? kpn_softplus_log2(h0[g / 4][(g % 4) * 4 + i]) : kpn_softplus100(h0[g / 4][(g % 4) * 4 + i]);
            }, h1);
            kpn_load_bias<1>(wl + W::boff(SEG_G2_2), h, o2);
            kpn_fuse_layer<F16, SEG_G2_2, 32, 1>(wl, lane, [&](auto gi, float (&x)[4]) {
                constexpr int g = decltype(gi)::value;
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = F16 ? kpn_softplus_log2(h1[g / 4][(g % 4) * 4 + i]) : kpn_softplus100(h1[g / 4][(g % 4) * 4 + i]);
            }, o2);
            // rows 0,1 live in regs 0,1 of the h=0 lanes (the fp16 stream of layers2.2 is packed times 2^10: exact unscale)
            sdf_raw = F16 ? o2[0][0] * (1.0f / KPN_F16_G22_SCALE) : o2[0][0];
            rad = F16 ? o2[0][1] * (1.0f / KPN_F16_G22_SCALE) : o2[0][1];
        }
        // Exact zero-density short path (render passes only).
        // A point with relu(rad) == 0 composites with weight 1 - exp(-0 * delta) = 0 exactly, like a masked point, so its
        // colour never reaches the image (0 * rgb, rgb finite): when EVERY point of the tile is such a point — free space
        // inside the visual hull of a trained density comes in runs along the rays, i.e. in whole tiles of the ray-ordered
        // valid list — compress and the colour head (77 % of this kernel) are skipped.
        unsigned long long live_m = 0ull;   // pass A: the tile's live points (h = 0 lanes)
        int live_base = 0, flush_n = 0;
        if constexpr (PH == 1) {
            // (a NaN counts as live: it must reach the range guard, not be dropped as "density 0")
            live_m = __ballot(h == 0 && ci_raw < count && !(rad <= 0.0f));
            stat_listed += (unsigned)(count - (t0 + t) * KPN_TILE < KPN_TILE ? count - (t0 + t) * KPN_TILE : KPN_TILE);
            stat_live += (unsigned)__popcll(live_m);
            // the queue cannot take this tile's live points: reserve list space for what it holds — the atomic is issued here so
            // that its round trip passes under the compress layer; the copy follows it
            if (qn + __popcll(live_m) > KPN_LIVEQ) {
                flush_n = qn;
                if (lane == 0) live_base = atomicAdd(tickets + 2, qn);
            }
            if (live_m == 0ull) {   // nobody needs this tile's latent values
                if (h == 0 && ci_raw < count) {
                    float* o = out + n * 5;
                    o[0] = 0.0f; o[1] = sdf_raw; o[2] = 0.0f; o[3] = 0.0f; o[4] = 0.0f;
                }
                return;
            }
        } else if (zero_skip && mode == 1 && ps.noise == nullptr) {
            // (a NaN counts as live: it must reach the range guard at the end of the tile, not be skipped as "density 0")
            const unsigned long long live = __ballot(h == 0 && ci_raw < count && !(rad <= 0.0f));
            stat_listed += (unsigned)(count - (t0 + t) * KPN_TILE < KPN_TILE ? count - (t0 + t) * KPN_TILE : KPN_TILE);
            stat_live += (unsigned)__popcll(live);
            if (live == 0ull) {
                if (h == 0 && ci_raw < count) {
                    float* o = out + n * 5;
                    o[0] = 0.0f; o[1] = sdf_raw; o[2] = 0.0f; o[3] = 0.0f; o[4] = 0.0f;
                }
                return;
            }
        }
        KPN_FUSE_STAMP(2);
        // ---- ibr_compress_gfeat 128 -> 24 (model.py:819), rows already in x' order ----
        {
            kpn_f32x16 acc[1];
            kpn_load_bias<1>(wl + W::boff(SEG_CMP), h, acc);
            if constexpr (F16) kpn_hlayer_presplit<8, 1, SEG_CMP>(wl + W::woff(SEG_CMP), lane, pooled_h, pooled_l, acc);
            else kpn_fuse_layer_regs<F16, SEG_CMP, 64, 1>(wl, lane, pooled, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) lat0[r] = acc[0][r];
        }
        if constexpr (PH == 1) {
            // the latent values over the tile's first four pooled slabs (this wave has consumed all sixteen; no other wave reads them)
            float4* dstl = const_cast<float4*>(rows) + lane;
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_) dstl[q_ * 64] = make_float4(lat0[4 * q_ + 0], lat0[4 * q_ + 1], lat0[4 * q_ + 2], lat0[4 * q_ + 3]);
            if (h == 0 && ci_raw < count) {   // eval_func with mask = 1 (model.py:981-996); a live point's rgb is pass B's
                float* o = out + n * 5;
                o[0] = fmaxf(rad, 0.0f); o[1] = sdf_raw;
                if (!((live_m >> lane) & 1ull)) { o[2] = 0.0f; o[3] = 0.0f; o[4] = 0.0f; }
            }
            if (flush_n > 0) {
                KPN_WAVE_SYNC();
                live_base = __shfl(live_base, 0);
                for (int i = lane; i < flush_n; i += 64) live[live_base + i] = myq[i];
                KPN_WAVE_SYNC();
                qn = 0;
            }
            if ((live_m >> lane) & 1ull) myq[qn + __popcll(live_m & ((1ull << lane) - 1ull))] = t * KPN_TILE + p;
            qn += __popcll(live_m);
            if (batch.bad != nullptr && batch.cond != KPN_RUN_IF_UNSAFE) {   // the range guard, as at the end of the fused kernel
                const float chk = fabsf(sdf_raw) + fabsf(rad);
                if (__ballot(h == 0 && ci_raw < count && !(chk < 3.0e38f)) != 0ull && lane == 0) atomicOr(batch.bad, 1);
            }
            return;
        }
        }   // PH != 2
        KPN_FUSE_STAMP(3);
        // ---- IBR head (model.py:1267-1302) ----
        // blend weights (model.py:1287-1289): w_v = (e_v - min_v e) / (sum + 1e-8), e_v = exp(|a|(dot_v - 1))
        kpn_view_gather gv;
        kpn_ibr_view iv;
        float emin = 3.0e38f, esum = 0.0f;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass)
#pragma unroll(VFIX > 0 ? VFIX : 1)
            for (int v = 0; v < V; ++v) {
                const float dot = scr[(lay.rec(t_scr, v) + 1) * 64 + p_scr].w;
                const float e = kpn_fast_exp(RMUL(ani, RSUB(dot, 1.0f)));
                if (pass == 0) emin = fminf(emin, e);  // min over ALL views (:1288)
                else if ((keep >> v) & 1u) esum = RADD(esum, RSUB(e, emin));
            }
        // fused mean/var over views of x' (utils.py:91-95): K-steps mean' (16 + 3 + pad), var' (16 + 3 + pad).
        // ONE pass over the views: a weighted Welford update per view (W += w, d = x - mu, mu += (w / W) d, M2 += w (1 - w / W) d^2)
        // with the normalised blend weights w_v = (e_v - min e) / (sum + 1e-8) (model.py:1287-1289), finished as
        //     mean' = sum w x = W mu,   var' = sum w (x - mean')^2 = M2 + W (mu (1 - W))^2      (W = sum w is 1 only up to the 1e-8).
        // Round 3 ran two passes and read every parked x' back for the second (an L2 round trip per view in a latency-bound phase).
        float mv[40];
#pragma unroll
        for (int i = 0; i < 40; ++i) mv[i] = 0.0f;
        float wtot = 0.0f;
        auto stats = [&](float dot, const kpn_ibr_view& iv) {
            const float wv = RSUB(kpn_fast_exp(RMUL(ani, RSUB(dot, 1.0f))), emin) / RADD(esum, 1e-8f);
            wtot = RADD(wtot, wv);
            const float r = wtot > 0.0f ? wv / wtot : 0.0f;
            const float c = wv * (1.0f - r);
#pragma unroll
            for (int i = 0; i < 19; ++i) {
                const float x = i < 16 ? iv.xb0[i] : iv.xb1[i - 16];
                const float d = RSUB(x, mv[i]);
                mv[i] = fmaf(r, d, mv[i]);
                mv[20 + i] = fmaf(RMUL(c, d), d, mv[20 + i]);
            }
        };
        // x' of a view is needed twice (the statistics here, the head below): the head re-runs the gather and the 624-MAC ray
        // encoder.  Rounds 2-3 parked x' in the row scratch between the passes (three passes then); with ONE statistics pass the
        // store + read-back (5.5 GB written and as much read back per 512 x 512 frame, most of it missing L2) measured 1.2 % of the
        // frame SLOWER than recomputing.
#pragma unroll(VFIX > 0 ? VFIX : 1)
        for (int v = 0; v < V; ++v) {
            if (!((keep >> v) & 1u)) continue;
            kpn_gather_view(scr + lay.rec(t_scr, v) * 64, l_scr, h, gv);
            kpn_encode_view<F16>(wl, lane, h, gv, lat0, iv);
            stats(gv.rd[3], iv);
        }
        {
            const float om = 1.0f - wtot;
#pragma unroll
            for (int i = 0; i < 19; ++i) {
                const float dm = RMUL(mv[i], om);
                mv[20 + i] = fmaf(RMUL(wtot, dm), dm, mv[20 + i]);
                mv[i] = RMUL(mv[i], wtot);
            }
        }
        KPN_FUSE_STAMP(4);
        // view-invariant part of base_layer.0: W[:, mean|var] * [mean, var] + b
        kpn_f32x16 base[2];
        kpn_load_bias<2>(wl + W::boff(SEG_BL_0A), h, base);
        kpn_fuse_layer_regs<F16, SEG_BL_0A, 40, 2>(wl, lane, mv, base);

        // per view: rest of the head; online softmax over views of the colour logits, blending the
        // SOURCE colours (model.py:1300-1301)
        float lmax = -3.0e38f, lden = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
        auto head = [&](const kpn_view_gather& g, const kpn_ibr_view& iv) {
            const float wv = RSUB(kpn_fast_exp(RMUL(ani, RSUB(g.rd[3], 1.0f))), emin) / RADD(esum, 1e-8f);
            float xin[20];
#pragma unroll
            for (int i = 0; i < 19; ++i) xin[i] = i < 16 ? iv.xb0[i] : iv.xb1[i - 16];
            xin[19] = 0.0f;
            kpn_f32x16 a[2] = {base[0], base[1]};
            kpn_fuse_layer_regs<F16, SEG_BL_0B, 20, 2>(wl, lane, xin, a);  // base_layer.0 (x part)
            kpn_f32x16 xa[1];
            kpn_load_bias<1>(wl + W::boff(SEG_BL_1), h, xa);
            kpn_fuse_layer<F16, SEG_BL_1, 32, 1>(wl, lane, [&](auto gi, float (&x)[4]) {  // base_layer.2
                constexpr int gq = decltype(gi)::value;
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = kpn_elu(a[gq / 4][(gq % 4) * 4 + i]);
            }, xa);
            float x[16], tin[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { x[r] = kpn_elu(xa[0][r]); tin[r] = x[r] * wv; }  // :1292-1294
            kpn_f32x16 va[1], vb[1];
            kpn_load_bias<1>(wl + W::boff(SEG_V1_0), h, va);
            kpn_fuse_layer_regs<F16, SEG_V1_0, 16, 1>(wl, lane, tin, va);  // vis_layer1.0
#pragma unroll
            for (int r = 0; r < 16; ++r) tin[r] = kpn_elu(va[0][r]);
            kpn_load_bias<1>(wl + W::boff(SEG_V1_1), h, vb);
            kpn_fuse_layer_regs<F16, SEG_V1_1, 16, 1>(wl, lane, tin, vb);  // vis_layer1.2 rows 0..31 (res)
            const float visr = kpn_elu(kpn_row_dot(wl + W::row(ROW_V1_VIS), h, tin));   // row 32 (vis)
            const float sv = kpn_sigmoid(visr);
#pragma unroll
            for (int r = 0; r < 16; ++r) { x[r] = x[r] + kpn_elu(vb[0][r]); tin[r] = x[r] * sv; }  // :1295-1297 (mask = 1)
            kpn_load_bias<1>(wl + W::boff(SEG_V2_0), h, va);
            kpn_fuse_layer_regs<F16, SEG_V2_0, 16, 1>(wl, lane, tin, va);  // vis_layer2.0
#pragma unroll
            for (int r = 0; r < 16; ++r) tin[r] = kpn_elu(va[0][r]);
            const float vis = kpn_sigmoid(kpn_row_dot(wl + W::row(ROW_V2_1), h, tin));  // vis_layer2.2 + Sigmoid
            float oin[20];  // out_layer.0 input [x32 | vis | ray_diff4]  (:1300)
#pragma unroll
            for (int r = 0; r < 16; ++r) oin[r] = x[r];
            oin[16] = h ? g.rd[0] : vis;
            oin[17] = h ? g.rd[2] : g.rd[1];
            oin[18] = h ? 0.0f : g.rd[3];
            oin[19] = 0.0f;
            kpn_load_bias<1>(wl + W::boff(SEG_O_0), h, va);
            kpn_fuse_layer_regs<F16, SEG_O_0, 20, 1>(wl, lane, oin, va);
            float o8[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) o8[r] = kpn_elu(va[0][r]);
            kpn_load_bias<1>(wl + W::boff(SEG_O_1), h, va);
            kpn_fuse_layer_regs<F16, SEG_O_1, 8, 1>(wl, lane, o8, va);
            // out_layer.2 has 8 outputs: rows 0..3 in registers 0..3 of the h = 0 lanes, rows 4..7 in those of the h = 1 lanes; the other
            // twelve registers of the block are padding (zero weights in out_layer.4's row vector): four ELUs, not sixteen
            float o4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o4[r] = kpn_elu(va[0][r]);
            const float logit = kpn_row_dot4(wl + W::row(ROW_O_2), h, o4);  // out_layer.4
            const float nmax = fmaxf(lmax, logit);
            const float sc_old = kpn_fast_exp(lmax - nmax), pn = kpn_fast_exp(logit - nmax);
            lden = lden * sc_old + pn;
            c0 = c0 * sc_old + g.rgb[0] * pn; c1 = c1 * sc_old + g.rgb[1] * pn; c2 = c2 * sc_old + g.rgb[2] * pn;
            lmax = nmax;
        };
        // (fetching the next view's gather record ahead of the current view's head measured no gain: 23.50 vs 23.43 ms per frame)
        // (touching the NEXT tile's block of the scratch — one dword per 128-B line — under the last view's head, so that its
        // lines are in L2 when the next tile starts: 23.93-23.97 vs 23.94-24.02 ms per frame, no gain; with EVERY tile reading tile 0's
        // block, i.e. no HBM misses at all, the frame is 0.9 ms faster: profiles/r04_z_ab_experiments.txt)
#pragma unroll 1
        for (int v = 0; v < V; ++v) {
            if (!((keep >> v) & 1u)) continue;  // logit -1e9 (masked_fill, :1300): softmax weight exactly 0
            kpn_gather_view(scr + lay.rec(t_scr, v) * 64, l_scr, h, gv);
            kpn_encode_view<F16>(wl, lane, h, gv, lat0, iv);
            head(gv, iv);
        }
        KPN_FUSE_STAMP(5);
        const float r0 = c0 / lden, r1 = c1 / lden, r2 = c2 / lden;
        if constexpr (PH == 2) {
            if (h == 0 && ci_raw < nlive_src) { float* o = out + n * 5; o[2] = r0; o[3] = r1; o[4] = r2; }
            if (batch.bad != nullptr && batch.cond != KPN_RUN_IF_UNSAFE) {
                const float chk = fabsf(r0) + fabsf(r1) + fabsf(r2);
                if (__ballot(h == 0 && ci_raw < nlive_src && !(chk < 3.0e38f)) != 0ull && lane == 0) atomicOr(batch.bad, 1);
            }
            return;
        }
        if (h == 0 && ci_raw < count) {
            float* o = out + n * 5;
            if (mode == 1) {  // eval_func with mask = 1 (model.py:981-996)
                if (ps.noise) rad = RADD(rad, RMUL(ps.noise[n], ps.noise_std));
                o[0] = fmaxf(rad, 0.0f); o[1] = sdf_raw;
            }
            else { o[0] = sdf_raw; o[1] = rad; }
            o[2] = r0; o[3] = r1; o[4] = r2;
        }
        // The range guard (kpn_field_shared.h): an operand of this kernel or of the rows kernel beyond fp16's range has become a
        // NaN by now (looked for BEFORE the relu above, which would drop it).  One flag per batch; the fp32-range kernels
        // launched behind this one evaluate the batch again when it is set.
        if (batch.bad != nullptr && batch.cond != KPN_RUN_IF_UNSAFE) {
            const float chk = fabsf(sdf_raw) + fabsf(rad) + fabsf(r0) + fabsf(r1) + fabsf(r2);
            if (__ballot(h == 0 && ci_raw < count && !(chk < 3.0e38f)) != 0ull && lane == 0) atomicOr(batch.bad, 1);
        }
    };
    // The NEXT ticket is drawn while the current tiles are computed: the atomic's round trip (3.7 k cycles per tile in the round-3
    // phase counts, in front of the dependent row loads) leaves the critical path.  A ticket names CH consecutive tiles:
    // same-address atomics retire at about one per 12 ns on this chip (measured, round 6: a pass A drawing one ticket and making
    // one list reservation per tile took 2.0 ms per launch for 85 k tiles — 170 k atomics on one cache line — and 0.53 ms with a
    // quarter of the tickets and one reservation per 256 slots; the fused kernel on a hull that is 82 % empty, i.e. mostly short-path
    // tiles: frame 18.5 -> 17.1 ms with two tiles per ticket).
    constexpr int CH = PHASE == 1 ? 4 : 2;
    int next_ticket = 0;
    if (lane == 0) next_ticket = atomicAdd(ticket, 1);
    int cur = 0, sub = CH;
    for (;;) {
        KPN_FUSE_STAMP(7);
        if (sub == CH) {
            cur = __shfl(next_ticket, 0);
            sub = 0;
            if (cur * CH < ntiles && lane == 0) next_ticket = atomicAdd(ticket, 1);
        }
        const int t = cur * CH + sub;       // tile relative to the batch = its slot in the row scratch (PHASE 2: tile of the live list)
        ++sub;
        if (t >= ntiles) break;
        tile(std::integral_constant<int, PHASE>{}, t, live, nlive);
    }
    if constexpr (PHASE == 1) {
        if (qn > 0) {                       // what is left in the queue
            KPN_WAVE_SYNC();
            int base = 0;
            if (lane == 0) base = atomicAdd(tickets + 2, qn);
            base = __shfl(base, 0);
            for (int i = lane; i < qn; i += 64) live[base + i] = myq[i];
        }
    }
    if (PHASE != 2 && lane == 0 && stat_listed != 0) {
        atomicAdd(&kpn_density_counts[0], (unsigned long long)stat_listed);
        atomicAdd(&kpn_density_counts[1], (unsigned long long)stat_live);
    }
}

__global__ __launch_bounds__(512, 2) void k_fuse_color(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                                       const int* __restrict__ list, const int* __restrict__ count_ptr,
                                                       int* __restrict__ tickets, const float* __restrict__ xscr, int mode,
                                                       int /*unused*/, float* __restrict__ out, kpn_batch batch, int zero_skip) {
    kpn_fuse_color_body<false>(sc, ps, wp, list, count_ptr, tickets, xscr, mode, nullptr, out, batch, zero_skip);
}
// the same per-point kernel with its weights as two fp16 pieces per value on v_mfma_f32_32x32x16_f16 (the default)
__global__ __launch_bounds__(512, 2) void k_fuse_color_h(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                                         const int* __restrict__ list, const int* __restrict__ count_ptr,
                                                         int* __restrict__ tickets, const float* __restrict__ xscr, int mode,
                                                         int /*unused*/, float* __restrict__ out, kpn_batch batch, int zero_skip) {
    kpn_fuse_color_body<true>(sc, ps, wp, list, count_ptr, tickets, xscr, mode, nullptr, out, batch, zero_skip);
}
// ... and with the three views of the shipped configuration unrolled (launched when V == 3 and every view is kept)
__global__ __launch_bounds__(512, 2) void k_fuse_color_h3(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                                          const int* __restrict__ list, const int* __restrict__ count_ptr,
                                                          int* __restrict__ tickets, const float* __restrict__ xscr, int mode,
                                                          int /*unused*/, float* __restrict__ out, kpn_batch batch, int zero_skip) {
    kpn_fuse_color_body<true, 3>(sc, ps, wp, list, count_ptr, tickets, xscr, mode, nullptr, out, batch, zero_skip);
}
// Density first (PHASE above; render passes on the POOL layout with the two-fp16-piece per-point arithmetic).  Pass A:
__global__ __launch_bounds__(512, 2) void k_density_h(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                                      const int* __restrict__ list, const int* __restrict__ count_ptr,
                                                      int* __restrict__ tickets, float* xscr, int* __restrict__ live,
                                                      float* __restrict__ out, kpn_batch batch) {
    kpn_fuse_color_body<true, 0, 1>(sc, ps, wp, list, count_ptr, tickets, xscr, 1, live, out, batch, 0);
}
// pass B: any V / the shipped V = 3 with every view kept
__global__ __launch_bounds__(512, 2) void k_colour_h(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                                     const int* __restrict__ list, const int* __restrict__ count_ptr,
                                                     int* __restrict__ tickets, const float* __restrict__ xscr, int* __restrict__ live,
                                                     float* __restrict__ out, kpn_batch batch) {
    kpn_fuse_color_body<true, 0, 2>(sc, ps, wp, list, count_ptr, tickets, xscr, 1, live, out, batch, 0);
}
__global__ __launch_bounds__(512, 2) void k_colour_h3(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                                      const int* __restrict__ list, const int* __restrict__ count_ptr,
                                                      int* __restrict__ tickets, const float* __restrict__ xscr, int* __restrict__ live,
                                                      float* __restrict__ out, kpn_batch batch) {
    kpn_fuse_color_body<true, 3, 2>(sc, ps, wp, list, count_ptr, tickets, xscr, 1, live, out, batch, 0);
}
// ---------------------------------------------------------------------------------------------
// Lane-map self test: D = A(32x2) * B(2x32) with asymmetric operands, written out row-major.
__global__ void k_selftest_mfma(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ Dm) {
    const int lane = threadIdx.x & 63;
    const float a = A[(lane & 31) * 2 + (lane >> 5)];
    const float b = B[(lane >> 5) * 32 + (lane & 31)];
    kpn_f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) Dm[KPN_ROWMAP(r, lane >> 5) * 32 + (lane & 31)] = c[r];
}
