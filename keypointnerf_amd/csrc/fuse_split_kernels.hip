// fuse_split_kernels.hip — EXPERIMENT kernels (A/B knobs KPN_FUSE_SPLIT=1|2, off by default; DESIGN.md sections 9.3, 9.4):
// k_fuse_color as two kernels, and the colour head over a second compacted list of the points with density > 0.
// Same arithmetic in the same order as k_fuse_color (the text of its two halves): bit-identical outputs, tested on the
// emulator and on the MI355X.  Measured on the bench frame they LOSE to the fused kernel (52.3 ms fused; 53.7 split;
// 57.5 split + compaction with every point live; with a density that is 0 in most of the hull: 44.0 fused with its
// tile-level short path, 45.4 split + compaction), which is why k_fuse_color stays the product path.
#include "kpn_device.h"

// ---------------------------------------------------------------------------------------------
// k_fuse_color split in two (KPN_FUSE_SPLIT=1, experiment of DESIGN.md section 9.3): the view pooling + layers2 + compress
// half streams the 64-vectors (256 B per row) and needs few registers; the colour-head half needs neither the pooled
// vector nor layers2's accumulators.  Same arithmetic, same operand order: outputs are bit-identical to k_fuse_color's.
//   k_pool_geo   : pooling, layers2 -> out[.,0..1] (mode 0: sdf_raw, rad; mode 1: eval_func), compress -> lat (x' rows 0..15 of a lane)
//   k_color_head : IBR head from the gather records + lat -> out[.,2..4]
// lat scratch: [tile relative to the batch][4 float4][64 lanes].
#ifndef KPN_SPLIT_OCC
#define KPN_SPLIT_OCC 3
#endif
__global__ __launch_bounds__(512, KPN_SPLIT_OCC) void k_pool_geo(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                                     const int* __restrict__ list, const int* __restrict__ count_ptr,
                                                     int* __restrict__ tickets, const float* __restrict__ xscr, int mode,
                                                     float* __restrict__ lat, float* __restrict__ out, kpn_batch batch,
                                                     int* __restrict__ count2, int* __restrict__ list2) {
    const int lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int count = *count_ptr;
    int t0, t1;
    if (!kpn_batch_range(batch, (count + KPN_TILE - 1) / KPN_TILE, t0, t1)) return;   // before the LDS staging
    const int ntiles = t1 - t0;
    const int V = sc.V;
    // all weights of this kernel live in LDS for the lifetime of the (persistent) workgroup;
    // wl is biased so that the packed-buffer offsets (kpn_seg_woff etc.) index it directly
    __shared__ __attribute__((aligned(16))) float wlds[kpn_seg_woff(SEG_RE_0) - kpn_k2_base()];
    kpn_stage_lds_range(wp, wlds, kpn_k2_base(), kpn_seg_woff(SEG_RE_0) - kpn_k2_base(), SEG_G2_0, SEG_RE_0);
    __syncthreads();
    const float* wl = wlds - kpn_k2_base();

    (void)wave; (void)nwaves;
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(tickets + 1, 1);
        t = __shfl(t, 0);
        if (t >= ntiles) break;             // t: tile relative to the batch = its slot in the row scratch
        const int ci_raw = (t0 + t) * KPN_TILE + p;
        const int ci = ci_raw < count ? ci_raw : count - 1;
        const int64_t n = list[ci];

        // ---- pooled mean / var over views of the 64-vector ----
        const float4* rows = reinterpret_cast<const float4*>(xscr) + ((size_t)t * V * KPN_ROW_SLABS) * 64;
        const uint32_t keep = sc.keep;  // train-time view dropout (all ones in eval): weights of dropped views are 0
        float pwsum;
        float pooled[64];  // K-steps 0..31 = mean (block b, reg r), 32..63 = var
        KPN_POOL_VIEWS(rows, V, keep, lane, p, pwsum, pooled);
        // ---- layers2: 128 -> 64 -> 64 -> 2 (utils.py:577-587), activations applied lazily ----
        float sdf_raw, rad;
        {
            kpn_f32x16 h0[2], h1[2], o2[1];
            kpn_load_bias<2>(wl + kpn_seg_boff(SEG_G2_0), h, h0);
            kpn_mfma_layer_regs<64, 2, 4, 1>(wl + kpn_seg_woff(SEG_G2_0), lane, pooled, h0);
            kpn_load_bias<2>(wl + kpn_seg_boff(SEG_G2_1), h, h1);
            kpn_mfma_layer<32, 2, 4, 1>(wl + kpn_seg_woff(SEG_G2_1), lane, [&](auto gi, float (&x)[4]) {
                constexpr int g = decltype(gi)::value;
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = kpn_softplus100(h0[g / 4][(g % 4) * 4 + i]);
            }, h1);
            kpn_load_bias<1>(wl + kpn_seg_boff(SEG_G2_2), h, o2);
            kpn_mfma_layer<32, 1, 4, 1>(wl + kpn_seg_woff(SEG_G2_2), lane, [&](auto gi, float (&x)[4]) {
                constexpr int g = decltype(gi)::value;
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = kpn_softplus100(h1[g / 4][(g % 4) * 4 + i]);
            }, o2);
            sdf_raw = o2[0][0];  // rows 0,1 live in regs 0,1 of the h=0 lanes
            rad = o2[0][1];
        }
        // ---- ibr_compress_gfeat 128 -> 24 (model.py:819), rows already in x' order ----
        float lat0[16];
        {
            kpn_f32x16 acc[1];
            kpn_load_bias<1>(wl + kpn_seg_boff(SEG_CMP), h, acc);
            kpn_mfma_layer_regs<64, 1, 4, 1>(wl + kpn_seg_woff(SEG_CMP), lane, pooled, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) lat0[r] = acc[0][r];
        }
        {
            float4* ld = reinterpret_cast<float4*>(lat) + (size_t)t * 4 * 64 + lane;
#pragma unroll
            for (int k = 0; k < 4; ++k) ld[k * 64] = make_float4(lat0[4 * k], lat0[4 * k + 1], lat0[4 * k + 2], lat0[4 * k + 3]);
        }
        if (mode == 1 && ps.noise && h == 0 && ci_raw < count) rad = RADD(rad, RMUL(ps.noise[n], ps.noise_std));
        if (list2) {
            // second compaction: the points whose colour can reach the image (density > 0; everything in a raw query)
            const int is_live = h == 0 && ci_raw < count && (mode != 1 || rad > 0.0f);
            const unsigned long long m = __ballot(is_live);
            int base = 0;
            if (lane == 0 && m) base = atomicAdd(count2, __popcll(m));
            base = __shfl(base, 0);
            if (is_live) list2[base + __popcll(m & ((1ull << lane) - 1ull))] = t * KPN_TILE + p;
            if (h == 0 && ci_raw < count && !is_live) { float* o = out + n * 5; o[2] = 0.0f; o[3] = 0.0f; o[4] = 0.0f; }
        }
        if (h == 0 && ci_raw < count) {
            float* o = out + n * 5;
            if (mode == 1) { o[0] = fmaxf(rad, 0.0f); o[1] = sdf_raw; }   // eval_func with mask = 1 (model.py:981-996)
            else { o[0] = sdf_raw; o[1] = rad; }
        }
    }
}

// COMPACT: the head runs over list2, the (relative tile, lane) addresses of the batch's points with density > 0 that
// k_pool_geo collected (KPN_FUSE_SPLIT=2): a lane's point data — gather records, latent, parked x' — then sit at a per-lane
// (tile tq, lane laneq) instead of (t, lane); the weight side of every MFMA is unchanged.
template <bool COMPACT>
__global__ __launch_bounds__(512, KPN_SPLIT_OCC) void k_color_head(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                                       const int* __restrict__ list, const int* __restrict__ count_ptr,
                                                       int* __restrict__ tickets, const float* __restrict__ xscr,
                                                       int park_x, const float* __restrict__ lat, float* __restrict__ out,
                                                       kpn_batch batch, const int* __restrict__ count2_ptr,
                                                       const int* __restrict__ list2) {
    const int lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int count = *count_ptr;
    int t0, t1;
    if (!kpn_batch_range(batch, (count + KPN_TILE - 1) / KPN_TILE, t0, t1)) return;   // before the LDS staging
    const int count2 = COMPACT ? *count2_ptr : 0;
    const int ntiles = COMPACT ? (count2 + KPN_TILE - 1) / KPN_TILE : t1 - t0;
    if (COMPACT && count2 == 0) return;
    const int V = sc.V;
    // all weights of this kernel live in LDS for the lifetime of the (persistent) workgroup;
    // wl is biased so that the packed-buffer offsets (kpn_seg_woff etc.) index it directly
    __shared__ __attribute__((aligned(16))) float wlds[kpn_fwd_floats() - kpn_seg_woff(SEG_RE_0)];
    kpn_stage_lds_range(wp, wlds, kpn_seg_woff(SEG_RE_0), kpn_fwd_floats() - kpn_seg_woff(SEG_RE_0), SEG_RE_0, SEG_COUNT);
    __syncthreads();
    const float* wl = wlds - kpn_seg_woff(SEG_RE_0);
    const float ani = wl[kpn_scalar_off() + 0];  // |ani_al|

    (void)wave; (void)nwaves;
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(tickets + 2, 1);
        t = __shfl(t, 0);
        if (t >= ntiles) break;             // t: tile relative to the batch = its slot in the row scratch
        int tq = t, pq = p, live_lane;
        if constexpr (COMPACT) {
            const int c2 = t * KPN_TILE + p;
            const int e = list2[c2 < count2 ? c2 : count2 - 1];   // pad lanes redo the last listed point; nothing is written
            tq = e >> 5; pq = e & 31;
            live_lane = c2 < count2;
        } else {
            live_lane = (t0 + t) * KPN_TILE + p < count;
        }
        const int laneq = (h << 5) | pq;
        const int ciq = (t0 + tq) * KPN_TILE + pq;
        const int64_t n = list[ciq < count ? ciq : count - 1];

        const float4* rows = reinterpret_cast<const float4*>(xscr) + ((size_t)tq * V * KPN_ROW_SLABS) * 64;
        const uint32_t keep = sc.keep;
        float lat0[16];
        {
            const float4* ld = reinterpret_cast<const float4*>(lat) + (size_t)tq * 4 * 64 + laneq;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 f = ld[k * 64];
                lat0[4 * k] = f.x; lat0[4 * k + 1] = f.y; lat0[4 * k + 2] = f.z; lat0[4 * k + 3] = f.w;
            }
        }
        // ---- IBR head (model.py:1267-1302) ----
        // blend weights (model.py:1287-1289): w_v = (e_v - min_v e) / (sum + 1e-8), e_v = exp(|a|(dot_v - 1))
        kpn_view_gather gv;
        kpn_ibr_view iv;
        float emin = 3.0e38f, esum = 0.0f;
        for (int pass = 0; pass < 2; ++pass)
            for (int v = 0; v < V; ++v) {
                const float dot = rows[((size_t)v * KPN_ROW_SLABS + 9) * 64 + pq].w;
                const float e = kpn_fast_exp(RMUL(ani, RSUB(dot, 1.0f)));
                if (pass == 0) emin = fminf(emin, e);  // min over ALL views (:1288)
                else if ((keep >> v) & 1u) esum = RADD(esum, RSUB(e, emin));
            }
        // fused mean/var over views of x' (utils.py:91-95): K-steps mean' (16 + 3 + pad), var' (16 + 3 + pad)
        float mv[40];
#pragma unroll
        for (int i = 0; i < 40; ++i) mv[i] = 0.0f;
        auto stats = [&](int pass, float dot, const kpn_ibr_view& iv) {
            const float wv = RSUB(kpn_fast_exp(RMUL(ani, RSUB(dot, 1.0f))), emin) / RADD(esum, 1e-8f);
#pragma unroll
            for (int i = 0; i < 19; ++i) {
                const float x = i < 16 ? iv.xb0[i] : iv.xb1[i - 16];
                if (pass == 0) mv[i] = RADD(mv[i], RMUL(x, wv));
                else { const float d = RSUB(x, mv[i]); mv[20 + i] = RADD(mv[20 + i], RMUL(wv, RMUL(d, d))); }
            }
        };
        // x' of a view is needed three times (two statistics passes, the head).  With park_x the first pass parks it
        // in slabs 0..4 of the view's row block — the 64-vector stored there is dead once it has been pooled — and the
        // later passes read it back (5 dwordx4 per lane) instead of re-running the gather and the ray encoder.
        float4* const park = const_cast<float4*>(rows) + laneq;
        auto park_store = [&](int v, const kpn_ibr_view& x) {
            float4* d = park + (size_t)v * KPN_ROW_SLABS * 64;
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k * 64] = make_float4(x.xb0[4 * k], x.xb0[4 * k + 1], x.xb0[4 * k + 2], x.xb0[4 * k + 3]);
            d[4 * 64] = make_float4(x.xb1[0], x.xb1[1], x.xb1[2], 0.0f);
        };
        auto park_load = [&](int v, kpn_ibr_view& x) {
            const float4* d = park + (size_t)v * KPN_ROW_SLABS * 64;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 f = d[k * 64];
                x.xb0[4 * k] = f.x; x.xb0[4 * k + 1] = f.y; x.xb0[4 * k + 2] = f.z; x.xb0[4 * k + 3] = f.w;
            }
            const float4 f = d[4 * 64];
            x.xb1[0] = f.x; x.xb1[1] = f.y; x.xb1[2] = f.z;
        };
        for (int pass = 0; pass < 2; ++pass)
            for (int v = 0; v < V; ++v) {
                if (!((keep >> v) & 1u)) continue;
                if (pass == 0 || !park_x) {
                    kpn_gather_view(xscr, tq, V, v, laneq, h, gv);
                    kpn_encode_view(wl, lane, h, gv, lat0, iv);
                    if (park_x) park_store(v, iv);
                    stats(pass, gv.rd[3], iv);
                } else {
                    park_load(v, iv);
                    stats(pass, rows[((size_t)v * KPN_ROW_SLABS + 9) * 64 + pq].w, iv);
                }
            }
        // view-invariant part of base_layer.0: W[:, mean|var] * [mean, var] + b
        kpn_f32x16 base[2];
        kpn_load_bias<2>(wl + kpn_seg_boff(SEG_BL_0A), h, base);
        kpn_mfma_layer_regs<40, 2, 4, 1>(wl + kpn_seg_woff(SEG_BL_0A), lane, mv, base);

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
            kpn_mfma_layer_regs<20, 2, 4, 1>(wl + kpn_seg_woff(SEG_BL_0B), lane, xin, a);  // base_layer.0 (x part)
            kpn_f32x16 xa[1];
            kpn_load_bias<1>(wl + kpn_seg_boff(SEG_BL_1), h, xa);
            kpn_mfma_layer<32, 1, 4, 1>(wl + kpn_seg_woff(SEG_BL_1), lane, [&](auto gi, float (&x)[4]) {  // base_layer.2
                constexpr int gq = decltype(gi)::value;
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = kpn_elu(a[gq / 4][(gq % 4) * 4 + i]);
            }, xa);
            float x[16], tin[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { x[r] = kpn_elu(xa[0][r]); tin[r] = x[r] * wv; }  // :1292-1294
            kpn_f32x16 va[1], vb[1];
            kpn_load_bias<1>(wl + kpn_seg_boff(SEG_V1_0), h, va);
            kpn_mfma_layer_regs<16, 1, 4, 1>(wl + kpn_seg_woff(SEG_V1_0), lane, tin, va);  // vis_layer1.0
#pragma unroll
            for (int r = 0; r < 16; ++r) tin[r] = kpn_elu(va[0][r]);
            kpn_load_bias<1>(wl + kpn_seg_boff(SEG_V1_1), h, vb);
            kpn_mfma_layer_regs<16, 1, 4, 1>(wl + kpn_seg_woff(SEG_V1_1), lane, tin, vb);  // vis_layer1.2 rows 0..31 (res)
            const float visr = kpn_elu(kpn_row_dot(wl + kpn_row_off(ROW_V1_VIS), h, tin));   // row 32 (vis)
            const float sv = kpn_sigmoid(visr);
#pragma unroll
            for (int r = 0; r < 16; ++r) { x[r] = x[r] + kpn_elu(vb[0][r]); tin[r] = x[r] * sv; }  // :1295-1297 (mask = 1)
            kpn_load_bias<1>(wl + kpn_seg_boff(SEG_V2_0), h, va);
            kpn_mfma_layer_regs<16, 1, 4, 1>(wl + kpn_seg_woff(SEG_V2_0), lane, tin, va);  // vis_layer2.0
#pragma unroll
            for (int r = 0; r < 16; ++r) tin[r] = kpn_elu(va[0][r]);
            const float vis = kpn_sigmoid(kpn_row_dot(wl + kpn_row_off(ROW_V2_1), h, tin));  // vis_layer2.2 + Sigmoid
            float oin[20];  // out_layer.0 input [x32 | vis | ray_diff4]  (:1300)
#pragma unroll
            for (int r = 0; r < 16; ++r) oin[r] = x[r];
            oin[16] = h ? g.rd[0] : vis;
            oin[17] = h ? g.rd[2] : g.rd[1];
            oin[18] = h ? 0.0f : g.rd[3];
            oin[19] = 0.0f;
            kpn_load_bias<1>(wl + kpn_seg_boff(SEG_O_0), h, va);
            kpn_mfma_layer_regs<20, 1, 4, 1>(wl + kpn_seg_woff(SEG_O_0), lane, oin, va);
            float o8[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) o8[r] = kpn_elu(va[0][r]);
            kpn_load_bias<1>(wl + kpn_seg_boff(SEG_O_1), h, va);
            kpn_mfma_layer_regs<8, 1, 4, 1>(wl + kpn_seg_woff(SEG_O_1), lane, o8, va);
#pragma unroll
            for (int r = 0; r < 16; ++r) tin[r] = kpn_elu(va[0][r]);
            const float logit = kpn_row_dot(wl + kpn_row_off(ROW_O_2), h, tin);  // out_layer.4
            const float nmax = fmaxf(lmax, logit);
            const float sc_old = kpn_fast_exp(lmax - nmax), pn = kpn_fast_exp(logit - nmax);
            lden = lden * sc_old + pn;
            c0 = c0 * sc_old + g.rgb[0] * pn; c1 = c1 * sc_old + g.rgb[1] * pn; c2 = c2 * sc_old + g.rgb[2] * pn;
            lmax = nmax;
        };
        for (int v = 0; v < V; ++v) {
            if (!((keep >> v) & 1u)) continue;  // logit -1e9 (masked_fill, :1300): softmax weight exactly 0
            kpn_gather_view(xscr, tq, V, v, laneq, h, gv);
            if (park_x) park_load(v, iv);
            else kpn_encode_view(wl, lane, h, gv, lat0, iv);
            head(gv, iv);
        }
        if (h == 0 && live_lane) {
            float* o = out + n * 5;
            o[2] = c0 / lden; o[3] = c1 / lden; o[4] = c2 / lden;
        }
    }
}

