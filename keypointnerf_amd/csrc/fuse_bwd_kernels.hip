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
// The colour head's reverse (ibr_compress_gfeat, IBRRenderingHead, d feat_tex) is NOT built yet: gradients of the
// r,g,b outputs are rejected by the C ABI (see kpn_query_backward).
#include "kpn_device.h"

struct kpn_fuse_bwd_bufs {
    float* Xp;    // [points][128] pooled (mean64 | var64)
    float* Xh0;   // [points][64]  softplus(layers2.0)
    float* Xh1;   // [points][64]  softplus(layers2.1)
    float* D20;   // [points][64]  dA of layers2.0
    float* D21;   // [points][64]  dA of layers2.1
    float* D22;   // [points][2]   d [sdf_raw, rad]
    float* dxrows;  // [rows][64]  d x_view
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
        float pwsum = 0.0f;
        for (int v = 0; v < V; ++v)
            if ((keep >> v) & 1u) pwsum = KADD(pwsum, rows[((size_t)v * KPN_ROW_SLABS + 8) * 64 + p].w);
        float pooled[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) pooled[i] = 0.0f;
        for (int pass = 0; pass < 2; ++pass)
            for (int v = 0; v < V; ++v) {
                if (!((keep >> v) & 1u)) continue;
                const float4* src = rows + ((size_t)v * KPN_ROW_SLABS) * 64;
                const float pw = src[8 * 64 + p].w / KADD(pwsum, 1e-6f);
#pragma unroll
                for (int q4 = 0; q4 < 8; ++q4) {
                    const float4 x = src[q4 * 64 + lane];
                    const float xe[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * q4 + e;
                        if (pass == 0) pooled[i] = KADD(pooled[i], KMUL(pw, xe[e]));
                        else { const float d = KSUB(xe[e], pooled[i]); pooled[32 + i] = KADD(pooled[32 + i], KMUL(pw, KMUL(d, d))); }
                    }
                }
            }
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
            if (ps.noise) rad = KADD(rad, KMUL(ps.noise[n], ps.noise_std));
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
        // ---- pooling reverse: dpool[b] (b < 2) = d mean, dpool[2 + b] = d var, same lane-register layout as pooled ----
        float sres[32];  // sum_u pw_u (x_u - mean)
#pragma unroll
        for (int i = 0; i < 32; ++i) sres[i] = 0.0f;
        for (int v = 0; v < V; ++v) {
            if (!((keep >> v) & 1u)) continue;
            const float4* src = rows + ((size_t)v * KPN_ROW_SLABS) * 64;
            const float pw = src[8 * 64 + p].w / KADD(pwsum, 1e-6f);
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
            const float pw = on ? src[8 * 64 + p].w / KADD(pwsum, 1e-6f) : 0.0f;
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
