// Device helpers shared by the field kernels (field_kernels.hip) and the pair-tile rows kernel, which is compiled as its
// own translation unit (geo_rows_pair_tu.hip): the point source, the batches of the capped row scratch, the colour
// head's gather record.
#pragma once
#include "kpn_device.h"

struct kpn_points {
    // explicit points (kpn_query): pts/view (N,3); or ray-marched (kpn_render_rays): cam_pos(3),
    // dirs (R,3), z (R,S): point n = ray n/S at depth z[n], view = dirs[n/S]  (model.py:1057-1061)
    const float* pts;
    const float* view;
    const float* cam_pos;
    const float* dirs;
    const float* z;
    int S;
    // train-time density noise of eval_func (model.py:993-994): rad += noise[n] * noise_std; NULL in eval
    const float* noise;
    float noise_std;
};
// The row scratch holds at most `tiles_cap` tiles x V rows.  A pass whose valid list is longer is evaluated in
// nb = ceil(ntiles / tiles_cap) batches (k_geo_rows + k_fuse_color per batch, the scratch reused); the host cannot know
// nb without a sync, so it launches the worst-case number of batches and the surplus ones return at once.  Tiles are
// split evenly: batch b owns [ntiles*b/nb, ntiles*(b+1)/nb).
//
// Range guard of the two-fp16-piece kernels (kpn_common.h kpn_f16_inputs_unsafe): `cond` says when a launch does its work —
//   KPN_RUN_ALWAYS   : a kernel in fp32's exponent range (or the caller chose to run without the guard);
//   KPN_RUN_IF_SAFE  : a two-fp16-piece kernel: stands aside when the weights or the maps are beyond fp16's range;
//   KPN_RUN_IF_UNSAFE: the fp32-range kernels launched BEHIND them for the same batch: work only when those stood aside or when
//                      the per-point kernel met a non-finite result (`bad` = the batch's flag in the pass's counter block; an
//                      overflowed operand becomes NaN in every accumulator it touches and no activation of these kernels turns a
//                      NaN into a number, so it reaches the per-point outputs).  Otherwise they return at once (3-4 us).
// Never a NaN where the reference is finite, and no host synchronisation to get there.
enum { KPN_RUN_ALWAYS = 0, KPN_RUN_IF_SAFE = 1, KPN_RUN_IF_UNSAFE = 2 };
// redone: counter of batches evaluated again (diagnostic); pool: layout below; clk (measurement hook, may be NULL): two shader-clock
// stamps (s_memtime) of the launch's first workgroup, entry and last work item done — with the launch's HIP-event time they give
// the clock the chip actually sustained under this kernel (kpn_profile_collect3)
struct kpn_batch { int index, tiles_cap; int cond; int* bad; int* redone; int pool; unsigned long long* clk; };

// Two layouts of the scratch between the rows kernel, k_row_records and the per-point kernel, in slabs of [64 lanes] float4:
//   ROWS (pool = 0): per (tile, view) KPN_ROW_SLABS = 10 slabs: 0..7 the view's 64-vector (a lane's 32 registers, block b = slab / 4),
//         8, 9 the colour head's gather record.  The per-point kernel pools over the views.  What a backward pass reads again.
//   POOL (pool = 1, the render / query passes with the pair-tile rows kernels): the rows kernel walks a tile pair through ALL its views
//         and pools on the fly (PoolModule's weighted mean / variance over views, reference src/utils.py:612-647,722-748, as a
//         weighted Welford update per view in LDS), so only the pooled 128-vector reaches memory: per TILE
//         16 + 2 V slabs: 0..7 mean, 8..15 variance (same register order), then 2 slabs of gather record per view.
//         V = 3: 22 KB per tile instead of 30; V = 10: 36 instead of 100.
struct kpn_tile_layout {
    int pool, V, ts;
    __device__ __forceinline__ kpn_tile_layout(int pool_, int V_) : pool(pool_), V(V_), ts(pool_ ? 16 + 2 * V_ : V_ * KPN_ROW_SLABS) {}
    __device__ __forceinline__ size_t tile(int t) const { return (size_t)t * ts; }                                    // first slab of tile t
    __device__ __forceinline__ size_t row(int t, int v) const { return (size_t)t * ts + (size_t)v * KPN_ROW_SLABS; }   // ROWS only
    __device__ __forceinline__ size_t rec(int t, int v) const { return (size_t)t * ts + (pool ? 16 + 2 * v : v * KPN_ROW_SLABS + 8); }
};
__host__ __device__ inline int kpn_tile_slabs(int pool, int V) { return pool ? 16 + 2 * V : V * KPN_ROW_SLABS; }
__device__ __forceinline__ bool kpn_batch_gate(const kpn_batch& b, const kpn_scene_dev& sc, const float* __restrict__ wp) {
    if (b.cond == KPN_RUN_ALWAYS) return true;
    const bool unsafe = kpn_f16_inputs_unsafe(sc, wp);
    if (b.cond == KPN_RUN_IF_SAFE) return !unsafe;
    return unsafe || (b.bad != nullptr && *b.bad != 0);
}
__device__ __forceinline__ bool kpn_batch_range(const kpn_batch& b, int ntiles, int& t0, int& t1) {
    const int nb = (ntiles + b.tiles_cap - 1) / b.tiles_cap;
    if (b.index >= nb) return false;
    t0 = (int)((int64_t)ntiles * b.index / nb);
    t1 = (int)((int64_t)ntiles * (b.index + 1) / nb);
    return true;
}

template <bool S = false>
__device__ __forceinline__ void kpn_get_point(const kpn_points& ps, int64_t n, float (&P)[3], float (&D)[3]) {
    if (ps.pts) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { P[k] = ps.pts[n * 3 + k]; D[k] = ps.view[n * 3 + k]; }
    } else {
        const int64_t r = n / ps.S;
        const float zz = ps.z[n];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            D[k] = ps.dirs[r * 3 + k];
            P[k] = kpn_add<S>(ps.cam_pos[k], kpn_mul<S>(D[k], zz));
        }
    }
}

// The same in two steps, so that the loads of the NEXT work item's points can be in flight while the current one is computed:
// kpn_point_fetch issues them, kpn_point_finish forms P and D.
struct kpn_point_raw { float a[3], b[3], z; };
__device__ __forceinline__ void kpn_point_fetch(const kpn_points& ps, int64_t n, kpn_point_raw& r) {
    if (ps.pts) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { r.a[k] = ps.pts[n * 3 + k]; r.b[k] = ps.view[n * 3 + k]; }
        r.z = 0.0f;
    } else {
        const int64_t ray = n / ps.S;
        r.z = ps.z[n];
#pragma unroll
        for (int k = 0; k < 3; ++k) { r.b[k] = ps.dirs[ray * 3 + k]; r.a[k] = 0.0f; }
    }
}
// S: strict arithmetic (separate multiply and add, kpn_device.h) — the point as k_mask_compact and the reference form it, bit for bit
template <bool S = false>
__device__ __forceinline__ void kpn_point_finish(const kpn_points& ps, const kpn_point_raw& r, float (&P)[3], float (&D)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        D[k] = r.b[k];
        P[k] = ps.pts ? r.a[k] : kpn_add<S>(ps.cam_pos[k], kpn_mul<S>(r.b[k], r.z));
    }
}

// Boundary-smooth pooling weight of one view (model.py:752-759, mask == 1), un-normalised
__device__ __forceinline__ float kpn_pix_weight(const kpn_proj& q) {
    const float c3[3] = {RADD(RMUL(0.5f, q.xn), 0.5f), RADD(RMUL(0.5f, q.yn), 0.5f), RADD(RMUL(0.5f, q.zn), 0.5f)};
    float w3[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float d = fminf(c3[i], RSUB(1.0f, c3[i]));
        w3[i] = kpn_sigmoid(RMUL(5.0f, RSUB(d / 0.1f, 1.0f)));
    }
    return RMUL(RMUL(w3[0], w3[1]), w3[2]);
}

// the same weight for the pooling inside the rows kernels (POOL layout): reciprocals by v_rcp_f32 (1 ulp) instead of IEEE divisions
// (ten instructions each, twelve of them per point pair and view in a VALU-only stretch); the weights enter a weighted mean
__device__ __forceinline__ float kpn_pix_weight_fast(const kpn_proj& q) {
#ifndef KPN_SIMT_EMU
    const float c3[3] = {RADD(RMUL(0.5f, q.xn), 0.5f), RADD(RMUL(0.5f, q.yn), 0.5f), RADD(RMUL(0.5f, q.zn), 0.5f)};
    float w = 1.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float d = fminf(c3[i], RSUB(1.0f, c3[i]));
        w *= __builtin_amdgcn_rcpf(1.0f + kpn_fast_exp(-RMUL(5.0f, RSUB(d * 10.0f, 1.0f))));
    }
    return w;
#else
    return kpn_pix_weight(q);
#endif
}

// the colour head's per-(point,view) gather record (query_color, model.py:806-832) in its two parts: A = [r,g,b, pooling
// weight | ray_diff direction(3), dot] (held by the h=0 lanes of a row), B = the 8 texture channels (model.py:818; h=1 lanes)
// S = true (k_row_records): the tap positions in the source IMAGE and the texture map are formed with strict arithmetic from a
// strictly formed point and projection, i.e. bit-identical to the reference's (and the oracle's) pixel coordinates.  With the free-to-
// contract flavour a coordinate differs by an ulp now and then — 2.4e-4 px in a 4096-px image, which on configs[4]'s white-noise
// source images (neighbouring pixels differ by 0.3 on average) moved the blended colour of 2 rays in 9,216 by 1.2e-4: above the bar,
// with nothing for the oracle's conditioning probe to find (round 5, bench.py secondary.configs4_full.parity).
template <bool S = false>
__device__ __forceinline__ void kpn_row_record_a(const kpn_scene_dev& sc, const float* __restrict__ tb, int v, const kpn_proj& q,
                                                 const float (&P)[3], const float (&D)[3], float4& rec0, float4& rec1) {
    const kpn_taps ti = kpn_make_taps<S>(q.xn, q.yn, sc.H, sc.W);
    const float4 c = kpn_tap4(sc.rgbm + (size_t)v * sc.H * sc.W * 4, 4, 0, ti);  // model.py:806
    rec0 = make_float4(c.x, c.y, c.z, kpn_pix_weight(q));
    const float* cp = tb + KPN_TBL_CPOS;                                          // model.py:823-832
    float cr[3] = {RSUB(P[0], cp[0]), RSUB(P[1], cp[1]), RSUB(P[2], cp[2])};
    const float nrm = fmaxf(sqrtf(kpn_dot3(cr[0], cr[1], cr[2], cr[0], cr[1], cr[2])), 1e-12f);
    cr[0] = cr[0] / nrm; cr[1] = cr[1] / nrm; cr[2] = cr[2] / nrm;
    const float r0 = RSUB(D[0], cr[0]), r1 = RSUB(D[1], cr[1]), r2 = RSUB(D[2], cr[2]);
    const float rc = fmaxf(sqrtf(kpn_dot3(r0, r1, r2, r0, r1, r2)), 1e-6f);
    rec1 = make_float4(r0 / rc, r1 / rc, r2 / rc, kpn_dot3(cr[0], cr[1], cr[2], D[0], D[1], D[2]));
}
template <bool S = false>
__device__ __forceinline__ void kpn_row_record_b(const kpn_scene_dev& sc, int v, const kpn_proj& q, float4& rec0, float4& rec1) {
    const kpn_taps tt = kpn_make_taps<S>(q.xn, q.yn, sc.th, sc.tw);
    const float* tx = sc.tex + (size_t)v * sc.th * sc.tw * 8;
    rec0 = kpn_tap4(tx, 8, 0, tt);
    rec1 = kpn_tap4(tx, 8, 4, tt);
}
// both parts by the lane layout of the row scratch: h = 0 lanes part A, h = 1 lanes part B (k_geo_rows, rows mode 0).  The point
// and its projection are formed again HERE with strict arithmetic (see kpn_row_record_a): the records of the three rows kernels are
// the same bits, whatever flavour the calling kernel uses for its own taps.
__device__ __forceinline__ void kpn_row_record(const kpn_scene_dev& sc, const kpn_points& ps, int64_t n, const float* __restrict__ tb, int v, int h,
                                               float4& rec0, float4& rec1) {
    float P[3], D[3];
    kpn_get_point<true>(ps, n, P, D);
    const kpn_proj q = kpn_project<true>(tb, P[0], P[1], P[2], sc);
    if (h == 0) kpn_row_record_a<true>(sc, tb, v, q, P, D, rec0, rec1);
    else kpn_row_record_b<true>(sc, v, q, rec0, rec1);
}
