// kpn_device.h — device-side helpers shared by the ray and field kernels (gfx950, wave64).
#pragma once
#include "kpn_common.h"

// "geometry" arithmetic: separate IEEE multiply/add (no FMA contraction) so that projections, ray
// set-up, AABB tests and mask thresholds are bit-identical to the scalar restatement they are tested
// against (validity masks are discrete decisions; everything downstream depends on them).
// NOTE: HIP's __fmul_rn / __fadd_rn are plain `x * y` / `x + y` (clang's __clang_hip_math.h) and hipcc compiles device
// code with -ffp-contract=fast-honor-pragmas, so they DO get fused into v_fma_f32 wherever the optimiser sees a
// multiply feeding an add.  Observed: `cam_pos + dir * z` of the ray-marched point was contracted in k_mask_compact,
// which moved one point of 262,144 across the fg-mask threshold (reference fixture case_p, ray 3866 / sample 42).  The
// helpers below switch contraction off for their own operations; an fmul and an fadd fuse only if BOTH allow it.
__device__ __forceinline__ float kpn_mul_nofma(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float kpn_add_nofma(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float kpn_sub_nofma(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
#define KMUL(a, b) kpn_mul_nofma((a), (b))
#define KADD(a, b) kpn_add_nofma((a), (b))
#define KSUB(a, b) kpn_sub_nofma((a), (b))
// Two arithmetic flavours of the shared geometry helpers (projection, bilinear taps, point set-up):
//   STRICT = true : separate multiplies and adds — every DISCRETE decision (validity, fg-mask threshold, AABB hits,
//                   sample ordering) is taken on these values: k_mask_compact and the ray kernels;
//   STRICT = false: plain operators, free to contract — the same quantities where they only feed continuous arithmetic
//                   (tap weights and encodings inside the MFMA-bound kernels, where the extra VALU issue slots of the
//                   unfused form cost 6 % of k_geo_rows).
template <bool S> __device__ __forceinline__ float kpn_mul(float a, float b) { if constexpr (S) return kpn_mul_nofma(a, b); else return a * b; }
template <bool S> __device__ __forceinline__ float kpn_add(float a, float b) { if constexpr (S) return kpn_add_nofma(a, b); else return a + b; }
template <bool S> __device__ __forceinline__ float kpn_sub(float a, float b) { if constexpr (S) return kpn_sub_nofma(a, b); else return a - b; }
#define RMUL(a, b) ((a) * (b))
#define RADD(a, b) ((a) + (b))
#define RSUB(a, b) ((a) - (b))
template <bool S = false>
__device__ __forceinline__ float kpn_dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
    return kpn_add<S>(kpn_add<S>(kpn_mul<S>(a0, b0), kpn_mul<S>(a1, b1)), kpn_mul<S>(a2, b2));
}

__device__ __forceinline__ float kpn_sigmoid(float x) { return 1.0f / (1.0f + kpn_fast_exp(-x)); }
// Softplus(beta=100, threshold=20), reference src/utils.py:523-524.  log(1+e^t)/100: the /100 makes
// the fast exp/log's ~1e-6 relative error an absolute error < 1e-8.
__device__ __forceinline__ float kpn_softplus100(float x) {
    const float sp = kpn_log2(1.0f + kpn_exp2(x * 144.269504088896341f)) * 6.93147180559945309e-3f;  // ln2/100
    return (x * 100.0f > 20.0f) ? x : sp;
}
// the same activation on a pre-activation kept in log2 units, u = 100 log2(e) x (the packers fold the factors into the streams,
// kpn_common.h kpn_cseg_wfactor / kpn_hseg_factor): 100 log2(e) softplus(x) = log2(1 + 2^u) = max(u, 0) + log2(1 + 2^-|u|).  Keeps a
// NaN (2^-|NaN| is a NaN), never overflows (2^-|u| <= 1); beyond the reference's threshold branch the second term is below an ulp of u.
__device__ __forceinline__ float kpn_softplus_log2(float u) { return fmaxf(u, 0.0f) + kpn_log2(1.0f + kpn_exp2(-fabsf(u))); }
// sin and cos of y (radians), |y| up to a few hundred: quadrant reduction by a three-term Cody-Waite
// split of pi/2 and degree-7/8 minimax polynomials on [-pi/4, pi/4] (about 1 ulp).  Register-light,
// unlike the libm sincosf with its Payne-Hanek slow path, which made the keypoint encoding spill.
__device__ __forceinline__ void kpn_sincos(float y, float& s, float& c) {
    const float k = rintf(y * 0.636619772367581343f);
    float r = fmaf(k, -1.5703125f, y);
    r = fmaf(k, -4.837512969970703125e-4f, r);
    r = fmaf(k, -7.54978995489188216e-8f, r);
    const float r2 = r * r;
    const float sp = fmaf(r2 * r, fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float cp = fmaf(r2 * r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                          fmaf(r2, -0.5f, 1.0f));
    const int q = (int)k;
    const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}
// sin(pi z), cos(pi z) for the keypoint encoding: v_sin_f32 / v_cos_f32 take their argument in REVOLUTIONS, and z / 2 is exact.
// Measured on the MI355X (scripts/sincos_probe.hip, against float64, |z| <= 8): max abs error 1.25e-7 for the pair, 5.4e-7 / 9.7e-7
// after one / two angle doublings — the polynomial form above, which rounds z * pi first as the reference does, is at
// 4.2e-7 / 9.8e-7 / 2.0e-6 for |z| <= 2 and grows with |z|.  Against the REFERENCE (sin of the rounded fp32 argument,
// spatial.py:36-37) the hardware form is off by that argument rounding, <= 1.5e-6 at 4 pi z for |z| <= 1.5 — the same order as
// the doubling formulas' error before — for 3 instructions instead of ~28 per (point, keypoint).
__device__ __forceinline__ void kpn_sincos_pi(float z, float& s, float& c) {
#ifndef KPN_SIMT_EMU
    const float r = z * 0.5f;
    s = __builtin_amdgcn_sinf(r);
    c = __builtin_amdgcn_cosf(r);
#else
    s = (float)sin(3.14159265358979323846 * (double)z);
    c = (float)cos(3.14159265358979323846 * (double)z);
#endif
}
__device__ __forceinline__ float kpn_elu(float x) {
    // ELU(x) = x if x >= 0 else e^x - 1 (torch.nn.ELU) as compare + select.  Round 3 used one v_med3_f32 (median(x, e^x - 1, 0));
    // a median drops NaNs (v_med3_f32 returns min3 when an input is a NaN, and min3 returns the numeric operand), so an overflowed
    // fp16 operand of k_fuse_color_h — NaN in every accumulator it touches — came out of the next ELU as 0 and the point as a
    // finite, wrong colour.  A select keeps the NaN (x = NaN: the compare is false, x is returned), so it reaches the per-point
    // outputs, where the range guard looks for it (kpn_field_shared.h kpn_batch).  (A sign-bit select, v_ashrrev + v_bfi, came out
    // of hipcc as three instructions: v_ashrrev_i32, v_max_i32, v_and_or_b32.)
    const float t = kpn_fast_exp(x) - 1.0f;
    return x < 0.0f ? t : x;
}

// ---------------------------------------------------------------------------------------------
// Projection of a world point into source view `tb` (per-view table): reference src/model.py:713-729.
struct kpn_proj {
    float xn, yn, zn;  // normalised to [-1,1]
    int in;            // inside the view volume (xy within +-1.01, z >= znear)
};
template <bool S = false>
__device__ __forceinline__ kpn_proj kpn_project(const float* __restrict__ tb, float px, float py, float pz,
                                                const kpn_scene_dev& sc) {
    const float* M = tb + KPN_TBL_KRT;
    const float vx = kpn_add<S>(kpn_dot3<S>(px, py, pz, M[0], M[1], M[2]), M[3]);
    const float vy = kpn_add<S>(kpn_dot3<S>(px, py, pz, M[4], M[5], M[6]), M[7]);
    const float vz = kpn_add<S>(kpn_dot3<S>(px, py, pz, M[8], M[9], M[10]), M[11]);
    const float x = vx / vz, y = vy / vz;
    kpn_proj q;
    q.xn = kpn_sub<S>(kpn_mul<S>(2.0f, x / kpn_sub<S>((float)sc.W, 1.0f)), 1.0f);
    q.yn = kpn_sub<S>(kpn_mul<S>(2.0f, y / kpn_sub<S>((float)sc.H, 1.0f)), 1.0f);
    q.zn = kpn_sub<S>(kpn_mul<S>(2.0f, kpn_sub<S>(vz, sc.znear)) / kpn_sub<S>(sc.zfar, sc.znear), 1.0f);
    const float eps = 1e-2f;
    q.in = (q.xn >= -1.0f - eps) && (q.xn <= 1.0f + eps) && (q.yn >= -1.0f - eps) && (q.yn <= 1.0f + eps) &&
           (q.zn >= -1.0f);
    return q;
}

// Bilinear taps of F.grid_sample(bilinear, border, align_corners=True) (reference src/utils.py:74-89)
// on an h x w map: pixel offsets (y*w+x) of the 4 taps and their weights.  After the border clip the
// +1 neighbours can only leave the map when their weight is exactly 0, so they are clamped instead
// of skipped.
struct kpn_taps {
    int o00, o01, o10, o11;
    float w00, w01, w10, w11;  // nw, ne, sw, se
};
template <bool S = false>
__device__ __forceinline__ kpn_taps kpn_make_taps(float xn, float yn, int h, int w) {
    float ix = kpn_mul<S>(kpn_add<S>(xn, 1.0f) / 2.0f, (float)(w - 1));
    float iy = kpn_mul<S>(kpn_add<S>(yn, 1.0f) / 2.0f, (float)(h - 1));
    ix = fminf(fmaxf(ix, 0.0f), (float)(w - 1));
    iy = fminf(fmaxf(iy, 0.0f), (float)(h - 1));
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float x1f = fx + 1.0f, y1f = fy + 1.0f;
    kpn_taps t;
    t.w00 = kpn_mul<S>(kpn_sub<S>(x1f, ix), kpn_sub<S>(y1f, iy));
    t.w01 = kpn_mul<S>(kpn_sub<S>(ix, fx), kpn_sub<S>(y1f, iy));
    t.w10 = kpn_mul<S>(kpn_sub<S>(x1f, ix), kpn_sub<S>(iy, fy));
    t.w11 = kpn_mul<S>(kpn_sub<S>(ix, fx), kpn_sub<S>(iy, fy));
    const int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
    t.o00 = y0 * w + x0; t.o01 = y0 * w + x1; t.o10 = y1 * w + x0; t.o11 = y1 * w + x1;
    return t;
}
// 4 consecutive channels of a channels-last map (C floats per pixel) starting at channel c0
template <bool S = false>
__device__ __forceinline__ float4 kpn_tap4(const float* __restrict__ map, int C, int c0, const kpn_taps& t) {
    const float4 a = *reinterpret_cast<const float4*>(map + (size_t)t.o00 * C + c0);
    const float4 b = *reinterpret_cast<const float4*>(map + (size_t)t.o01 * C + c0);
    const float4 c = *reinterpret_cast<const float4*>(map + (size_t)t.o10 * C + c0);
    const float4 d = *reinterpret_cast<const float4*>(map + (size_t)t.o11 * C + c0);
    float4 r;  // same tap order as ATen: nw, ne, sw, se
    r.x = kpn_add<S>(kpn_add<S>(kpn_add<S>(kpn_mul<S>(a.x, t.w00), kpn_mul<S>(b.x, t.w01)), kpn_mul<S>(c.x, t.w10)), kpn_mul<S>(d.x, t.w11));
    r.y = kpn_add<S>(kpn_add<S>(kpn_add<S>(kpn_mul<S>(a.y, t.w00), kpn_mul<S>(b.y, t.w01)), kpn_mul<S>(c.y, t.w10)), kpn_mul<S>(d.y, t.w11));
    r.z = kpn_add<S>(kpn_add<S>(kpn_add<S>(kpn_mul<S>(a.z, t.w00), kpn_mul<S>(b.z, t.w01)), kpn_mul<S>(c.z, t.w10)), kpn_mul<S>(d.z, t.w11));
    r.w = kpn_add<S>(kpn_add<S>(kpn_add<S>(kpn_mul<S>(a.w, t.w00), kpn_mul<S>(b.w, t.w01)), kpn_mul<S>(c.w, t.w10)), kpn_mul<S>(d.w, t.w11));
    return r;
}

// ---------------------------------------------------------------------------------------------
// One Linear layer on the matrix cores (see kpn_common.h for the operand maps and stream layout).
//   wseg : segment base; A stream [KS/G][64][G*NOB], bias [NOB][2][16]
//   acc[ob][r]: 32-row output block ob.
// The K loop is fully unrolled (B operands are registers) but software-pipelined by hand in groups of
// G K-steps: the next group's A operands are fetched (dwordx4) and its B operands produced while the
// current group's MFMAs issue, and a scheduling barrier closes every group — left alone, the
// compiler hoists all ~1e3 weight loads of a layer to its top and spills them.
template <int N>
struct kpn_ic { static constexpr int value = N; };
template <int I, int N, class F>
__device__ __forceinline__ void kpn_static_for(F&& f) {
    if constexpr (I < N) {
        f(kpn_ic<I>{});
        kpn_static_for<I + 1, N>(f);
    }
}
#ifdef KPN_SIMT_EMU
#define KPN_SCHED_BARRIER() ((void)0)
#define KPN_SCHED_GROUP(mask, n) ((void)0)
#define KPN_PIN_POINTER(p) ((void)0)
#define KPN_FENCE_RW(v) ((void)0)
#define KPN_FENCE_R(v) ((void)0)
#else
// Empty volatile asm statements keep their program order.  "+v"(acc) after a group's MFMAs makes the
// next group's MFMAs depend on it (the DAG scheduler otherwise reorders MFMAs of different
// accumulators across the whole unrolled layer and spills their operands); "v"(w) makes the prefetched
// operands of the NEXT group be waited for here, i.e. one group (>= 1k cycles of MFMA) after issue.
#define KPN_FENCE_RW(v) asm volatile("" : "+v"(v))
#define KPN_FENCE_R(v) asm volatile("" ::"v"(v))
#define KPN_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// scheduling pipeline inside a group region: first the next group's operand fetches, then the MFMAs
#define KPN_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
// an opaque re-definition of a (wave-uniform) pointer: loads through it cannot be hoisted above this point
#define KPN_PIN_POINTER(p) asm volatile("" : "+s"(p))
#endif

template <int NOB>
__device__ __forceinline__ void kpn_load_bias(const float* __restrict__ bseg, int h, kpn_f32x16 (&acc)[NOB]) {
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        const float4* b4 = reinterpret_cast<const float4*>(bseg + (ob * 2 + h) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = b4[q];
            acc[ob][4 * q + 0] = b.x; acc[ob][4 * q + 1] = b.y; acc[ob][4 * q + 2] = b.z; acc[ob][4 * q + 3] = b.w;
        }
    }
}
// MEM = 0: the stream is in global memory (L2-resident); MEM = 1: in LDS
template <int NQ, int MEM>
__device__ __forceinline__ void kpn_load_group(const float* __restrict__ gbase, int lane, kpn_f32x4 (&w)[NQ]) {
    if constexpr (MEM == 0) {
        const kpn_gptr4 src = KPN_GLOBAL4(gbase) + lane * NQ;
#pragma unroll
        for (int q = 0; q < NQ; ++q) w[q] = src[q];
    } else {
        // LDS copy of a stream: [group][q][64 lanes] float4 (re-laid by kpn_stage_lds_streams).  ds_read_b128 is served in
        // four 16-lane groups over a 256-B bank row: consecutive lanes 16 B apart fill it exactly, whereas the global
        // layout's 32-B lane stride (NQ = 2) put two lanes of every group on each 16-B slot (2-way conflict on every read:
        // SQ_LDS_BANK_CONFLICT 3.0e8 vs SQ_INSTS_LDS 2.8e8 per two frames, profiles/r02_a_pmc_counters.txt).
        const kpn_lptr4 src = KPN_LDS4(gbase) + lane;
#pragma unroll
        for (int q = 0; q < NQ; ++q) w[q] = src[q * 64];
    }
}
// in_fn(kpn_ic<g>, float (&x)[G]) produces the B operands of K-steps [g*G, (g+1)*G)
#ifndef KPN_WDEPTH
#define KPN_WDEPTH 1   // weight groups in flight ahead of the one being multiplied (global streams); 2 measured: no gain
#endif
template <int KS, int NOB, int G, int MEM = 0, class InFn>
__device__ __forceinline__ void kpn_mfma_layer(const float* __restrict__ wseg, int lane, InFn&& in_fn,
                                               kpn_f32x16 (&acc)[NOB]) {
    static_assert(KS % G == 0, "K-steps come in whole groups");
    static_assert((G * NOB) % 4 == 0, "a lane's operands of one group are whole float4s");
    constexpr int NG = KS / G, NQ = G * NOB / 4;
    constexpr int DEPTH = (MEM == 0) ? KPN_WDEPTH : 1, NBUF = DEPTH + 1;
    kpn_f32x4 w[NBUF][NQ];
    float x[2][G];
    kpn_static_for<0, DEPTH>([&](auto di) {
        constexpr int dg = decltype(di)::value;
        if constexpr (dg < NG) {
            const float* gp = wseg + (size_t)dg * 64 * (4 * NQ);
            kpn_load_group<NQ, MEM>(gp, lane, w[dg % NBUF]);
        }
    });
    in_fn(kpn_ic<0>{}, x[0]);
    kpn_static_for<0, NG>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
        constexpr int cur = g & 1, nxt = cur ^ 1;
        constexpr int wcur = g % NBUF, wnext = (g + 1) % NBUF, wload = (g + DEPTH) % NBUF;
        if constexpr (g + DEPTH < NG) {
            const float* gp = wseg + (size_t)(g + DEPTH) * 64 * (4 * NQ);
            if constexpr (MEM == 0) KPN_PIN_POINTER(gp);
            kpn_load_group<NQ, MEM>(gp, lane, w[wload]);
        }
        if constexpr (g + 1 < NG) in_fn(kpn_ic<g + 1>{}, x[nxt]);
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
                acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[wcur][(i * NOB + ob) / 4][(i * NOB + ob) % 4], x[cur][i],
                                                               acc[ob], 0, 0, 0);

#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) KPN_FENCE_RW(acc[ob]);
        if constexpr (g + 1 < NG) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) KPN_FENCE_R(w[wnext][q]);
            if constexpr (g + DEPTH < NG)
                KPN_SCHED_GROUP(MEM == 0 ? 0x020 : 0x100, NQ);  // VMEM reads (global) / DS reads (LDS) of a later group first
            // then this group's MFMAs, each followed by a few of the VALU / transcendental instructions that
            // produce the next group's B operands, so that no VALU clump leaves the matrix pipe idle
#pragma unroll
            for (int i = 0; i < G * NOB; ++i) {
                KPN_SCHED_GROUP(0x008, 1);
                KPN_SCHED_GROUP(0x002, 3);
                KPN_SCHED_GROUP(0x400, 1);
            }
        }
        KPN_SCHED_BARRIER();
    });
}
// B operands taken from a register array: K-step s reads src[s]
template <int KS, int NOB, int G = 4, int MEM = 0, int NSRC>
__device__ __forceinline__ void kpn_mfma_layer_regs(const float* __restrict__ wseg, int lane, const float (&src)[NSRC],
                                                    kpn_f32x16 (&acc)[NOB]) {
    static_assert(NSRC >= KS, "operand array too short");
    kpn_mfma_layer<KS, NOB, G, MEM>(wseg, lane, [&](auto gi, float (&x)[G]) {
        constexpr int g = decltype(gi)::value;
#pragma unroll
        for (int i = 0; i < G; ++i) x[i] = src[g * G + i];
    }, acc);
}

// The same Linear layer on v_mfma_f32_32x32x16_f16 with two fp16 pieces per operand and three products per term set (the
// k_geo_rows_f2 arithmetic, geo_rows_pair_kernels.hip) for the per-point kernel: the stream is the LDS copy of a kpn_cseg_*
// segment (kpn_common.h), KS fp32 K-steps taken eight at a time.  in_fn has kpn_mfma_layer's signature with G = 4
// (in_fn(kpn_ic<g>, float (&x)[4]) = K-steps 4g .. 4g+3); a chunk is two such groups.  Compiler-scheduled, every instruction
// compiler-selected (no asm statement: hipcc pads every MFMA <-> VALU pair itself); two waves share a SIMD in this kernel.  3 MFMAs of 32 cycles per (chunk, block) against 8 of 64 on the fp32 pipe.
template <int KS, int NOB, int LID = -1, class InFn>
__device__ __forceinline__ void kpn_hlayer(const float* __restrict__ wseg, int lane, InFn&& in_fn, kpn_f32x16 (&acc)[NOB]) {
    constexpr int NC = (KS + 7) / 8, NG = (KS + 3) / 4;
    const kpn_lptr4 base = KPN_LDS4(wseg) + lane;
    // (a second accumulator for the lo-piece weight products of the one-block layers — two dependent chains of half the length —
    // measured 0.6 % SLOWER on the frame: the matrix pipe forwards an accumulator to the next MFMA, the chains are not what the
    // per-view heads wait for)
    kpn_static_for<0, NC>([&](auto ci) {
        constexpr int c = decltype(ci)::value;
        float x[8];
        {
            float lo[4], hi[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            in_fn(kpn_ic<2 * c>{}, lo);
            if constexpr (2 * c + 1 < NG) in_fn(kpn_ic<2 * c + 1>{}, hi);
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[i] = lo[i]; x[4 + i] = hi[i]; }
        }
        kpn_u32x4 bh, bl;
        kpn_split_f16x8(x, bh, bl);
#ifdef KPN_PRECISION_PROBE
        if (LID >= 0 && KPN_PROBE(8 + LID - SEG_G2_0, 0)) bl = kpn_u32x4{0u, 0u, 0u, 0u};
#endif
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const kpn_f32x4 ah = base[((c * NOB + ob) * 2 + 0) * 64];
            kpn_f32x4 al = base[((c * NOB + ob) * 2 + 1) * 64];
#ifdef KPN_PRECISION_PROBE
            if (LID >= 0 && KPN_PROBE(8 + LID - SEG_G2_0, 1)) al = kpn_f32x4{0.f, 0.f, 0.f, 0.f};
#endif
            // Three products (ll dropped: <= 2^-24 of a term, as in the rows kernel), in the order hh lh hl.  (Round 4: with the order
            // hh hl lh this kernel came out wrong and non-deterministic on the MI355X.  Not the order: the operand splits were asm
            // statements then, and in that build the register allocator had put their outputs into dead registers of an MFMA's
            // destination tuple still in flight — a write-after-write pair hipcc only pads when it sees the VALU instruction
            // (kpn_common.h kpn_split_f16x8, scripts/repro_asm_waw_hazard.hip).  With compiler-selected splits both orders pass the
            // soak and the GPU suite: profiles/r05_a_waw_hazard.txt.)
            acc[ob] = kpn_mfma_f16(ah, bh, acc[ob]);
            if constexpr (KPN_FUSE_F16_PRODUCTS == 4) { acc[ob] = kpn_mfma_f16(ah, bl, acc[ob]); acc[ob] = kpn_mfma_f16(al, bh, acc[ob]); acc[ob] = kpn_mfma_f16(al, bl, acc[ob]); }
            else { acc[ob] = kpn_mfma_f16(al, bh, acc[ob]); acc[ob] = kpn_mfma_f16(ah, bl, acc[ob]); }
        }
    });
}
// the same layer with B operands that are ALREADY split (one kpn_split_f16x8 per chunk, shared by several layers that consume the
// same vector: layers2.0 and ibr_compress_gfeat both take the pooled 128-vector)
template <int NC, int NOB, int LID = -1>
__device__ __forceinline__ void kpn_hlayer_presplit(const float* __restrict__ wseg, int lane, const kpn_u32x4 (&bh)[NC], const kpn_u32x4 (&bl_in)[NC],
                                                    kpn_f32x16 (&acc)[NOB]) {
    const kpn_lptr4 base = KPN_LDS4(wseg) + lane;
    kpn_static_for<0, NC>([&](auto ci) {
        constexpr int c = decltype(ci)::value;
        kpn_u32x4 blc = bl_in[c];
#ifdef KPN_PRECISION_PROBE
        if (LID >= 0 && KPN_PROBE(8 + LID - SEG_G2_0, 0)) blc = kpn_u32x4{0u, 0u, 0u, 0u};
#endif
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const kpn_f32x4 ah = base[((c * NOB + ob) * 2 + 0) * 64];
            kpn_f32x4 al = base[((c * NOB + ob) * 2 + 1) * 64];
#ifdef KPN_PRECISION_PROBE
            if (LID >= 0 && KPN_PROBE(8 + LID - SEG_G2_0, 1)) al = kpn_f32x4{0.f, 0.f, 0.f, 0.f};
#endif
            acc[ob] = kpn_mfma_f16(ah, bh[c], acc[ob]);
            if constexpr (KPN_FUSE_F16_PRODUCTS == 4) { acc[ob] = kpn_mfma_f16(ah, blc, acc[ob]); acc[ob] = kpn_mfma_f16(al, bh[c], acc[ob]); acc[ob] = kpn_mfma_f16(al, blc, acc[ob]); }
            else { acc[ob] = kpn_mfma_f16(al, bh[c], acc[ob]); acc[ob] = kpn_mfma_f16(ah, blc, acc[ob]); }
        }
    });
}
template <int KS, int NOB, int LID = -1, int NSRC>
__device__ __forceinline__ void kpn_hlayer_regs(const float* __restrict__ wseg, int lane, const float (&src)[NSRC], kpn_f32x16 (&acc)[NOB]) {
    static_assert(NSRC >= KS, "operand array too short");
    kpn_hlayer<KS, NOB, LID>(wseg, lane, [&](auto gi, float (&x)[4]) {
        constexpr int g = decltype(gi)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = (g * 4 + i < KS) ? src[g * 4 + i] : 0.0f;
    }, acc);
}

// A Linear layer of the backward chains on v_mfma_f32_32x32x16_bf16 with three bf16 pieces per operand and six products per term
// set (kpn_common.h BH_*): the stream lives in global memory (L2), KS fp32 K-steps taken CW = 7 or 8 at a time.  in_fn has
// kpn_mfma_layer's signature for the segment's G (7: one group per chunk, 4: two).  The B pieces of chunk c + 1 are produced
// while the MFMAs of chunk c issue; the A pieces are fetched in two halves of the output blocks, each half re-loaded right after
// its last MFMA has been issued (an MFMA captures its operands at issue), so that its loads fly under the other half's MFMAs.
template <int BH, class InFn>
__device__ __forceinline__ void kpn_blayer(const float* __restrict__ wp, int lane, InFn&& in_fn,
                                           kpn_f32x16 (&acc)[kpn_bh_shape(BH).nob]) {
    constexpr int KS = kpn_bh_shape(BH).ks, NOB = kpn_bh_shape(BH).nob, G = kpn_bh_shape(BH).g;
    constexpr int CW = kpn_bh_cw(BH), NC = kpn_bh_chunks(BH), NG = KS / G;
    constexpr int H0 = (NOB + 1) / 2, H1 = NOB - H0;
    const float* wseg = wp + kpn_bh_off(BH);
    kpn_bf16x8 wa[3][H0], wb[3][H1 > 0 ? H1 : 1];
    kpn_bf16x8 bp[2][3];                                 // [buffer][piece] of the B operand
    // a RUNNING base pointer (advanced from half to half, re-defined through an empty asm): computed as `segment + constant`
    // the bases are loop-invariant, and hipcc hoists them all out of the tile loop into SGPRs it then spills (242 of them here)
    int prev_pos = 0;
    const float* gp = wseg;
    auto load_half = [&](int c, int ob0, int n, auto& w) {
        const int pos = (c * NOB + ob0) * (3 * 64 * 4);
        gp += pos - prev_pos;
        prev_pos = pos;
        KPN_PIN_POINTER(gp);
        const kpn_gptr4 src = KPN_GLOBAL4(gp) + lane;
#pragma unroll
        for (int k = 0; k < n; ++k)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) w[pc][k] = kpn_as_bf16x8(src[(k * 3 + pc) * 64]);
    };
    auto produce = [&](auto ci, int buf) {
        constexpr int c = decltype(ci)::value;
        float x[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (G == 7) {
            float x7[7];
            in_fn(kpn_ic<c>{}, x7);
#pragma unroll
            for (int i = 0; i < 7; ++i) x[i] = x7[i];
        } else {
            float lo[4], hi[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            in_fn(kpn_ic<2 * c>{}, lo);
            if constexpr (2 * c + 1 < NG) in_fn(kpn_ic<2 * c + 1>{}, hi);
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[i] = lo[i]; x[4 + i] = hi[i]; }
        }
        kpn_split_bf16x8(x, bp[buf][0], bp[buf][1], bp[buf][2]);
    };
    auto mfma_half = [&](int ob0, int n, auto& w, const kpn_bf16x8 (&b)[3]) {
#pragma unroll
        for (int k = 0; k < n; ++k) {
            acc[ob0 + k] = KPN_MFMA16(w[0][k], b[0], acc[ob0 + k]);
            acc[ob0 + k] = KPN_MFMA16(w[0][k], b[1], acc[ob0 + k]);
            acc[ob0 + k] = KPN_MFMA16(w[1][k], b[0], acc[ob0 + k]);
            acc[ob0 + k] = KPN_MFMA16(w[1][k], b[1], acc[ob0 + k]);
            acc[ob0 + k] = KPN_MFMA16(w[0][k], b[2], acc[ob0 + k]);
            acc[ob0 + k] = KPN_MFMA16(w[2][k], b[0], acc[ob0 + k]);
        }
    };
    load_half(0, 0, H0, wa);
    if constexpr (H1 > 0) load_half(0, H0, H1, wb);
    produce(kpn_ic<0>{}, 0);
    kpn_static_for<0, NC>([&](auto ci) {
        constexpr int c = decltype(ci)::value;
        constexpr int cur = c & 1, nxt = cur ^ 1;
        if constexpr (c + 1 < NC) produce(kpn_ic<c + 1>{}, nxt);
        mfma_half(0, H0, wa, bp[cur]);
#pragma unroll
        for (int k = 0; k < H0; ++k) KPN_FENCE_RW(acc[k]);
        if constexpr (c + 1 < NC) load_half(c + 1, 0, H0, wa);
        if constexpr (H1 > 0) {
            mfma_half(H0, H1, wb, bp[cur]);
#pragma unroll
            for (int k = 0; k < H1; ++k) KPN_FENCE_RW(acc[H0 + k]);
            if constexpr (c + 1 < NC) load_half(c + 1, H0, H1, wb);
        }
        KPN_SCHED_BARRIER();
    });
}

// ---------------------------------------------------------------------------------------------
// Scatter of a tile's feature-map gradients through the four bilinear taps (reverse of feat_sample, reference src/utils.py:74-89)
// with RUN-LENGTH COMBINING.  The points of a tile are consecutive samples of a few rays, so neighbouring points project into the
// same texels — and in training all 1024 rays of the 32 x 32 patch cover a handful of texels of a 1/8-resolution map: issued
// one atomic per (point, tap), the float atomics serialised on those cache lines and were 43 % of k_geo_rows_bwd and 45 % of
// k_color_bwd (ablation on the MI355X: 1.53 -> 0.88 ms and 0.95 -> 0.52 ms per call without them).  Here a lane walks the
// points in order and adds up the contributions of a run of equal texel offsets in a register; one atomic per run and tap.
//   sg: staged values [point][ld], tap_o / tap_w: per point the four texel offsets / weights (LDS), npt points.
// C = 64 channels: lane = channel, the four taps in four accumulators.
__device__ __forceinline__ void kpn_scatter_rle64(float* __restrict__ gmap, const float* __restrict__ sg, int ld, const int4* __restrict__ tap_o,
                                                  const float4* __restrict__ tap_w, int npt, int lane) {
    if (npt <= 0) return;
    float* g = gmap + lane;
    int4 cur = tap_o[0];
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    for (int pt = 0; pt < npt; ++pt) {
        const float val = sg[pt * ld + lane];
        const int4 o = tap_o[pt];
        const float4 w = tap_w[pt];
        if (o.x != cur.x) { kpn_atomic_add(g + (size_t)cur.x * 64, a0); a0 = 0.0f; cur.x = o.x; }
        if (o.y != cur.y) { kpn_atomic_add(g + (size_t)cur.y * 64, a1); a1 = 0.0f; cur.y = o.y; }
        if (o.z != cur.z) { kpn_atomic_add(g + (size_t)cur.z * 64, a2); a2 = 0.0f; cur.z = o.z; }
        if (o.w != cur.w) { kpn_atomic_add(g + (size_t)cur.w * 64, a3); a3 = 0.0f; cur.w = o.w; }
        a0 = fmaf(val, w.x, a0); a1 = fmaf(val, w.y, a1); a2 = fmaf(val, w.z, a2); a3 = fmaf(val, w.w, a3);
    }
    kpn_atomic_add(g + (size_t)cur.x * 64, a0); kpn_atomic_add(g + (size_t)cur.y * 64, a1);
    kpn_atomic_add(g + (size_t)cur.z * 64, a2); kpn_atomic_add(g + (size_t)cur.w * 64, a3);
}
// The same walk over all KPN_TILE points with the taps in REGISTERS (lane p holds point p's four offsets and weights; a pad
// lane repeats the last point with zero weights): v_readlane_b32 puts a point's offsets and weights into SGPRs, the run tests
// become scalar compares and branches, and the only LDS access of a step is the staged value.  The LDS-table form above cost
// 43 k cycles per (tile, view) of k_geo_rows_bwd with the atomics compiled out (three dependent LDS reads and four exec-mask
// branches per point).
#ifndef KPN_SIMT_EMU
__device__ __forceinline__ int kpn_readlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float kpn_readlane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
#else
static inline int kpn_readlane(int v, int l) { return __shfl(v, l); }
static inline float kpn_readlane(float v, int l) { return __shfl(v, l); }
#endif
__device__ __forceinline__ void kpn_scatter_rle64_regs(float* __restrict__ gmap, const float* __restrict__ sg, int ld, const int (&o)[4],
                                                       const float (&w)[4], int lane) {
    float* g = gmap + lane;
    int cur[4];
    float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) cur[k] = kpn_readlane(o[k], 0);
    kpn_static_for<0, KPN_TILE>([&](auto pi) {
        constexpr int pt = decltype(pi)::value;
        const float val = sg[pt * ld + lane];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ok = kpn_readlane(o[k], pt);
            if (ok != cur[k]) { kpn_atomic_add(g + (size_t)cur[k] * 64, a[k]); a[k] = 0.0f; cur[k] = ok; }
            a[k] = fmaf(val, kpn_readlane(w[k], pt), a[k]);
        }
    });
#pragma unroll
    for (int k = 0; k < 4; ++k) kpn_atomic_add(g + (size_t)cur[k] * 64, a[k]);
}
// C = 8 channels: lane = (half of the points, tap, channel): lanes 0..31 walk points [0, 16), lanes 32..63 points [16, 32)
__device__ __forceinline__ void kpn_scatter_rle8(float* __restrict__ gmap, const float* __restrict__ sg, int ld, const int4* __restrict__ tap_o,
                                                 const float4* __restrict__ tap_w, int npt, int lane) {
    const int c = lane & 7, tau = (lane >> 3) & 3, p0 = 16 * (lane >> 5);
    const int p1 = npt < p0 + 16 ? npt : p0 + 16;
    if (p0 >= p1) return;
    auto off = [&](int pt) { const int4 o = tap_o[pt]; return tau == 0 ? o.x : (tau == 1 ? o.y : (tau == 2 ? o.z : o.w)); };
    auto wgt = [&](int pt) { const float4 w = tap_w[pt]; return tau == 0 ? w.x : (tau == 1 ? w.y : (tau == 2 ? w.z : w.w)); };
    int cur = off(p0);
    float a = 0.0f;
    for (int pt = p0; pt < p1; ++pt) {
        const int o = off(pt);
        if (o != cur) { kpn_atomic_add(gmap + (size_t)cur * 8 + c, a); a = 0.0f; cur = o; }
        a = fmaf(sg[pt * ld + c], wgt(pt), a);
    }
    kpn_atomic_add(gmap + (size_t)cur * 8 + c, a);
}

// A single-output Linear over a lane's 16 chained features: both halves of a point add their partial
// dot products (lanes p and p+32) and every lane gets  W[row,:].x + b.
__device__ __forceinline__ float kpn_row_dot(const float* __restrict__ rowvec, int h, const float (&x)[16]) {
    const float4* w4 = reinterpret_cast<const float4*>(rowvec + h * 16);
    float acc = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 w = w4[q];
        acc = fmaf(w.x, x[4 * q + 0], acc); acc = fmaf(w.y, x[4 * q + 1], acc);
        acc = fmaf(w.z, x[4 * q + 2], acc); acc = fmaf(w.w, x[4 * q + 3], acc);
    }
    return acc + __shfl_xor(acc, 32) + rowvec[32];
}
// the same for a layer whose input has 8 rows (registers 0..3 of either half; the row vector's other entries are zero padding)
__device__ __forceinline__ float kpn_row_dot4(const float* __restrict__ rowvec, int h, const float (&x)[4]) {
    const float4 w = *reinterpret_cast<const float4*>(rowvec + h * 16);
    const float acc = fmaf(w.w, x[3], fmaf(w.z, x[2], fmaf(w.y, x[1], w.x * x[0])));
    return acc + __shfl_xor(acc, 32) + rowvec[32];
}
