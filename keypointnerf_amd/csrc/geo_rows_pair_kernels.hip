// k_geo_rows_h2: the rows of layers1 on v_mfma_f32_32x32x16_bf16 with split-bf16 operands, TWO 32-point tiles per wavefront
// and ONE wavefront per SIMD (512 registers).  Included by kpn_api.hip after field_kernels.hip; same rows, same row scratch,
// same arithmetic as k_geo_rows_h (every accumulator receives the same six products per step in the same order).
//
// Why this shape (DESIGN.md sections 4.3 and 9.2):
//  * the split-bf16 chain needs 888 MFMAs of 32 cycles per (tile, view) against 1,120 of 64 cycles on the fp32 pipe: the
//    matrix time drops 2.5x, and the VALU work that produces the B operands (activation + three-way split, ~13 instructions
//    per value) becomes comparable to it.  With one 32-point tile per wave the two only overlap across the two waves of a
//    SIMD; with two independent tiles in ONE wave they overlap inside one instruction stream (about five single-issue
//    instructions hide under one of these MFMAs, MI355X_MICROARCH.md);
//  * both tiles multiply by the same weights: every A operand fetched from L2 serves 64 points instead of 32 (the weight
//    stream of this layer set is 444 KB per work item — half the L2 traffic per row);
//  * one wave per SIMD is also the configuration in which the split-bf16 chain has never produced a wrong value (section 9.2:
//    the unexplained failures need two waves sharing a SIMD).
//
// The instruction stream is pipelined BY HAND (the compiler's scheduler, left alone or steered with sched_group_barrier,
// clumps a step's VALU work and then issues its 48 MFMAs back to back — measured 9.7 ms per launch, no faster than the fp32
// kernel): every MFMA of step s is followed by one slice of the work that produces step s+1's B operands, and a scheduling
// barrier after each (MFMA, slice) keeps that order.  A step has 12·NOB MFMAs and 8 operand pairs (2 tiles x 4 pairs of
// values: v_cvt_pk_bf16_f32 converts two values at once), i.e. 6 (NOB = 4) or 3 (NOB = 2) MFMAs per pair:
//     NOB = 4:  value 2j | value 2j+1 | hi piece + residual | mid piece + residual | lo piece | hook (loads for step s+2)
//     NOB = 2:  both values | hi + mid | lo + hook
//
// A work item is (tile pair, view): tiles 2j and 2j+1 of the batch.  Lane l holds point p = l & 31 of BOTH tiles, half
// h = l >> 5 of the K slots / output rows, exactly like k_geo_rows_h.

#ifndef KPN_H2_REGIONS
#define KPN_H2_REGIONS 8
#endif
#ifndef KPN_H2_GV
#define KPN_H2_GV 5
#endif
#ifndef KPN_H2_L0_STEPS
#define KPN_H2_L0_STEPS 16   // timing experiments shorten the chain (wrong results)
#endif
#ifndef KPN_H2_L1_STEPS
#define KPN_H2_L1_STEPS 8
#endif
// Softplus(beta=100, threshold=20) as kpn_softplus100, with the threshold taken on x itself (x > 0.2 instead of 100 x > 20:
// the two differ for the one or two floats next to 0.2, where both branches agree to 2e-11) - one VALU instruction less
__device__ __forceinline__ float kpn_h2_softplus100(float x) {
#if defined(KPN_ABLATE_ACT)
    return x;
#elif defined(KPN_H2_OLD_SOFTPLUS)
    return kpn_softplus100(x);
#else
    const float sp = kpn_log2(1.0f + kpn_exp2(x * 144.269504088896341f)) * 6.93147180559945309e-3f;
    return x > 0.2f ? x : sp;
#endif
}
constexpr int kpn_h2_pa(int pr) { return pr == 2 || pr == 3 ? 1 : (pr == 5 ? 2 : 0); }   // A piece of product pr: h h m m h l
constexpr int kpn_h2_pb(int pr) { return pr == 1 || pr == 3 ? 1 : (pr == 4 ? 2 : 0); }   // B piece:                h m h m l h

// val_fn(kpn_ic<step>, kpn_ic<tile>, kpn_ic<e>) -> the value this lane supplies at K slot e of the step for the tile;
// hook_fn(kpn_ic<step>, kpn_ic<tile>): called once per (step, tile) two steps ahead of the step it names, after the last
// value of step - 1 has been produced (memory loads whose data val_fn(step, ...) consumes are issued here).
template <int KS16, int NOB, class ValFn, class HookFn>
__device__ __forceinline__ void kpn_mfma16_layer2(const float* __restrict__ hseg, int lane, ValFn&& val_fn, HookFn&& hook_fn,
                                                  kpn_f32x16 (&acc)[2][NOB]) {
    static_assert(NOB == 4 || NOB == 2, "6 or 3 MFMAs per operand pair");
    constexpr int H0 = NOB / 2, H1 = NOB - H0;
    constexpr int MF = 12 * NOB, PP = MF / 8;
    kpn_bf16x8 xp[2][2][3];                              // [buffer][tile][piece]
    kpn_bf16x8 wa[3][H0], wb[3][H1];                     // the A pieces of the two halves of the output blocks
    auto load_half = [&](int s, int ob0, int n, auto& w) {
        // a half's pieces are contiguous ([step][block][piece][lane]): one scalar base in its middle, immediate offsets
        // of -3..+2 KB (the 13-bit signed range of global_load), the lane offset in one register for the whole kernel
#ifdef KPN_DBG_H2_SAMEW   // timing experiment (wrong results): every step reads the weights of step 0 -> the stream stays in L1
        const float* gp = hseg + (size_t)(s * 0) * (3 * NOB * 64 * 4) + (size_t)(ob0 * 3 + 3) * (64 * 4);
#else
        const float* gp = hseg + (size_t)s * (3 * NOB * 64 * 4) + (size_t)(ob0 * 3 + 3) * (64 * 4);
#endif
        KPN_PIN_POINTER(gp);
        const kpn_gptr4 src = KPN_GLOBAL4(gp) + lane;
#pragma unroll
        for (int k = 0; k < n; ++k)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) w[pc][k] = kpn_as_bf16x8(src[(k * 3 + pc - 3) * 64]);
    };
    // MFMA number m of a step: half, then product (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi — the order k_geo_rows_h
    // uses per accumulator), then block, then tile: the accumulators of a half rotate, consecutive MFMAs are independent
    auto mfma = [&](auto mi, const kpn_bf16x8 (&x)[2][3]) {
        constexpr int m = decltype(mi)::value;
        if constexpr (m < 12 * H0) {
            constexpr int pr = m / (2 * H0), k = (m / 2) % H0, t = m % 2;
            acc[t][k] = KPN_MFMA16(wa[kpn_h2_pa(pr)][k], x[t][kpn_h2_pb(pr)], acc[t][k]);
        } else {
            constexpr int mm = m - 12 * H0, pr = mm / (2 * H1), k = (mm / 2) % H1, t = mm % 2;
            acc[t][H0 + k] = KPN_MFMA16(wb[kpn_h2_pa(pr)][k], x[t][kpn_h2_pb(pr)], acc[t][H0 + k]);
        }
    };
    // operand pair j (values 2j, 2j+1) of tile t of step sn -> buffer b: value, three-way split (residuals exact in fp32)
    auto produce = [&](auto sn, int b, auto ti, auto ji) {
        constexpr int t = decltype(ti)::value, j = decltype(ji)::value;
        const float x0 = val_fn(sn, ti, kpn_ic<2 * j>{}), x1 = val_fn(sn, ti, kpn_ic<2 * j + 1>{});
        const kpn_bf16_t h0 = kpn_to_bf(x0), h1 = kpn_to_bf(x1);
        const float p0 = x0 - kpn_bf_to_f(h0), p1 = x1 - kpn_bf_to_f(h1);
#ifdef KPN_DBG_H2_NOSPLIT   // timing experiment (wrong results): the three pieces are the same register
        xp[b][t][0][2 * j] = h0; xp[b][t][0][2 * j + 1] = h1;
        xp[b][t][1][2 * j] = h0; xp[b][t][1][2 * j + 1] = h1;
        xp[b][t][2][2 * j] = h0; xp[b][t][2][2 * j + 1] = h1;
        (void)p0; (void)p1;
#else
        const kpn_bf16_t m0 = kpn_to_bf(p0), m1 = kpn_to_bf(p1);
        xp[b][t][0][2 * j] = h0; xp[b][t][0][2 * j + 1] = h1;
        xp[b][t][1][2 * j] = m0; xp[b][t][1][2 * j + 1] = m1;
        xp[b][t][2][2 * j] = kpn_to_bf(p0 - kpn_bf_to_f(m0)); xp[b][t][2][2 * j + 1] = kpn_to_bf(p1 - kpn_bf_to_f(m1));
#endif
    };
    // ---- prologue: the weights of step 0 fly while its operands are produced (nothing to hide them under) ----
    load_half(0, 0, H0, wa);
    load_half(0, H0, H1, wb);
    hook_fn(kpn_ic<0>{}, kpn_ic<0>{}); hook_fn(kpn_ic<0>{}, kpn_ic<1>{});
    if constexpr (KS16 > 1) { hook_fn(kpn_ic<1>{}, kpn_ic<0>{}); hook_fn(kpn_ic<1>{}, kpn_ic<1>{}); }
    kpn_static_for<0, 4>([&](auto ji) {
        produce(kpn_ic<0>{}, 0, kpn_ic<0>{}, ji);
        produce(kpn_ic<0>{}, 0, kpn_ic<1>{}, ji);
    });
    KPN_SCHED_BARRIER();
    // One scheduling region per quarter step: MF/4 MFMAs and the production of operand pair j of BOTH tiles for the next step —
    // four independent dependency chains (a lone chain of dependent VALU instructions exposes the ALU latency when no other
    // wave shares the SIMD), spread over the MFMAs by the group pattern; nothing crosses a region's end.
    kpn_static_for<0, KS16>([&](auto si) {
        constexpr int s = decltype(si)::value;
        constexpr int cur = s & 1, nxt = cur ^ 1;
#if KPN_H2_REGIONS == 8
        // one region per operand pair: MF/8 MFMAs, each followed by its share of the pair's VALU work (group pattern)
        kpn_static_for<0, 8>([&](auto ri) {
            constexpr int r = decltype(ri)::value, t = r % 2, j = r / 2;
            if constexpr (s + 1 < KS16) produce(kpn_ic<s + 1>{}, nxt, kpn_ic<t>{}, kpn_ic<j>{});
            kpn_static_for<r * (MF / 8), (r + 1) * (MF / 8)>([&](auto mi) {
                constexpr int m = decltype(mi)::value;
                mfma(mi, xp[cur]);
                if constexpr (s + 1 < KS16 && m == 12 * H0 - 1) load_half(s + 1, 0, H0, wa);
                if constexpr (s + 1 < KS16 && m == MF - 1) load_half(s + 1, H0, H1, wb);
            });
            if constexpr (s + 2 < KS16 && r == 7) { hook_fn(kpn_ic<s + 2>{}, kpn_ic<0>{}); hook_fn(kpn_ic<s + 2>{}, kpn_ic<1>{}); }
            if constexpr (s + 1 < KS16) {
#pragma unroll
                for (int i = 0; i < MF / 8; ++i) {
                    KPN_SCHED_GROUP(0x008, 1);
                    KPN_SCHED_GROUP(0x002, NOB == 4 ? KPN_H2_GV : 2 * KPN_H2_GV);
                }
            }
            KPN_SCHED_BARRIER();
        });
#else
        constexpr int RG = KPN_H2_REGIONS;                 // scheduling regions per step
        kpn_static_for<0, RG>([&](auto ri) {
            constexpr int r = decltype(ri)::value;
            constexpr int j0 = r * (4 / RG), j1 = (r + 1) * (4 / RG);
            if constexpr (s + 1 < KS16) {
                kpn_static_for<j0, j1>([&](auto ji) {
                    produce(kpn_ic<s + 1>{}, nxt, kpn_ic<0>{}, ji);
                    produce(kpn_ic<s + 1>{}, nxt, kpn_ic<1>{}, ji);
                });
            }
            kpn_static_for<j0 * (MF / 4), j1 * (MF / 4)>([&](auto mi) {
                constexpr int m = decltype(mi)::value;
                mfma(mi, xp[cur]);
                // each half's weight registers are reloaded right after that half's last MFMA has been issued (an MFMA
                // captures its operands at issue: scripts/mfma16_war_probe.hip)
                if constexpr (s + 1 < KS16 && m == 12 * H0 - 1) load_half(s + 1, 0, H0, wa);
                if constexpr (s + 1 < KS16 && m == MF - 1) load_half(s + 1, H0, H1, wb);
            });
            if constexpr (s + 2 < KS16 && r == RG - 1) { hook_fn(kpn_ic<s + 2>{}, kpn_ic<0>{}); hook_fn(kpn_ic<s + 2>{}, kpn_ic<1>{}); }
            KPN_SCHED_BARRIER();
        });
#endif
    });
}

#ifndef KPN_SIMT_EMU
#define KPN_H2_BOUNDS __launch_bounds__(256, 1)
#else
#define KPN_H2_BOUNDS
#endif
__global__ KPN_H2_BOUNDS void k_geo_rows_h2(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                            const int* __restrict__ list, const int* __restrict__ count_ptr,
                                            int* __restrict__ tickets, float* __restrict__ xscr, kpn_batch batch) {
#if defined(KPN_H2_PAD) && !defined(KPN_SIMT_EMU)   // soak builds: shift every instruction of the kernel by 4 * KPN_H2_PAD bytes
#pragma unroll
    for (int i = 0; i < KPN_H2_PAD; ++i) asm volatile("s_nop 0");
#endif
    const int lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    const int count = *count_ptr;
    const int ntiles = (count + KPN_TILE - 1) / KPN_TILE;
    int t0, t1;
    if (!kpn_batch_range(batch, ntiles, t0, t1)) return;
    const int nbt = t1 - t0;                               // tiles of this batch
    const int nwork = ((nbt + 1) >> 1) * sc.V;
    const float pe_pi = 3.14159274101257324f;
    __shared__ __attribute__((aligned(16))) float bias_s[4][128];
    {
        const int segs[4] = {SEG_G1_0A, SEG_G1_1, SEG_G1_2, SEG_G1_3};
        for (int i = threadIdx.x; i < 4 * 128; i += blockDim.x) {
            const int sg = i >> 7, k = i & 127;
            bias_s[sg][k] = k < kpn_seg_bfloats(segs[sg]) ? wp[kpn_seg_boff(segs[sg]) + k] : 0.0f;
        }
    }
    __syncthreads();
    auto no_hook = [](auto, auto) {};
    for (;;) {
        int wi = 0;
        if (lane == 0) wi = atomicAdd(tickets + 0, 1);
        wi = __shfl(wi, 0);
        if (wi >= nwork) break;
        const int pair = wi / sc.V, v = wi - pair * sc.V;
        const bool has1 = 2 * pair + 1 < nbt;              // an odd batch ends in half a pair: tile 1 is computed, not stored
        const float* tb = sc.table + (size_t)v * KPN_TBL_STRIDE;
        float P[2][3], D[2][3];
        kpn_proj q[2];
        float4* dst[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tr = 2 * pair + t;
            int ci = (t0 + tr) * KPN_TILE + p;
            if (ci >= count) ci = count - 1;
            kpn_get_point(ps, (int64_t)list[ci], P[t], D[t]);
            q[t] = kpn_project(tb, P[t][0], P[t][1], P[t][2], sc);
            dst[t] = reinterpret_cast<float4*>(xscr) + ((size_t)(tr * sc.V + v) * KPN_ROW_SLABS) * 64 + lane;
        }
        if (!((sc.keep >> v) & 1u)) {                      // a dropped view: zero rows, the record is still needed
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t == 1 && !has1) break;
                float4 rec0, rec1;
                kpn_row_record(sc, tb, v, h, q[t], P[t], D[t], rec0, rec1);
#pragma unroll
                for (int k = 0; k < 8; ++k) dst[t][k * 64] = make_float4(0.f, 0.f, 0.f, 0.f);
                dst[t][8 * 64] = rec0;
                dst[t][9 * 64] = rec1;
            }
            continue;
        }
        // ---- layers1.0 as ONE 16-step chain (HSEG_G1_0A and HSEG_G1_0B are adjacent, same step size): steps 0-11 one
        //      keypoint each (7 encoding values + a zero slot), steps 12-15 eight geo0 channels each ----
        kpn_f32x16 a0[2][4];
        {
            static_assert(kpn_hseg_off(HSEG_G1_0B) == kpn_hseg_off(HSEG_G1_0A) + 12 * kpn_hseg_step_floats(HSEG_G1_0A), "adjacent segments");
            const float* E = tb + KPN_TBL_EXT;
            float cx[2], cy[2], cz[2];
            kpn_taps tp[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                cx[t] = RADD(kpn_dot3(P[t][0], P[t][1], P[t][2], E[0], E[1], E[2]), E[3]);
                cy[t] = RADD(kpn_dot3(P[t][0], P[t][1], P[t][2], E[4], E[5], E[6]), E[7]);
                cz[t] = RADD(kpn_dot3(P[t][0], P[t][1], P[t][2], E[8], E[9], E[10]), E[11]);
                kpn_load_bias<4>(bias_s[0], h, a0[t]);
                tp[t] = kpn_make_taps(q[t].xn, q[t].yn, sc.g0h, sc.g0w);
            }
            const float* kc = tb + KPN_TBL_KCAM + (12 * h) * 3;
            const float* g0 = sc.geo0 + (size_t)v * sc.g0h * sc.g0w * 64 + 32 * h;
            float pw[2], ps1[2], pc1[2], ps2[2], pc2[2];   // the keypoint in flight: weight, sin/cos of pi z and of 2 pi z
            float4 raw[2][8], gf[2];                       // geo0: the four taps of two float4 of channels; the blended float4
            kpn_mfma16_layer2<KPN_H2_L0_STEPS, 4>(wp + kpn_hseg_off(HSEG_G1_0A), lane, [&](auto si, auto ti, auto ei) -> float {
                constexpr int s = decltype(si)::value, t = decltype(ti)::value, e = decltype(ei)::value;
                if constexpr (s < 12) {
                    if constexpr (e == 0) {
                        const float dx = RSUB(cx[t], kc[s * 3 + 0]), dy = RSUB(cy[t], kc[s * 3 + 1]), dz = RSUB(cz[t], kc[s * 3 + 2]);
                        const float d2 = RADD(RADD(RMUL(dx, dx), RMUL(dy, dy)), RMUL(dz, dz));
                        pw[t] = kpn_fast_exp(-d2 / sc.two_sigma2);
                        kpn_sincos(RMUL(dz, pe_pi), ps1[t], pc1[t]);
                        return dz * pw[t];
                    } else if constexpr (e == 1) { return ps1[t] * pw[t];
                    } else if constexpr (e == 2) { return pc1[t] * pw[t];
                    } else if constexpr (e == 3) {
                        ps2[t] = 2.0f * ps1[t] * pc1[t]; pc2[t] = 1.0f - 2.0f * ps1[t] * ps1[t];
                        return ps2[t] * pw[t];
                    } else if constexpr (e == 4) { return pc2[t] * pw[t];
                    } else if constexpr (e == 5) { return (2.0f * ps2[t] * pc2[t]) * pw[t];
                    } else if constexpr (e == 6) { return (1.0f - 2.0f * ps2[t] * ps2[t]) * pw[t];
                    } else { return 0.0f; }
                } else {
                    if constexpr (e % 4 == 0) {               // blend one float4 of channels: same tap order as ATen (nw, ne, sw, se)
                        const float4 a = raw[t][e], b = raw[t][e + 1], c = raw[t][e + 2], d = raw[t][e + 3];
                        const kpn_taps& w = tp[t];
                        gf[t].x = RADD(RADD(RADD(RMUL(a.x, w.w00), RMUL(b.x, w.w01)), RMUL(c.x, w.w10)), RMUL(d.x, w.w11));
                        gf[t].y = RADD(RADD(RADD(RMUL(a.y, w.w00), RMUL(b.y, w.w01)), RMUL(c.y, w.w10)), RMUL(d.y, w.w11));
                        gf[t].z = RADD(RADD(RADD(RMUL(a.z, w.w00), RMUL(b.z, w.w01)), RMUL(c.z, w.w10)), RMUL(d.z, w.w11));
                        gf[t].w = RADD(RADD(RADD(RMUL(a.w, w.w00), RMUL(b.w, w.w01)), RMUL(c.w, w.w10)), RMUL(d.w, w.w11));
                        return gf[t].x;
                    } else if constexpr (e % 4 == 1) { return gf[t].y;
                    } else if constexpr (e % 4 == 2) { return gf[t].z;
                    } else { return gf[t].w; }
                }
            }, [&](auto si, auto ti) {                        // the taps of geo step s - 12: channels 32h + 8(s-12) .. +7
                constexpr int s = decltype(si)::value, t = decltype(ti)::value;
                if constexpr (s >= 12) {
                    const float* gp = g0 + 8 * (s - 12);
                    const kpn_taps& w = tp[t];
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        raw[t][4 * f + 0] = *reinterpret_cast<const float4*>(gp + (size_t)w.o00 * 64 + 4 * f);
                        raw[t][4 * f + 1] = *reinterpret_cast<const float4*>(gp + (size_t)w.o01 * 64 + 4 * f);
                        raw[t][4 * f + 2] = *reinterpret_cast<const float4*>(gp + (size_t)w.o10 * 64 + 4 * f);
                        raw[t][4 * f + 3] = *reinterpret_cast<const float4*>(gp + (size_t)w.o11 * 64 + 4 * f);
                    }
                }
            }, a0);
        }
        // chained step s of a 128-vector: registers 8(s%2)..+7 of block s/2
        kpn_f32x16 a1[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) kpn_load_bias<4>(bias_s[1], h, a1[t]);
        kpn_mfma16_layer2<KPN_H2_L1_STEPS, 4>(wp + kpn_hseg_off(HSEG_G1_1), lane, [&](auto si, auto ti, auto ei) -> float {
            constexpr int s = decltype(si)::value, t = decltype(ti)::value, e = decltype(ei)::value;
            return kpn_h2_softplus100(a0[t][s / 2][(s % 2) * 8 + e]);
        }, no_hook, a1);
        kpn_f32x16 a2[2][4];
        {
            float4 f[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const kpn_taps tp = kpn_make_taps(q[t].xn, q[t].yn, sc.g1h, sc.g1w);
                f[t] = kpn_tap4(sc.geo1 + (size_t)v * sc.g1h * sc.g1w * 8, 8, 4 * h, tp);
                kpn_load_bias<4>(bias_s[2], h, a2[t]);
            }
            kpn_mfma16_layer2<9, 4>(wp + kpn_hseg_off(HSEG_G1_2), lane, [&](auto si, auto ti, auto ei) -> float {
                constexpr int s = decltype(si)::value, t = decltype(ti)::value, e = decltype(ei)::value;
                if constexpr (s < 8) return kpn_h2_softplus100(a1[t][s / 2][(s % 2) * 8 + e]);
                else if constexpr (e == 0) return f[t].x;
                else if constexpr (e == 1) return f[t].y;
                else if constexpr (e == 2) return f[t].z;
                else if constexpr (e == 3) return f[t].w;
                else return 0.0f;
            }, no_hook, a2);
        }
        {
            kpn_f32x16 acc[2][2];
            float4 rec0[2], rec1[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                rec0[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                rec1[t] = rec0[t];
                kpn_load_bias<2>(bias_s[3], h, acc[t]);
            }
            kpn_mfma16_layer2<8, 2>(wp + kpn_hseg_off(HSEG_G1_3), lane, [&](auto si, auto ti, auto ei) -> float {
                constexpr int s = decltype(si)::value, t = decltype(ti)::value, e = decltype(ei)::value;
                return kpn_h2_softplus100(a2[t][s / 2][(s % 2) * 8 + e]);
            }, [&](auto si, auto ti) {                        // the colour head's gather record, under this layer's MFMAs
                constexpr int s = decltype(si)::value, t = decltype(ti)::value;
                if constexpr (s == 2) kpn_row_record(sc, tb, v, h, q[t], P[t], D[t], rec0[t], rec1[t]);
            }, acc);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t == 1 && !has1) break;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd)
                        dst[t][(b * 4 + qd) * 64] =
                            make_float4(acc[t][b][4 * qd + 0], acc[t][b][4 * qd + 1], acc[t][b][4 * qd + 2], acc[t][b][4 * qd + 3]);
                dst[t][8 * 64] = rec0[t];
                dst[t][9 * 64] = rec1[t];
            }
        }
    }
}
