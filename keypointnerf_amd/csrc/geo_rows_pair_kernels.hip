// k_geo_rows_h2: the rows of layers1 on v_mfma_f32_32x32x16_bf16 with split-bf16 operands, TWO 32-point tiles per wavefront
// and ONE wavefront per SIMD (512 registers).  Its own translation unit on the device (geo_rows_pair_tu.hip); same rows, same row scratch,
// same arithmetic as k_geo_rows_h (every accumulator receives the same six products per step in the same order).
//
// Why this shape (DESIGN.md sections 4.6 and 9.2):
//  * the split-bf16 chain needs 888 MFMAs of 32 cycles per (tile, view) against 1,120 of 64 cycles on the fp32 pipe: the
//    matrix time drops 2.5x, and the VALU work that produces the B operands (activation + three-way split, ~13 instructions
//    per value) becomes comparable to it.  With one 32-point tile per wave the two only overlap across the two waves of a
//    SIMD; with two independent tiles in ONE wave they overlap inside one instruction stream (about five single-issue
//    instructions hide under one of these MFMAs, MI355X_MICROARCH.md);
//  * both tiles multiply by the same weights: every A operand fetched from L2 serves 64 points instead of 32 (the weight
//    stream of this layer set is 444 KB per work item — half the L2 traffic per row);
//  * (rounds 2-3 also noted that the one-tile kernel at two waves per SIMD produced rare wrong half-tiles and this shape never
//    did.  Round 5 found the cause, and it is not the occupancy: an inline-asm operand split whose output registers the allocator
//    had placed inside the destination tuple of an MFMA still in flight loses against the MFMA's write-back — DESIGN.md 4.5, 9.2,
//    scripts/repro_asm_waw_hazard.hip.  The asm slices of THIS kernel are hand-placed and audited on every build:
//    tests/test_isa_audit.py.)
//
// The instruction stream is pipelined BY HAND (the compiler's scheduler, left alone or steered with sched_group_barrier,
// clumps a step's VALU work and then issues its 48 MFMAs back to back — measured 9.7 ms per launch, no faster than the fp32
// kernel): every MFMA of step s is followed by one slice of the work that produces step s+1's B operands, and a scheduling
// barrier after each (MFMA, slice) keeps that order.  A step has 12·NOB MFMAs and 8 operand pairs (2 tiles x 4 pairs of
// values: v_cvt_pk_bf16_f32 converts two values at once), i.e. 6 (NOB = 4) or 3 (NOB = 2) MFMAs per pair:
//     NOB = 4:  six slices of 3-5 instructions, one per MFMA (kpn_h2_slice)
//     NOB = 2:  two slices per MFMA
//
// A work item is (tile pair, view): tiles 2j and 2j+1 of the batch.  Lane l holds point p = l & 31 of BOTH tiles, half
// h = l >> 5 of the K slots / output rows, exactly like k_geo_rows_h.

#ifndef KPN_H2_LOOKAHEAD
#define KPN_H2_LOOKAHEAD true   // the next layer's step-0 operands are produced under the last step
#endif
// Empty volatile asm statements keep their program order and anchor the (side-effect free) MFMAs and VALU slices between the
// scheduling barriers: without them instruction selection is free to sink a whole layer's MFMAs below all of its slices —
// it did, for layers1.1 — and the barriers then fence nothing.  Accumulators live in AGPRs ("a"), the rest in VGPRs.
#ifdef KPN_SIMT_EMU
#define KPN_H2_PIN_ACC(v) ((void)0)
#else
#define KPN_H2_PIN_ACC(v) asm volatile("" : "+a"(v))
#endif
// ---- two operand schemes (SC) ----
//   kpn_sc_bf16x3 (rows mode 2): x = h + m + l in bf16 (8 + 8 + 8 significant bits), six products per term set
//                 hh hm mh mm hl lh (everything above 2^-24 relative), v_mfma_f32_32x32x16_bf16.  fp32's exponent range.
//   kpn_sc_f16x2  (rows mode 3, the default): x = h + l in fp16 (11 + 11 bits; the residual x - h is formed exactly by ONE
//                 v_fma_mix_f32 per value, which reads the fp16 half in place), three products hh hl lh (four with ll until round 4's last build),
//                 v_mfma_f32_32x32x16_f16: 1.5x fewer MFMAs and 2 instead of 5.5 split instructions per value.  Measured
//                 (scripts/f16_split_probe.hip, MI355X): the residual is always exact, |x - (h + l)| <= 2^-23 |x| with an
//                 absolute floor of 2^-24 (fp16's subnormal quantum; the f16 MFMA honours subnormal inputs), a K = 256 dot
//                 product is as close to fp64 as the bf16x3 form and the fp32 fma chain (1.6e-7 / 1.8e-7 / 2.0e-7 of the
//                 sum of |terms|) — EXCEPT for operands far below 1e-2, where the floor shows: layers1.3's weights, scaled
//                 by ln(2)/100 for the log2-unit activation, are therefore packed times 2^10 and the rows multiplied by 2^-10
//                 when they are stored.  The price is fp16's range: an operand beyond 65504 (a pre-activation beyond 454 in
//                 natural units, a packed weight beyond 65504) becomes inf and the row NaN.  (Round 3 called that "loudly
//                 wrong"; it was not — a clamp or a v_med3 turns the NaN back into a finite, wrong number.  Since round 4 every
//                 activation of these kernels keeps a NaN and the RANGE GUARD, kpn_field_shared.h, has such a batch evaluated
//                 again by the fp32-range kernels: DESIGN.md 4.6.)  kpn_set_geo_rows_mode(2) is the range-safe alternative.
struct kpn_sc_bf16x3 {
    static constexpr int NP = 3, NPROD = 6, NSLICE = 6;
    static constexpr int pa(int pr) { return pr == 2 || pr == 3 ? 1 : (pr == 5 ? 2 : 0); }   // A piece of product pr: h h m m h l
    static constexpr int pb(int pr) { return pr == 1 || pr == 3 ? 1 : (pr == 4 ? 2 : 0); }   // B piece:                h m h m l h
    static constexpr int hseg_base() { return kpn_hseg_off(0); }
    static constexpr float out_up = 1.0f, out_down = 1.0f;
    static constexpr bool PREFETCH = false;   // register-saturated (256 VGPRs): the next item's data would only add spills
    static constexpr int WBUF = 1;
    static __device__ __forceinline__ kpn_f32x16 mfma(kpn_f32x4 a, kpn_u32x4 b, kpn_f32x16 c);
};
struct kpn_sc_f16x2 {
    // KPN_F16_PRODUCTS (kpn_common.h) = 3: the products hh hl lh.  |l| <= 2^-12 |x| for both operands, so the fourth product ll is
    // at most 2^-24 of the term — half an fp32 ulp, what ONE rounding of the reference's fp32 fma chain costs — and the bf16 scheme
    // above drops its terms of that size (ml, lm) as well.  25 % fewer MFMAs; three slices of 5-6 instructions, one per gap.
    static constexpr int NP = 2, NPROD = KPN_F16_PRODUCTS, NSLICE = KPN_F16_PRODUCTS == 3 ? 3 : 4;
    static constexpr int pa(int pr) { return NPROD == 3 ? (pr == 2 ? 1 : 0) : pr >> 1; }       // h h l (l)
    static constexpr int pb(int pr) { return NPROD == 3 ? (pr == 1 ? 1 : 0) : pr & 1; }        // h l h (l)
    static constexpr int hseg_base() { return kpn_fseg_off(0); }
    static constexpr float out_up = KPN_F16_ROW_SCALE, out_down = 1.0f / KPN_F16_ROW_SCALE;    // layers1.3 is packed times 2^10
    static constexpr bool PREFETCH = true;    // the next work item's ticket / list entries / points fetched under this one's layers
#ifndef KPN_H2_WBUF
#define KPN_H2_WBUF 2
#endif
    static constexpr int WBUF = KPN_H2_WBUF;   // weight registers double-buffered (kpn_mfma16_layer2)
    static __device__ __forceinline__ kpn_f32x16 mfma(kpn_f32x4 a, kpn_u32x4 b, kpn_f32x16 c);
};
// ---- the operand production: one volatile asm BLOCK per slice ----
// The VALU work between two MFMAs is budgeted in issue slots (five hide under one v_mfma_f32_32x32x16_bf16 at one wave per
// SIMD, a transcendental counts about two: scripts/mfma16_filler_probe.hip), so what hipcc selects matters as much as where it
// puts it.  Written in C++ with pin statements (round 2) the three-way split came out as one v_cvt_pk_bf16_f32 PER VALUE
// (second source zero) plus v_mov copies: 7.5 instructions per value instead of the 5.5 of
//     pk = cvt_pk(x0, x1);  t0 = pk << 16;  t1 = pk & 0xffff0000;  x0 -= t0;  x1 -= t1        (twice, then one more cvt_pk),
// and every pin ("+v" in an empty asm) made hipcc's hazard recogniser assume a partial-register write and put an s_nop in
// front of the next reader: 1,088 of them per work item.  A slice is therefore ONE asm statement holding its 2-5 instructions:
// they are selected as written, stay in program order (volatile statements are not reordered against each other), need no
// pins, and the recogniser — which neither looks inside an asm statement nor counts it as a wait state — sees each
// statement's results consumed only behind the next MFMA.  The one software hazard such a stream can meet on gfx950, a
// transcendental's result read by a non-transcendental VALU instruction in the very next issue slot (one wait state), is
// excluded by construction: inside a block another instruction always sits between a v_exp_f32 / v_log_f32 and the first
// use of its result, and across blocks the MFMA does.  The accumulators are read by the blocks themselves
// (v_accvgpr_read_b32 from an "a" operand; left to the compiler every value was read twice): an accumulator block is final
// at least three MFMA issues (96 cycles) before its first read (static_assert in kpn_mfma16_layer2).
#ifndef KPN_SIMT_EMU
__device__ __forceinline__ kpn_bf16x8 kpn_as_bf16x8(kpn_u32x4 v) { return __builtin_bit_cast(kpn_bf16x8, v); }
#define KPN_H2_USE(v) asm volatile("" ::"v"(v))      // the value exists before this point; defines nothing (no assumed hazard)
__device__ __forceinline__ kpn_f32x16 kpn_sc_bf16x3::mfma(kpn_f32x4 a, kpn_u32x4 b, kpn_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kpn_bf16x8, a), __builtin_bit_cast(kpn_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ kpn_f32x16 kpn_sc_f16x2::mfma(kpn_f32x4 a, kpn_u32x4 b, kpn_f32x16 c) {
    return kpn_mfma_f16(a, b, c);
}
#else
static inline kpn_bf16x8 kpn_as_bf16x8(kpn_u32x4 v) { kpn_bf16x8 r; memcpy(&r, &v, 16); return r; }
#define KPN_H2_USE(v) ((void)0)
static inline uint32_t kpn_emu_cvt_pk(float lo, float hi) { return (uint32_t)kpn_f2bf(lo) | ((uint32_t)kpn_f2bf(hi) << 16); }
static inline float kpn_emu_bf_lo(uint32_t pk) { return kpn_bf2f((uint16_t)(pk & 0xffffu)); }
static inline float kpn_emu_bf_hi(uint32_t pk) { return kpn_bf2f((uint16_t)(pk >> 16)); }
static inline uint32_t kpn_emu_cvt_pk_f16(float lo, float hi) { return (uint32_t)kpn_f2h(lo) | ((uint32_t)kpn_f2h(hi) << 16); }
static inline float kpn_emu_h_lo(uint32_t pk) { return kpn_h2f((uint16_t)(pk & 0xffffu)); }
static inline float kpn_emu_h_hi(uint32_t pk) { return kpn_h2f((uint16_t)(pk >> 16)); }
inline kpn_f32x16 kpn_sc_bf16x3::mfma(kpn_f32x4 a, kpn_u32x4 b, kpn_f32x16 c) {
    kpn_bf16x8 av, bv; memcpy(&av, &a, 16); memcpy(&bv, &b, 16);
    return simt_mfma_f32_32x32x16_bf16(av, bv, c);
}
inline kpn_f32x16 kpn_sc_f16x2::mfma(kpn_f32x4 a, kpn_u32x4 b, kpn_f32x16 c) { return kpn_mfma_f16(a, b, c); }
#endif

// Softplus(beta 100, threshold 20) in LOG2 UNITS (kpn_common.h KPN_H2_LOG2ACT): the packer folds 100 log2(e) into the weights
// and biases of the layer that PRODUCES a pre-activation and ln(2)/100 into the weights of the layer that CONSUMES the activation
// (kpn_hseg_factor), so that the accumulators hold u = 100 log2(e) x and the next layer wants y' = 100 log2(e) softplus(x) =
// log2(1 + 2^u), computed as
//       max(u, 0) + log2(1 + 2^-|u|)
// — v_exp_f32 with the -|.| source modifiers, +1, v_log_f32, v_max_f32, v_add_f32: as many instructions as the round-3 form
// max(u, log2(1 + 2^min(u, 126))), no clamp (2^-|u| <= 1), and beyond the reference's threshold branch (x if 100 x > 20,
// src/utils.py:523-524) the second term is below an ulp of u.  And it KEEPS a NaN: v_min_f32 / v_max_f32 return the numeric
// operand when the other is a NaN, so the old form turned a NaN pre-activation — which is what an operand beyond fp16's range
// makes of every accumulator it touches (h = inf, l = -inf, inf - inf) — into a finite 126; here 2^-|NaN| is a NaN and the sum
// carries it to the stored row, where the per-point kernel's range guard finds it (kpn_field_shared.h kpn_batch).

// ---- staged production of one operand pair (two fp32 values -> three bf16 pieces each = one dword per piece), in SIX slices.
// With ACT the values go through the activation first; issue slots per slice (transcendental = 2): 6 6 5 4 4 2
//   0: read u0 u1, exp 0, exp 1          1: max 0, 1 + e0, 1 + e1, log 0     2: log 1, max 1, add 0, add 1, hi pieces
//   3: unpack hi 0/1, residual 0/1       4: mid pieces, unpack 0/1, residual 0   5: residual 1, lo pieces
// Without ACT: 0: x0 = v0()   1: x1 = v1()   2: hi pieces   3..5 as above.
// fp16 double split (kpn_sc_f16x2), FOUR slices, issue slots 6 6 5 3 (measured: a pair's 4 MFMAs + these slices run at 32.1
// cycles per MFMA, scripts/f16_split_probe.hip):
//   0, 1: as above     2: log 1, max 1, add 0, add 1, hi pieces (v_cvt_pk_f16_f32)     3: x0 - h0, x1 - h1 (v_fma_mix_f32), lo pieces
// (issue slots 6 5 6 3).  A transcendental's result is never read by the next VALU instruction: another instruction or an MFMA
// always sits between.
struct kpn_h2_pair { float x0, x1, e0, e1; uint32_t pk; };
template <class SC, bool ACT, int Q, int J, class V0, class V1>
__device__ __forceinline__ void kpn_h2_slice(kpn_h2_pair& p, kpn_u32x4 (&dst)[SC::NP], V0&& v0, V1&& v1) {
    constexpr bool F16 = SC::NP == 2;
    if constexpr (F16 && SC::NSLICE == 3) {
        // three products: THREE slices, one per MFMA gap (NOB = 4; two for NOB = 2), issue slots 7 7 6 (a transcendental = 2).
        // One asm statement per gap: hipcc puts an s_nop between two adjacent asm statements.  Concatenated in order the sixteen
        // instructions never read a transcendental's result in the next slot.
        //   0: read u0 u1, exp 0, exp 1, max 0     1: 1 + e0, 1 + e1, log 0, log 1, max 1     2: add 0, add 1, hi pieces, x0 - h0, x1 - h1, lo pieces
        //   without ACT: 0: x0 = v0()   1: x1 = v1()   2: hi pieces, x0 - h0, x1 - h1, lo pieces
        if constexpr (Q == 0) {
            if constexpr (ACT) {
                const float a0 = v0(), a1 = v1();
#ifndef KPN_SIMT_EMU
                asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_exp_f32 %2, -|%0|\n\tv_exp_f32 %3, -|%1|\n\tv_max_f32 %0, 0, %0"
                             : "=&v"(p.x0), "=&v"(p.x1), "=&v"(p.e0), "=&v"(p.e1) : "a"(a0), "a"(a1));
#else
                p.x0 = a0; p.x1 = a1;
                p.e0 = kpn_exp2(-fabsf(p.x0)); p.e1 = kpn_exp2(-fabsf(p.x1));
                p.x0 = fmaxf(p.x0, 0.0f);
#endif
            } else { p.x0 = v0(); KPN_H2_USE(p.x0); }
        } else if constexpr (Q == 1) {
            if constexpr (ACT) {
#ifndef KPN_SIMT_EMU
                asm volatile("v_add_f32 %0, 1.0, %0\n\tv_add_f32 %1, 1.0, %1\n\tv_log_f32 %0, %0\n\tv_log_f32 %1, %1\n\tv_max_f32 %2, 0, %2"
                             : "+v"(p.e0), "+v"(p.e1), "+v"(p.x1));
#else
                p.e0 = 1.0f + p.e0; p.e1 = 1.0f + p.e1; p.e0 = kpn_log2(p.e0); p.e1 = kpn_log2(p.e1); p.x1 = fmaxf(p.x1, 0.0f);
#endif
            } else { p.x1 = v1(); KPN_H2_USE(p.x1); }
        } else {
#ifndef KPN_SIMT_EMU
            uint32_t lo;
            if constexpr (ACT)
                asm volatile("v_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %5\n\tv_cvt_pk_f16_f32 %0, %2, %3\n\t"
                             "v_fma_mix_f32 %2, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                             "v_cvt_pk_f16_f32 %1, %2, %3"
                             : "=&v"(p.pk), "=&v"(lo), "+v"(p.x0), "+v"(p.x1) : "v"(p.e0), "v"(p.e1));
            else
                asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
                             "v_fma_mix_f32 %2, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                             "v_cvt_pk_f16_f32 %1, %2, %3"
                             : "=&v"(p.pk), "=&v"(lo), "+v"(p.x0), "+v"(p.x1));
            dst[0][J] = p.pk; dst[1][J] = lo;
#else
            if constexpr (ACT) { p.x0 = p.x0 + p.e0; p.x1 = p.x1 + p.e1; }
            p.pk = kpn_emu_cvt_pk_f16(p.x0, p.x1);
            dst[0][J] = p.pk;
            p.x0 = p.x0 - kpn_emu_h_lo(p.pk); p.x1 = p.x1 - kpn_emu_h_hi(p.pk);
            dst[1][J] = kpn_emu_cvt_pk_f16(p.x0, p.x1);
#endif
        }
    } else if constexpr (F16 && Q == 2) {
        if constexpr (ACT) {
#ifndef KPN_SIMT_EMU
            asm volatile("v_log_f32 %3, %3\n\tv_max_f32 %1, 0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\tv_cvt_pk_f16_f32 %4, %0, %1"
                         : "+v"(p.x0), "+v"(p.x1), "+v"(p.e0), "+v"(p.e1), "=&v"(p.pk));
#else
            p.e1 = kpn_log2(p.e1);
            p.x1 = fmaxf(p.x1, 0.0f);
            p.x0 = p.x0 + p.e0; p.x1 = p.x1 + p.e1;
            p.pk = kpn_emu_cvt_pk_f16(p.x0, p.x1);
#endif
        } else {
#ifndef KPN_SIMT_EMU
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p.pk) : "v"(p.x0), "v"(p.x1));
#else
            p.pk = kpn_emu_cvt_pk_f16(p.x0, p.x1);
#endif
        }
        dst[0][J] = p.pk;
    } else if constexpr (F16 && Q == 3) {
#ifndef KPN_SIMT_EMU
        uint32_t lo;
        asm volatile("v_fma_mix_f32 %1, %3, -1.0, %1 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %2, %3, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                     "v_cvt_pk_f16_f32 %0, %1, %2" : "=&v"(lo), "+v"(p.x0), "+v"(p.x1) : "v"(p.pk));
        dst[1][J] = lo;
#else
        p.x0 = p.x0 - kpn_emu_h_lo(p.pk); p.x1 = p.x1 - kpn_emu_h_hi(p.pk);
        dst[1][J] = kpn_emu_cvt_pk_f16(p.x0, p.x1);
#endif
    } else if constexpr (Q == 0) {
        if constexpr (ACT) {
            const float a0 = v0(), a1 = v1();             // two accumulator elements (AGPRs)
#ifndef KPN_SIMT_EMU
            asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\t"
                         "v_exp_f32 %2, -|%0|\n\tv_exp_f32 %3, -|%1|"
                         : "=&v"(p.x0), "=&v"(p.x1), "=&v"(p.e0), "=&v"(p.e1) : "a"(a0), "a"(a1));
#else
            p.x0 = a0; p.x1 = a1;
            p.e0 = kpn_exp2(-fabsf(p.x0)); p.e1 = kpn_exp2(-fabsf(p.x1));
#endif
        } else { p.x0 = v0(); KPN_H2_USE(p.x0); }
    } else if constexpr (Q == 1) {
        if constexpr (ACT) {
#ifndef KPN_SIMT_EMU
            asm volatile("v_max_f32 %2, 0, %2\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %1, 1.0, %1\n\tv_log_f32 %0, %0" : "+v"(p.e0), "+v"(p.e1), "+v"(p.x0));
#else
            p.x0 = fmaxf(p.x0, 0.0f); p.e0 = 1.0f + p.e0; p.e1 = 1.0f + p.e1; p.e0 = kpn_log2(p.e0);
#endif
        } else { p.x1 = v1(); KPN_H2_USE(p.x1); }
    } else if constexpr (Q == 2) {
        if constexpr (ACT) {
#ifndef KPN_SIMT_EMU
            asm volatile("v_log_f32 %3, %3\n\tv_max_f32 %1, 0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\tv_cvt_pk_bf16_f32 %4, %0, %1"
                         : "+v"(p.x0), "+v"(p.x1), "+v"(p.e0), "+v"(p.e1), "=&v"(p.pk));
#else
            p.e1 = kpn_log2(p.e1);
            p.x1 = fmaxf(p.x1, 0.0f);
            p.x0 = p.x0 + p.e0; p.x1 = p.x1 + p.e1;
            p.pk = kpn_emu_cvt_pk(p.x0, p.x1);
#endif
        } else {
#ifndef KPN_SIMT_EMU
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p.pk) : "v"(p.x0), "v"(p.x1));
#else
            p.pk = kpn_emu_cvt_pk(p.x0, p.x1);
#endif
        }
        dst[0][J] = p.pk;
    } else if constexpr (Q == 3) {
#ifndef KPN_SIMT_EMU
        asm volatile("v_lshlrev_b32 %2, 16, %4\n\tv_and_b32 %3, 0xffff0000, %4\n\tv_sub_f32 %0, %0, %2\n\tv_sub_f32 %1, %1, %3"
                     : "+v"(p.x0), "+v"(p.x1), "=&v"(p.e0), "=&v"(p.e1) : "v"(p.pk));
#else
        p.x0 = p.x0 - kpn_emu_bf_lo(p.pk); p.x1 = p.x1 - kpn_emu_bf_hi(p.pk);
#endif
    } else if constexpr (Q == 4) {
#ifndef KPN_SIMT_EMU
        asm volatile("v_cvt_pk_bf16_f32 %2, %0, %3\n\tv_lshlrev_b32 %1, 16, %2\n\tv_sub_f32 %0, %0, %1\n\tv_and_b32 %1, 0xffff0000, %2"
                     : "+v"(p.x0), "=&v"(p.e1), "=&v"(p.pk) : "v"(p.x1));
#else
        p.pk = kpn_emu_cvt_pk(p.x0, p.x1);
        p.x0 = p.x0 - kpn_emu_bf_lo(p.pk); p.e1 = kpn_emu_bf_hi(p.pk);
#endif
        dst[1][J] = p.pk;
    } else {
#ifndef KPN_SIMT_EMU
        uint32_t lo;
        asm volatile("v_sub_f32 %1, %1, %3\n\tv_cvt_pk_bf16_f32 %0, %2, %1" : "=&v"(lo), "+v"(p.x1) : "v"(p.x0), "v"(p.e1));
        dst[2][J] = lo;
#else
        p.x1 = p.x1 - p.e1;
        dst[2][J] = kpn_emu_cvt_pk(p.x0, p.x1);
#endif
    }
}

// One Linear layer for two tiles.  The instruction stream is laid out by hand: every MFMA is followed by one slice (NOB = 4;
// two slices for NOB = 2) of the work that produces the NEXT step's B operands and by a scheduling barrier.
//   val_fn(kpn_ic<step>, kpn_ic<tile>, kpn_ic<e>)   the value at K slot e (the pre-activation for steps < ACT)
//   stage_fn(kpn_ic<step>, kpn_ic<tile>, kpn_ic<j>, kpn_ic<q>)   q = 0..2, steps >= ACT only: extra work in the slices of pair j
//            that are nearly empty without an activation (look-ahead computations, loads for later steps)
//   tail_fn(kpn_ic<m>)   after MFMA m of the LAST step, which has no operands of its own layer left to produce
//   HAVE0: the operands of step 0 arrive in x0 (produced under the previous layer's last step) instead of being produced
//          in a prologue that nothing hides;  NEXT: produce the next layer's step-0 operands (next_fn(kpn_ic<tile>, kpn_ic<e>),
//          through the activation) into xn under the second half of the last step — the next layer's step 0 reads output
//          block 0 only, whose accumulators are final once the first half (blocks 0 and 1) has been issued.
struct kpn_h2_identity { static constexpr int at(int p) { return p; } };
//   SMAP::at(position) = step of the weight stream executed at that position of the chain (the order of a layer's K steps is
//   free; val_fn / stage_fn are called with positions)
template <class SC, int KS16, int NOB, int ACT, bool HAVE0, bool NEXT, class SMAP = kpn_h2_identity, class ValFn, class StageFn, class TailFn, class NextFn>
__device__ __forceinline__ void kpn_mfma16_layer2(const float* __restrict__ hseg, int lane, ValFn&& val_fn, StageFn&& stage_fn,
                                                  TailFn&& tail_fn, NextFn&& next_fn, kpn_f32x16 (&acc)[2][NOB],
                                                  kpn_u32x4 (&x0)[2][SC::NP], kpn_u32x4 (&xn)[2][SC::NP]) {
    static_assert(NOB == 4 || NOB == 2, "NPROD * NOB / 4 MFMAs per operand pair");
    static_assert(!NEXT || NOB == 4, "the look-ahead needs blocks 0/1 in the first half of a step");
    constexpr int NP = SC::NP, NPROD = SC::NPROD, NSLICE = SC::NSLICE;
    constexpr int H0 = NOB / 2, H1 = NOB - H0;
    constexpr int MF = 2 * NPROD * NOB;                  // MFMAs per step
    constexpr int SPG = 8 * NSLICE / MF;                 // slices per MFMA gap (8 operand pairs of NSLICE slices per step): 1, 2 or 4
    static_assert(SPG * MF == 8 * NSLICE, "whole slices per gap");
    kpn_u32x4 xp[2][2][NP];                              // [buffer][tile][piece]: four dwords = eight 16-bit values
    // the A pieces of the two halves of the output blocks (raw dwords).  SC::WBUF = 2 (the fp16 scheme, which has the registers):
    // two buffers, step s reads buffer s & 1 and a half's registers are reloaded — right after its last MFMA has been issued —
    // with the weights of step s + 2: a step and a half of MFMAs (36 x 32 cycles) between a load and its first use instead of
    // half a step (12 MFMAs = 384 cycles with three products, less than an L2 round trip under load): rows kernel -2.4 %.
    // (Handing the registers on from layer to layer as well — the first two steps of layers1.1-1.3 fetched under the previous
    // layer's last two — measured no different: 3.240 vs 3.239 ms per launch; not kept.)
    constexpr int WBUF = SC::WBUF;
    auto wsel = [](int s) constexpr { return SC::WBUF == 2 ? (s & 1) : 0; };
    kpn_f32x4 wa[WBUF][NP][H0], wb[WBUF][NP][H1];
    kpn_h2_pair pr{0.f, 0.f, 0.f, 0.f, 0u};
    // A step's pieces are contiguous ([step][block][piece][lane] x 16 B: 8 KB with four output blocks): ONE uniform base for the
    // layer (SGPR pair), a per-lane byte offset in a VGPR that points at the MIDDLE of the current step, and immediate offsets of
    // -4 .. +3 KB (the 13-bit signed range of global_load; NP = 2 only — the three-piece scheme's 12-KB steps need base
    // adjustments) for its eight 1-KB rows: the offset register is advanced once per step
    // by one v_add_u32 (re-defined through an empty asm so that the loads cannot rise above it).  Until round 4 the base was a
    // running SGPR pointer advanced per HALF step: s_add_u32 + s_addc_u32 = 290 scalar instructions per (tile pair, view) in a
    // kernel that is bound by its issue slots at one wave per SIMD (profiles/r05_b_rows_kernel_ablations.txt).  (Computed as
    // `segment + constant` the bases are loop-invariant, and hipcc hoists them out of the work loop into SGPRs it then spills.)
    constexpr int STEP_BYTES = NP * NOB * 64 * 16;
    int prev_step = 0;
    uint32_t voff = (uint32_t)lane * 16u + (uint32_t)(STEP_BYTES / 2);
#ifndef KPN_SIMT_EMU
    asm volatile("" : "+v"(voff));
#endif
    auto load_half = [&](int s, int ob0, int n, auto& w) {
        // (the offset follows the step whichever half asks first — an advisor finding of round 5: with `ob0 == 0 &&` in the condition
        // a future reorder of the load_virtual calls would have read the previous step's weights for a second half)
        if (s != prev_step) {                       // (compile-time: s and prev_step are constants after unrolling)
            voff += (uint32_t)((s - prev_step) * STEP_BYTES);
            prev_step = s;
#ifndef KPN_SIMT_EMU
            asm volatile("" : "+v"(voff));
#endif
        }
        const char* row0 = reinterpret_cast<const char*>(hseg) + voff;
#pragma unroll
        for (int k = 0; k < n; ++k)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc)
                w[pc][k] = *KPN_GLOBAL4(row0 + ((ob0 + k) * NP + pc) * 1024 - STEP_BYTES / 2);
    };
    // half hf (0 / 1) of step vs into the buffer that step reads
    auto load_virtual = [&](auto vsi, auto hfi) {
        constexpr int vs = decltype(vsi)::value, hf = decltype(hfi)::value, b = wsel(vs);
        if constexpr (vs < KS16) {
            if constexpr (hf == 0) load_half(SMAP::at(vs), 0, H0, wa[b]); else load_half(SMAP::at(vs), H0, H1, wb[b]);
        }
    };
    // MFMA number m of a step: half, then product (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi — the order k_geo_rows_h
    // uses per accumulator), then block, then tile: the accumulators of a half rotate, consecutive MFMAs are independent
    auto mfma = [&](auto mi, auto wi, const kpn_u32x4 (&x)[2][NP]) {
        constexpr int m = decltype(mi)::value, wsel = decltype(wi)::value;
        if constexpr (m < 2 * NPROD * H0) {
            constexpr int prd = m / (2 * H0), k = (m / 2) % H0, t = m % 2;
            acc[t][k] = SC::mfma(wa[wsel][SC::pa(prd)][k], x[t][SC::pb(prd)], acc[t][k]);
            KPN_H2_PIN_ACC(acc[t][k]);
        } else {
            constexpr int mm = m - 2 * NPROD * H0, prd = mm / (2 * H1), k = (mm / 2) % H1, t = mm % 2;
            acc[t][H0 + k] = SC::mfma(wb[wsel][SC::pa(prd)][k], x[t][SC::pb(prd)], acc[t][H0 + k]);
            KPN_H2_PIN_ACC(acc[t][H0 + k]);
        }
    };
    // slice q of pair j of tile t of step sn -> buffer b
    auto slice = [&](auto sn, int b, auto ti, auto ji, auto qi) {
        constexpr int t = decltype(ti)::value, j = decltype(ji)::value, q = decltype(qi)::value;
        constexpr bool act = decltype(sn)::value < ACT;
        if constexpr (!act && q < 3) stage_fn(sn, ti, ji, qi);
        kpn_h2_slice<SC, act, q, j>(pr, xp[b][t], [&]() { return val_fn(sn, ti, kpn_ic<2 * j>{}); }, [&]() { return val_fn(sn, ti, kpn_ic<2 * j + 1>{}); });
    };
    load_virtual(kpn_ic<0>{}, kpn_ic<0>{}); load_virtual(kpn_ic<0>{}, kpn_ic<1>{});
    if constexpr (WBUF == 2) { load_virtual(kpn_ic<1>{}, kpn_ic<0>{}); load_virtual(kpn_ic<1>{}, kpn_ic<1>{}); }
    if constexpr (HAVE0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) xp[0][t][pc] = x0[t][pc];
    } else {   // prologue: nothing to hide the production of step 0 under
        kpn_static_for<0, 8>([&](auto pi) {
            constexpr int t = decltype(pi)::value % 2, j = decltype(pi)::value / 2;
            kpn_static_for<0, NSLICE>([&](auto qi) { slice(kpn_ic<0>{}, 0, kpn_ic<t>{}, kpn_ic<j>{}, qi); });
        });
    }
    KPN_SCHED_BARRIER();
    kpn_static_for<0, KS16>([&](auto si) {
        constexpr int s = decltype(si)::value;
        constexpr int cur = s & 1, nxt = cur ^ 1;
#ifdef KPN_PRECISION_PROBE   // probe builds only (kpn_common.h): this step's lo pieces of B / A replaced by zero
        if constexpr (NP == 2) {
            constexpr int lid = KS16 == 16 ? 0 : (KS16 == 9 ? 2 : (NOB == 2 ? 3 : 1));
            constexpr int grp = SMAP::at(s) < 12 ? 4 : 5;          // layers1.0: keypoint-encoding steps / sampled-channel steps
            const bool drop_b = KPN_PROBE(lid, 0) || (lid == 0 && KPN_PROBE(grp, 0));
            const bool drop_a = KPN_PROBE(lid, 1) || (lid == 0 && KPN_PROBE(grp, 1));
            if (drop_b) { xp[cur][0][1] = kpn_u32x4{0u, 0u, 0u, 0u}; xp[cur][1][1] = kpn_u32x4{0u, 0u, 0u, 0u}; }
            if (drop_a) {
#pragma unroll
                for (int k = 0; k < H0; ++k) wa[wsel(s)][1][k] = kpn_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < H1; ++k) wb[wsel(s)][1][k] = kpn_f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#endif
        kpn_static_for<0, MF>([&](auto mi) {
            constexpr int m = decltype(mi)::value;
            mfma(mi, kpn_ic<wsel(s)>{}, xp[cur]);
            if constexpr (s + 1 < KS16) {
                kpn_static_for<0, SPG>([&](auto ri) {   // slice k of the step: pair k / NSLICE (tile = pair % 2), slice k % NSLICE of it
                    constexpr int k = m * SPG + decltype(ri)::value, pair = k / NSLICE;
                    slice(kpn_ic<s + 1>{}, nxt, kpn_ic<pair % 2>{}, kpn_ic<pair / 2>{}, kpn_ic<k % NSLICE>{});
                });
                // each half's weight registers are reloaded right after that half's last MFMA has been issued (an MFMA
                // captures its operands at issue: scripts/mfma16_war_probe.hip)
                if constexpr (m == 2 * NPROD * H0 - 1) load_virtual(kpn_ic<s + WBUF>{}, kpn_ic<0>{});
                if constexpr (m == MF - 1) load_virtual(kpn_ic<s + WBUF>{}, kpn_ic<1>{});
            } else {
                tail_fn(mi);
                if constexpr (NEXT && m >= MF / 2) {      // the slices of the next layer's step 0 under the MFMAs of the second half
                    kpn_static_for<0, 2 * SPG>([&](auto ri) {   // all 8 NSLICE slices in half a step: twice as many per gap
                        constexpr int k = (m - MF / 2) * 2 * SPG + decltype(ri)::value, pair = k / NSLICE, t = pair % 2, j = pair / 2;
                        auto v0 = [&]() { return next_fn(kpn_ic<t>{}, kpn_ic<2 * j>{}); };
                        auto v1 = [&]() { return next_fn(kpn_ic<t>{}, kpn_ic<2 * j + 1>{}); };
                        kpn_h2_slice<SC, true, k % NSLICE, j>(pr, xn[t], v0, v1);
                    });
                }
            }
            KPN_SCHED_BARRIER();
        });
    });
}

// OPTION: layers1.0's 16 steps executed with a geometry step (stream steps 12..15: bilinear gathers) after every three encoding
// steps (stream steps 0..11), so that the gathers of the four waves of a CU, which run in near lockstep and share one L1, are
// spread out.  Measured: layers1.0 41.2 k instead of 45.2 k cycles per work item, the launch 1.3 % faster — and the different
// summation order moves one ray of the V = 16 oracle comparison (tests/test_gpu_parity.py::test_render_vs_oracle, tiny
// densities) from below to 1.3e-4, above the 1e-4 bar.  Not worth it: off by default (-DKPN_H2_L0_INTERLEAVE=1 switches it on).
#ifndef KPN_H2_L0_INTERLEAVE
#define KPN_H2_L0_INTERLEAVE 0
#endif
#ifndef KPN_H2_L0_AHEAD
#define KPN_H2_L0_AHEAD 2   // interleaved order: positions between a geometry step's gathers and its first use (2 or 3)
#endif
struct kpn_h2_l0_order {
    static constexpr int at(int p) { return KPN_H2_L0_INTERLEAVE ? ((p % 4 == 3) ? 12 + p / 4 : p - p / 4) : p; }
};
#ifdef KPN_H2_TIMING   // debug builds: cycles (s_memtime) per phase of the work items of one wave, summed: prologue, 4 layers, epilogue
__device__ unsigned long long kpn_h2_cycles[8];
#define KPN_H2_STAMP(i) do { const unsigned long long now_ = clock64(); if (blockIdx.x == 3 && threadIdx.x == 64) atomicAdd(&kpn_h2_cycles[i], now_ - stamp_); stamp_ = now_; } while (0)
#else
#define KPN_H2_STAMP(i) ((void)0)
#endif
#ifndef KPN_SIMT_EMU
#ifdef KPN_H2_NUM_VGPR   // experiment: fewer registers -> heavy scratch spills at one wave per SIMD
#ifndef KPN_H2_WAVES_PER_SIMD
#define KPN_H2_WAVES_PER_SIMD 1
#endif
#define KPN_H2_BOUNDS __launch_bounds__(256, KPN_H2_WAVES_PER_SIMD) __attribute__((amdgpu_num_vgpr(KPN_H2_NUM_VGPR)))
#else
#define KPN_H2_BOUNDS __launch_bounds__(256, 1)
#endif
#else
#define KPN_H2_BOUNDS
#endif
// POOL = false: a work item is (tile pair, view); the 64-vector of every (point, view) goes to the row scratch (ROWS layout,
//        kpn_field_shared.h) and the per-point kernel pools over the views — what the training passes keep for their backward.
// POOL = true : a work item is a tile pair walked through ALL its views, pooled on the fly: view v's 64-vectors update a weighted
//        Welford state (running mean and sum of weighted squared deviations, weights = the boundary-smooth view weights of
//        model.py:752-759) that lives in LDS (32 KB per wave: a lane's 2 x 64 floats for its two tiles), and only the pooled
//        mean / variance (PoolModule, utils.py:612-647, 731-748) is written: 128 floats per point instead of V x 64, one
//        dependent fetch instead of V in the per-point kernel, no pooling arithmetic there.  Welford, not sums of x and x^2: the
//        views agree to a few per cent on most points, var << mean^2, and the subtraction would cancel.
template <class SC, bool POOL>
__device__ __forceinline__ void kpn_geo_rows_pair_body(const kpn_scene_dev& sc, const kpn_points& ps, const float* __restrict__ wp,
                                                       const int* __restrict__ list, const int* __restrict__ count_ptr,
                                                       int* __restrict__ tickets, float* __restrict__ xscr, const kpn_batch& batch) {
    constexpr int NP = SC::NP;
    if (!kpn_batch_gate(batch, sc, wp)) return;            // the range guard (kpn_field_shared.h): wave-uniform, before anything else
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
    const int nwork = ((nbt + 1) >> 1) * (POOL ? 1 : sc.V);
    const kpn_tile_layout lay(POOL ? 1 : 0, sc.V);
    const uint32_t keep_bits = sc.keep & ((sc.V >= 32) ? 0xFFFFFFFFu : ((1u << sc.V) - 1u));
    const float neg_inv_two_sigma2 = -1.0f / sc.two_sigma2;   // exp(-d2 / 2 sigma^2) as one multiply per keypoint
    __shared__ __attribute__((aligned(16))) float bias_s[4][128];
    // POOL: the Welford state of this wave's tile pair, [32 slabs][64 lanes] float4: slabs 0..15 running mean (tile t, block b,
    // quad q -> slab 8 t + 4 b + q), 16..31 the weighted squared deviations.  Lane-private: no synchronisation.
    // Slabs 32, 33: the two points (x, y, z) + their weight sums so far; slab 34: this view's weights — values that live across the
    // whole view would otherwise sit in VGPRs through the four layers (the kernel is at the 256-register limit: held in registers
    // they came back as scratch reloads in the epilogue, 7 k cycles per view).
    __shared__ __attribute__((aligned(16))) float4 pool_s[POOL ? 4 * 35 * 64 : 1];
    float4* const pst = pool_s + (POOL ? ((threadIdx.x >> 6) * 35 * 64 + lane) : 0);
    {
        const int segs[4] = {SEG_G1_0A, SEG_G1_1, SEG_G1_2, SEG_G1_3};
        for (int i = threadIdx.x; i < 4 * 128; i += blockDim.x) {
            const int sg = i >> 7, k = i & 127;
            // log2-unit activations: the pre-activations of layers1.0-1.2 are kept scaled by 100 log2(e), biases included
            const float bscale = (KPN_H2_LOG2ACT && sg < 3) ? KPN_H2_ACT_SCALE : (sg == 3 ? SC::out_up : 1.0f);
            bias_s[sg][k] = k < kpn_seg_bfloats(segs[sg]) ? wp[kpn_seg_boff(segs[sg]) + k] * bscale : 0.0f;
        }
    }
    __syncthreads();
#ifdef KPN_H2_TIMING
    unsigned long long stamp_ = clock64();
#endif
    // The work items are software-pipelined (SC::PREFETCH: the fp16 scheme, which has the registers for it): the NEXT item's
    // ticket is drawn while layers1.0 runs, its two list entries are fetched while layers1.1 runs and its points while
    // layers1.2 runs (POOL: of the item's first view), so that the three dependent round trips (atomic -> list -> point: 5-6 k
    // cycles that nothing covered at one wave per SIMD) are off the critical path.  Clamped indices keep the look-ahead of a
    // ticket beyond the last item in bounds.
    int nx_ticket = 0;                    // lane 0: the raw result of the next ticket's atomic
    int nx_wi = 0;
    int64_t nx_n[2] = {0, 0};
    kpn_point_raw nx_raw[2];
    auto draw_ticket = [&]() { if (lane == 0) nx_ticket = atomicAdd(tickets + 0, 1); };
    auto fetch_list = [&]() {
        nx_wi = __shfl(nx_ticket, 0);
        const int npair = POOL ? nx_wi : nx_wi / sc.V;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int64_t ci = (int64_t)(t0 + 2 * npair + t) * KPN_TILE + p;
            if (ci >= count) ci = count - 1;
            nx_n[t] = (int64_t)list[ci];
        }
    };
    auto fetch_points = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t) kpn_point_fetch(ps, nx_n[t], nx_raw[t]);
    };
    draw_ticket(); fetch_list(); fetch_points();           // the first item: nothing to hide behind
#ifndef KPN_SIMT_EMU
    const bool stamp_clk = batch.clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
    if (stamp_clk) batch.clk[0] = clock64();
#endif
    for (;;) {
        KPN_H2_STAMP(6);
        const int wi = nx_wi;
        if (wi >= nwork) {
#ifndef KPN_SIMT_EMU
            if (stamp_clk) batch.clk[1] = clock64();
#endif
            return;
        }
        const int pair = POOL ? wi : wi / sc.V;
        const int v_begin = POOL ? 0 : wi - pair * sc.V, v_end = POOL ? sc.V : v_begin + 1;
        const bool has1 = 2 * pair + 1 < nbt;              // an odd batch ends in half a pair: tile 1 is computed, not stored
        float P[2][3], D[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) kpn_point_finish(ps, nx_raw[t], P[t], D[t]);
        bool prefetched = false;                           // the next item's ticket / list / points: under the first computed view
        bool first_view = true;
        if constexpr (POOL) {                              // the points and the (zero) weight sums wait in LDS between the views
            pst[32 * 64] = make_float4(P[0][0], P[0][1], P[0][2], 0.0f);
            pst[33 * 64] = make_float4(P[1][0], P[1][1], P[1][2], 0.0f);
        }
      for (int v = v_begin; v < v_end; ++v) {
        const float* tb = sc.table + (size_t)v * KPN_TBL_STRIDE;
        if constexpr (POOL) {
            const float4 p0 = pst[32 * 64], p1 = pst[33 * 64];
            P[0][0] = p0.x; P[0][1] = p0.y; P[0][2] = p0.z; P[1][0] = p1.x; P[1][1] = p1.y; P[1][2] = p1.z;
        }
        kpn_proj q[2];
        float4* dst[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tr = 2 * pair + t;
            q[t] = kpn_project(tb, P[t][0], P[t][1], P[t][2], sc);
            if constexpr (!POOL) dst[t] = reinterpret_cast<float4*>(xscr) + lay.row(tr, v) * 64 + lane;
        }
        if (!((sc.keep >> v) & 1u)) {                      // a dropped view: zero rows (its gather records: k_row_records)
            if constexpr (!POOL) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (t == 1 && !has1) break;
#pragma unroll
                    for (int k = 0; k < 8; ++k) dst[t][k * 64] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            continue;                                      // (POOL: weight 0 in the pooling, nothing to do)
        }
        // POOL: this view's boundary-smooth weight of the two points (model.py:752-758), parked in LDS until the epilogue
        if constexpr (POOL) pst[34 * 64] = make_float4(kpn_pix_weight_fast(q[0]), kpn_pix_weight_fast(q[1]), 0.0f, 0.0f);
        // the next item's ticket / list / points ride under the LAST view of this item (POOL: fetched under the first view they
        // would sit in 18 registers through all the other views)
        const bool pf = SC::PREFETCH && !prefetched && (!POOL || (keep_bits >> (v + 1)) == 0u);
        if (pf) draw_ticket();
        KPN_H2_STAMP(0);
        // ---- layers1.0 as ONE 16-step chain (HSEG_G1_0A and HSEG_G1_0B are adjacent, same step size): steps 0-11 one
        //      keypoint each (7 encoding values + a zero slot), steps 12-15 eight geo0 channels each ----
        static_assert(kpn_xseg_off(HSEG_G1_0B, NP) == kpn_xseg_off(HSEG_G1_0A, NP) + 12 * kpn_xseg_step_floats(HSEG_G1_0A, NP), "adjacent segments");
        kpn_f32x16 a0[2][4], a1[2][4];
        kpn_u32x4 xa[2][NP], xb[2][NP];                       // step-0 operands handed from one layer to the next
        {
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
            const float* g0 = sc.geo0 + (size_t)v * sc.g0h * sc.g0w * 64 + 8 * h;
            // The keypoint encoding of step n (weight w = exp(-d^2 / 2 sigma^2), sin/cos of pi z, doubled twice) is computed ONE
            // STEP AHEAD in four small stages that ride in the nearly empty slices of the step before (an encoding step has no
            // activation to compute): [parity of the step][tile] holds the finished dz, w, sin, cos.  sin / cos of pi z are
            // v_sin_f32 / v_cos_f32 of z / 2 revolutions (kpn_sincos_pi, kpn_device.h: as accurate as the 28-instruction
            // Cody-Waite + minimax form this replaced; measured 1.2 % of the kernel — the encoding steps were not VALU-bound).
            // Tried and dropped for the geometry steps: the taps fetched COALESCED, four whole texels per
            // instruction landing in LDS (global_load_lds_dwordx4, the texel index of a point by ds_bpermute, a rotated chunk order
            // for conflict-free reads): correct, 5 % SLOWER — 510 more instructions per work item in a kernel that is issue-bound,
            // although with every lane gathering the same texels (scripts/experiments/ablation_switches.patch: KPN_DBG_H2_SAMETAP) the kernel runs 12 % faster.
            float fdz[2][2], fw[2][2], fs1[2][2], fc1[2][2];
            float kcv[2][3] = {{kc[0], kc[1], kc[2]}, {kc[3], kc[4], kc[5]}};   // keypoints of steps 0 and 1 (this half's 12)
            float tdx[2], tdy[2], ty[2], targ[2], ts2[2], tc2[2];
            float4 raw[2][8];                               // geo0: the four taps of two float4 of channels
            auto pe_stage = [&](auto ni, auto ti, auto gi) {   // stage gi (0..9; 4..9 empty) of the encoding of keypoint step ni, tile ti
                constexpr int n = decltype(ni)::value, t = decltype(ti)::value, g = decltype(gi)::value, b = n & 1;
                if constexpr (g == 0) {
                    // the keypoint's camera-frame coordinates were fetched two steps ago (a load followed by its use costs a full
                    // L2 round trip when no other wave shares the SIMD); tile 1 is the last user: it refills the slot for step n+2
                    tdx[t] = RSUB(cx[t], kcv[b][0]); tdy[t] = RSUB(cy[t], kcv[b][1]); fdz[b][t] = RSUB(cz[t], kcv[b][2]);
                    ty[t] = fdz[b][t] * 0.5f;               // pi z in revolutions
                    if constexpr (t == 1 && n + 2 < 12) { kcv[b][0] = kc[(n + 2) * 3 + 0]; kcv[b][1] = kc[(n + 2) * 3 + 1]; kcv[b][2] = kc[(n + 2) * 3 + 2]; }
                } else if constexpr (g == 1) {
                    const float d2 = RADD(RADD(RMUL(tdx[t], tdx[t]), RMUL(tdy[t], tdy[t])), RMUL(fdz[b][t], fdz[b][t]));
                    targ[t] = d2 * neg_inv_two_sigma2;
                } else if constexpr (g == 2) {
                    fw[b][t] = kpn_fast_exp(targ[t]);
#ifndef KPN_SIMT_EMU
                    fs1[b][t] = __builtin_amdgcn_sinf(ty[t]);
#else
                    fs1[b][t] = (float)sin(6.28318530717958647692 * (double)ty[t]);
#endif
                } else if constexpr (g == 3) {
#ifndef KPN_SIMT_EMU
                    fc1[b][t] = __builtin_amdgcn_cosf(ty[t]);
#else
                    fc1[b][t] = (float)cos(6.28318530717958647692 * (double)ty[t]);
#endif
                }
            };
            auto geo_loads = [&](auto si, auto ti, auto fi) {   // the taps of float4 f (0/1) of geo step s: channels 16(s-12) + 8h + 4f ..
                constexpr int s = decltype(si)::value, t = decltype(ti)::value, f = decltype(fi)::value;
                const float* gp = g0 + 16 * (s - 12) + 4 * f;
                const kpn_taps& w = tp[t];
                raw[t][4 * f + 0] = *reinterpret_cast<const float4*>(gp + (size_t)w.o00 * 64);
                raw[t][4 * f + 1] = *reinterpret_cast<const float4*>(gp + (size_t)w.o01 * 64);
                raw[t][4 * f + 2] = *reinterpret_cast<const float4*>(gp + (size_t)w.o10 * 64);
                raw[t][4 * f + 3] = *reinterpret_cast<const float4*>(gp + (size_t)w.o11 * 64);
            };
            // prologue of the work item: the encoding of step 0 (that of step 1 rides in the production of step 0's operands)
            kpn_static_for<0, 10>([&](auto gi) {
                pe_stage(kpn_ic<0>{}, kpn_ic<0>{}, gi); pe_stage(kpn_ic<0>{}, kpn_ic<1>{}, gi);
            });
            kpn_mfma16_layer2<SC, 16, 4, 0, false, KPN_H2_LOOKAHEAD, kpn_h2_l0_order>(wp + kpn_xseg_off(HSEG_G1_0A, NP), lane,
                [&](auto pi, auto ti, auto ei) -> float {       // value e of the step at position p
                    constexpr int s = kpn_h2_l0_order::at(decltype(pi)::value), t = decltype(ti)::value, e = decltype(ei)::value, b = s & 1;
                    if constexpr (s < 12) {
                        if constexpr (e == 0) { return fdz[b][t] * fw[b][t];
                        } else if constexpr (e == 1) { return fs1[b][t] * fw[b][t];
                        } else if constexpr (e == 2) { return fc1[b][t] * fw[b][t];
                        } else if constexpr (e == 3) {
                            ts2[t] = 2.0f * fs1[b][t] * fc1[b][t]; tc2[t] = 1.0f - 2.0f * fs1[b][t] * fs1[b][t];
                            return ts2[t] * fw[b][t];
                        } else if constexpr (e == 4) { return tc2[t] * fw[b][t];
                        } else if constexpr (e == 5) { return (2.0f * ts2[t] * tc2[t]) * fw[b][t];
                        } else if constexpr (e == 6) { return (1.0f - 2.0f * ts2[t] * ts2[t]) * fw[b][t];
                        } else { return 0.0f; }
                    } else {                                   // one blended channel: same tap order as ATen (nw, ne, sw, se)
                        constexpr int f = e / 4, c = e % 4;
                        const kpn_taps& w = tp[t];
                        auto comp = [](const float4& x) { return c == 0 ? x.x : (c == 1 ? x.y : (c == 2 ? x.z : x.w)); };
                        return RADD(RADD(RADD(RMUL(comp(raw[t][4 * f + 0]), w.w00), RMUL(comp(raw[t][4 * f + 1]), w.w01)),
                                         RMUL(comp(raw[t][4 * f + 2]), w.w10)), RMUL(comp(raw[t][4 * f + 3]), w.w11));
                    }
                },
                [&](auto pi, auto ti, auto ji, auto qi) {      // the slices an encoding / geometry step leaves nearly empty
                    constexpr int p0 = decltype(pi)::value, t = decltype(ti)::value, j = decltype(ji)::value, qq = decltype(qi)::value;
                    constexpr int g = 3 * j + qq - 2;          // slices (j,q) = (0,2) .. (3,2) -> stages 0 .. 9
                    // the encoding of the NEXT position, if it is an encoding step
                    if constexpr (p0 + 1 < 16 && g >= 0) {
                        constexpr int nx = kpn_h2_l0_order::at(p0 + 1);
                        if constexpr (nx < 12) pe_stage(kpn_ic<nx>{}, ti, kpn_ic<g>{});
                    }
                    // The taps of a geometry step go into the single `raw` buffer: its first float4 of channels may be fetched once
                    // values 0..3 of the previous geometry step exist (pair 1's spare slice), the second after values 4..7 (pair 3's).
                    // Consecutive geometry steps (the default order): one position ahead; the FIRST one nine positions ahead, while the
                    // encoding steps leave the registers free — a gather from the feature maps is thousands of cycles away.
                    // Interleaved order: two positions ahead (the previous geometry step ended three positions ago).
                    if constexpr (qq == 2 && (j == 1 || j == 3)) {
                        if constexpr (KPN_H2_L0_INTERLEAVE) {
                            if constexpr (p0 + KPN_H2_L0_AHEAD < 16) {
                                constexpr int nx2 = kpn_h2_l0_order::at(p0 + KPN_H2_L0_AHEAD);
                                if constexpr (nx2 >= 12) geo_loads(kpn_ic<nx2>{}, ti, kpn_ic<j / 2>{});
                            }
                        } else {
                            if constexpr (p0 == 3) geo_loads(kpn_ic<12>{}, ti, kpn_ic<j / 2>{});
                            if constexpr (p0 + 1 > 12 && p0 + 1 < 16) geo_loads(kpn_ic<p0 + 1>{}, ti, kpn_ic<j / 2>{});
                        }
                    }
                },
                [&](auto mi) {                                 // last step of layers1.0: the biases of layers1.1
                    constexpr int m = decltype(mi)::value;
                    if constexpr (m < 2) kpn_load_bias<4>(bias_s[1], h, a1[m]);
                },
                [&](auto ti, auto ei) -> float { return a0[decltype(ti)::value][0][decltype(ei)::value]; },
                a0, xa, xb);
        }
        KPN_H2_STAMP(1);
        if (pf) fetch_list();
        // chained step s of a 128-vector: registers 8(s%2)..+7 of block s/2
        kpn_f32x16 a2[2][4];
        float4 hraw[2][4];                                  // the four taps of the 4 hd channels of this half (layers1.2, step 8)
        float hd[2][4];
        kpn_taps tp1[2];
        kpn_mfma16_layer2<SC, 8, 4, 8, KPN_H2_LOOKAHEAD, KPN_H2_LOOKAHEAD>(wp + kpn_xseg_off(HSEG_G1_1, NP), lane,
            [&](auto si, auto ti, auto ei) -> float {
                constexpr int s = decltype(si)::value, t = decltype(ti)::value, e = decltype(ei)::value;
                return a0[t][s / 2][(s % 2) * 8 + e];
            },
            [](auto, auto, auto, auto) {},
            [&](auto mi) {                                     // last step: biases of layers1.2, the hd taps
                constexpr int m = decltype(mi)::value;
                if constexpr (m < 2) {
                    kpn_load_bias<4>(bias_s[2], h, a2[m]);
                    tp1[m] = kpn_make_taps(q[m].xn, q[m].yn, sc.g1h, sc.g1w);
                } else if constexpr (m < 4) {
                    constexpr int t = m - 2;
                    const float* gp = sc.geo1 + (size_t)v * sc.g1h * sc.g1w * 8 + 4 * h;
                    hraw[t][0] = *reinterpret_cast<const float4*>(gp + (size_t)tp1[t].o00 * 8);
                    hraw[t][1] = *reinterpret_cast<const float4*>(gp + (size_t)tp1[t].o01 * 8);
                    hraw[t][2] = *reinterpret_cast<const float4*>(gp + (size_t)tp1[t].o10 * 8);
                    hraw[t][3] = *reinterpret_cast<const float4*>(gp + (size_t)tp1[t].o11 * 8);
                } else if constexpr (m >= 20 && m < 22) {
                    constexpr int t = m - 20;
                    const kpn_taps& w = tp1[t];
                    hd[t][0] = RADD(RADD(RADD(RMUL(hraw[t][0].x, w.w00), RMUL(hraw[t][1].x, w.w01)), RMUL(hraw[t][2].x, w.w10)), RMUL(hraw[t][3].x, w.w11));
                    hd[t][1] = RADD(RADD(RADD(RMUL(hraw[t][0].y, w.w00), RMUL(hraw[t][1].y, w.w01)), RMUL(hraw[t][2].y, w.w10)), RMUL(hraw[t][3].y, w.w11));
                } else if constexpr (m >= 22 && m < 24) {
                    constexpr int t = m - 22;
                    const kpn_taps& w = tp1[t];
                    hd[t][2] = RADD(RADD(RADD(RMUL(hraw[t][0].z, w.w00), RMUL(hraw[t][1].z, w.w01)), RMUL(hraw[t][2].z, w.w10)), RMUL(hraw[t][3].z, w.w11));
                    hd[t][3] = RADD(RADD(RADD(RMUL(hraw[t][0].w, w.w00), RMUL(hraw[t][1].w, w.w01)), RMUL(hraw[t][2].w, w.w10)), RMUL(hraw[t][3].w, w.w11));
                }
            },
            [&](auto ti, auto ei) -> float { return a1[decltype(ti)::value][0][decltype(ei)::value]; },
            a1, xb, xa);
        KPN_H2_STAMP(2);
        if constexpr (!POOL) { if (pf) fetch_points(); }   // POOL: after the views (the raw points would occupy 14 registers through layers1.2 / 1.3)
        kpn_f32x16 acc[2][2];
        kpn_mfma16_layer2<SC, 9, 4, 8, KPN_H2_LOOKAHEAD, KPN_H2_LOOKAHEAD>(wp + kpn_xseg_off(HSEG_G1_2, NP), lane,
            [&](auto si, auto ti, auto ei) -> float {
                constexpr int s = decltype(si)::value, t = decltype(ti)::value, e = decltype(ei)::value;
                if constexpr (s < 8) return a1[t][s / 2][(s % 2) * 8 + e];
                else if constexpr (e < 4) return hd[t][e];
                else return 0.0f;
            },
            [](auto, auto, auto, auto) {},
            [&](auto mi) {                                     // last step: biases of layers1.3
                constexpr int m = decltype(mi)::value;
                if constexpr (m < 2) kpn_load_bias<2>(bias_s[3], h, acc[m]);
            },
            [&](auto ti, auto ei) -> float { return a2[decltype(ti)::value][0][decltype(ei)::value]; },
            a2, xa, xb);
        KPN_H2_STAMP(3);
        kpn_mfma16_layer2<SC, 8, 2, 8, KPN_H2_LOOKAHEAD, false>(wp + kpn_xseg_off(HSEG_G1_3, NP), lane,
            [&](auto si, auto ti, auto ei) -> float {
                constexpr int s = decltype(si)::value, t = decltype(ti)::value, e = decltype(ei)::value;
                return a2[t][s / 2][(s % 2) * 8 + e];
            },
            [](auto, auto, auto, auto) {}, [](auto) {}, [](auto, auto) -> float { return 0.0f; }, acc, xb, xa);
        KPN_H2_STAMP(4);
        prefetched |= pf;
        if constexpr (!POOL) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t == 1 && !has1) break;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd)
                        dst[t][(b * 4 + qd) * 64] =   // (the fp16 scheme's layers1.3 is packed times 2^10: exact power-of-two unscale)
                            make_float4(acc[t][b][4 * qd + 0] * SC::out_down, acc[t][b][4 * qd + 1] * SC::out_down,
                                        acc[t][b][4 * qd + 2] * SC::out_down, acc[t][b][4 * qd + 3] * SC::out_down);
            }
        } else {
            // Weighted Welford update of the pooled statistics with this view's rows x (weight w = the view's boundary-smooth
            // weight, un-normalised; X = 2^10 x in the fp16 scheme, unscaled at the end):
            //     W' = W + w,  r = w / W',  d = X - mean,  mean' = mean + r d,  M2' = M2 + w (1 - r) d^2      (4 instructions per value)
            // The last kept view finishes: the reference normalises the weights by (sum + 1e-6) (model.py:759), so with S = sum w,
            //     mean_ref = mu S / (S + 1e-6),   var_ref = sum_v pw_v (x_v - mean_ref)^2 = (M2 + S (mu 1e-6 / (S + 1e-6))^2) / (S + 1e-6)
            // (the second term is NOT negligible: near a frustum corner the weights are a product of three sigmoids, 3e-7, the same
            // size as the 1e-6 — the reference's mean is then far from mu and its variance is mostly this term)
            const bool last_view = (keep_bits >> (v + 1)) == 0u;
            // ONE branch selects among four straight-line bodies (first / last view known at compile time inside each): written as
            // run-time tests inside the slab loop, hipcc kept them there — twelve branches per slab, 64 accumulator moves between
            // the arms, 4-5 k cycles per view for 330 instructions' worth of work at one wave per SIMD.
            auto welford = [&](auto first_c, auto last_c) {
                constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
                const float4 wv4 = pst[34 * 64];
                // (the lane offset goes through an opaque statement: formed from loop invariants only, the eight store addresses of a
                // tile were hoisted to the top of the work item and held — spilled — through every view)
                int lane_late = lane;
#ifndef KPN_SIMT_EMU
                asm volatile("" : "+v"(lane_late));
#endif
#pragma unroll
                for (int t = 0; t < 2; ++t) dst[t] = reinterpret_cast<float4*>(xscr) + lay.tile(2 * pair + t) * 64 + lane_late;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float4 pw4 = pst[(32 + t) * 64];
                    const float w = t == 0 ? wv4.x : wv4.y;
                    const float wn = pw4.w + w;
                    const float r = wn > 0.0f ? w / wn : 0.0f;
                    const float c = w * (1.0f - r);
                    const float inv = 1.0f / (wn + 1e-6f);
                    const float ms = wn * inv * SC::out_down, vs = inv * (SC::out_down * SC::out_down);   // exact powers of two folded in
                    const float om2 = wn * (1e-6f * inv) * (1e-6f * inv);                                   // S (1 - s)^2
                    const bool store = t == 0 || has1;
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int qd = 0; qd < 4; ++qd) {
                            const int slab = 8 * t + 4 * b + qd;
                            float mu[4], m2[4];
                            if constexpr (FIRST) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) { mu[e] = acc[t][b][4 * qd + e]; m2[e] = 0.0f; }
                            } else {
                                const float4 a = pst[slab * 64], cc = pst[(16 + slab) * 64];
                                const float am[4] = {a.x, a.y, a.z, a.w}, cm[4] = {cc.x, cc.y, cc.z, cc.w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float d = acc[t][b][4 * qd + e] - am[e];
                                    mu[e] = fmaf(r, d, am[e]);
                                    m2[e] = fmaf(c * d, d, cm[e]);
                                }
                            }
                            if constexpr (!LAST) {
                                pst[slab * 64] = make_float4(mu[0], mu[1], mu[2], mu[3]);
                                if constexpr (!FIRST) pst[(16 + slab) * 64] = make_float4(m2[0], m2[1], m2[2], m2[3]);
                                else pst[(16 + slab) * 64] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            } else if (store) {
                                dst[t][(4 * b + qd) * 64] = make_float4(mu[0] * ms, mu[1] * ms, mu[2] * ms, mu[3] * ms);
                                dst[t][(8 + 4 * b + qd) * 64] = make_float4(fmaf(om2 * mu[0], mu[0], m2[0]) * vs, fmaf(om2 * mu[1], mu[1], m2[1]) * vs,
                                                                            fmaf(om2 * mu[2], mu[2], m2[2]) * vs, fmaf(om2 * mu[3], mu[3], m2[3]) * vs);
                            }
                        }
                    if constexpr (!LAST) pst[(32 + t) * 64] = make_float4(pw4.x, pw4.y, pw4.z, wn);
                }
            };
            if (first_view) { if (last_view) welford(kpn_ic<1>{}, kpn_ic<1>{}); else welford(kpn_ic<1>{}, kpn_ic<0>{}); }
            else { if (last_view) welford(kpn_ic<0>{}, kpn_ic<1>{}); else welford(kpn_ic<0>{}, kpn_ic<0>{}); }
            first_view = false;
        }
      }   // views of the work item
        if (!prefetched) { draw_ticket(); fetch_list(); fetch_points(); }
        else if constexpr (POOL) fetch_points();
    }
}

// rows mode 2: three bf16 pieces, six products (fp32's exponent range)
__global__ KPN_H2_BOUNDS void k_geo_rows_h2(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                            const int* __restrict__ list, const int* __restrict__ count_ptr,
                                            int* __restrict__ tickets, float* __restrict__ xscr, kpn_batch batch) {
    kpn_geo_rows_pair_body<kpn_sc_bf16x3, false>(sc, ps, wp, list, count_ptr, tickets, xscr, batch);
}
// rows mode 3 (the default): two fp16 pieces, three products
__global__ KPN_H2_BOUNDS void k_geo_rows_f2(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                            const int* __restrict__ list, const int* __restrict__ count_ptr,
                                            int* __restrict__ tickets, float* __restrict__ xscr, kpn_batch batch) {
    kpn_geo_rows_pair_body<kpn_sc_f16x2, false>(sc, ps, wp, list, count_ptr, tickets, xscr, batch);
}
// the same two, pooling over the views inside the kernel (POOL layout of the scratch): the render and query passes
__global__ KPN_H2_BOUNDS void k_geo_rows_h2p(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                             const int* __restrict__ list, const int* __restrict__ count_ptr,
                                             int* __restrict__ tickets, float* __restrict__ xscr, kpn_batch batch) {
    kpn_geo_rows_pair_body<kpn_sc_bf16x3, true>(sc, ps, wp, list, count_ptr, tickets, xscr, batch);
}
__global__ KPN_H2_BOUNDS void k_geo_rows_f2p(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                             const int* __restrict__ list, const int* __restrict__ count_ptr,
                                             int* __restrict__ tickets, float* __restrict__ xscr, kpn_batch batch) {
    kpn_geo_rows_pair_body<kpn_sc_f16x2, true>(sc, ps, wp, list, count_ptr, tickets, xscr, batch);
}

// The colour head's gather records of a batch (slabs 8, 9 of every (tile, view) block of the row scratch; kpn_row_record_a / _b,
// model.py:806-832), for the pair-tile rows kernels: inside those the two parts ran as a divergent branch of the half-waves
// (~800 issue slots per work item in a VALU-bound stream plus exposed tap latency).  Here a lane owns one (point, view) pair and
// computes BOTH parts without divergence, with enough waves in flight to hide the taps: a wavefront takes the 64 points of a
// tile pair for ONE view (the view's table entries are wave-uniform: scalar loads) and writes part A to the row's h = 0 slot
// and part B to its h = 1 slot.
__global__ __launch_bounds__(256) void k_row_records(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp, const int* __restrict__ list,
                                                     const int* __restrict__ count_ptr, float* __restrict__ xscr, kpn_batch batch) {
    const int lane = threadIdx.x & 63, p = lane & 31, tsel = lane >> 5;
    // (KPN_RUN_IF_UNSAFE: the records of EVERY point for the fp32-range kernels behind a density-first pass, whose own records
    // cover the live points only — returns at once unless the range guard evaluates the batch again)
    if (batch.cond != KPN_RUN_ALWAYS && !kpn_batch_gate(batch, sc, wp)) return;
    const int count = *count_ptr;
    int t0, t1;
    if (!kpn_batch_range(batch, (count + KPN_TILE - 1) / KPN_TILE, t0, t1)) return;
    const int nbt = t1 - t0;
    const int npairs = (nbt + 1) >> 1;
    const kpn_tile_layout lay(batch.pool, sc.V);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    // A wavefront takes a tile pair through ALL its views (the point is fetched once, not once per view) and fetches the NEXT
    // pair's list entry and point under the current pair's views: list -> point -> projection -> taps is a chain of dependent
    // round trips, and with one (pair, view) per iteration nothing overlapped them (0.20 ms per launch for 64 B per row).
    auto fetch = [&](int pair, kpn_point_raw& raw) {
        int64_t ci = (int64_t)(t0 + 2 * pair + tsel) * KPN_TILE + p;
        if (ci >= count) ci = count - 1;
        kpn_point_fetch(ps, (int64_t)list[ci], raw);
    };
    kpn_point_raw raw, raw_next;
    if (wave < npairs) fetch(wave, raw);
    for (int pair = wave; pair < npairs; pair += nwaves) {      // wave-uniform
        if (pair + nwaves < npairs) fetch(pair + nwaves, raw_next);
        const int tr = 2 * pair + tsel;
        float P[3], D[3];
        kpn_point_finish<true>(ps, raw, P, D);               // strict: the reference's pixel coordinates bit for bit (kpn_row_record_a)
        for (int v = 0; v < sc.V; ++v) {
            const float* tb = sc.table + (size_t)v * KPN_TBL_STRIDE;
            const kpn_proj q = kpn_project<true>(tb, P[0], P[1], P[2], sc);
            float4 a0, a1, b0, b1;
            kpn_row_record_a<true>(sc, tb, v, q, P, D, a0, a1);
            kpn_row_record_b<true>(sc, v, q, b0, b1);
            if (tr < nbt) {                                      // an odd batch ends in half a pair
                float4* rec = reinterpret_cast<float4*>(xscr) + lay.rec(tr, v) * 64;
                rec[p] = a0; rec[32 + p] = b0; rec[64 + p] = a1; rec[64 + 32 + p] = b1;
            }
        }
        raw = raw_next;
    }
}

// The same records for the LIVE points of a density-first render pass only (field_kernels.hip, PHASE): a lane takes one entry of
// the batch's live list (pass A, k_density_h: slot = tile * 32 + point of the row scratch), forms that point's records for every
// view and writes them where the full kernel would have — pass B (k_colour_h*) addresses the scratch by the same slot.  The
// records of dead points (density exactly 0: their colour never reaches the image) are not formed at all.
__global__ __launch_bounds__(256) void k_row_records_live(kpn_scene_dev sc, kpn_points ps, const float* __restrict__ wp,
                                                          const int* __restrict__ list, const int* __restrict__ count_ptr,
                                                          const int* __restrict__ tickets, const int* __restrict__ live,
                                                          float* __restrict__ xscr, kpn_batch batch) {
    if (batch.cond != KPN_RUN_ALWAYS && !kpn_batch_gate(batch, sc, wp)) return;
    const int count = *count_ptr;
    int t0, t1;
    if (!kpn_batch_range(batch, (count + KPN_TILE - 1) / KPN_TILE, t0, t1)) return;
    const int nlive = tickets[2];
    const kpn_tile_layout lay(batch.pool, sc.V);
    const int stride = gridDim.x * blockDim.x;
    auto fetch = [&](int e, int& slot, kpn_point_raw& raw) {
        slot = live[e < nlive ? e : nlive - 1];
        kpn_point_fetch(ps, (int64_t)list[t0 * KPN_TILE + slot], raw);
    };
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    // wave-uniform trip count (the view tables are read with scalar loads): the lanes beyond the list redo its last entry
    const int e_wave = e - (threadIdx.x & 63);
    kpn_point_raw raw, raw_next;
    int slot = 0, slot_next = 0;
    if (e_wave < nlive) fetch(e, slot, raw);
    for (int ew = e_wave; ew < nlive; ew += stride, e += stride) {
        if (ew + stride < nlive) fetch(e + stride, slot_next, raw_next);
        float P[3], D[3];
        kpn_point_finish<true>(ps, raw, P, D);               // strict, as k_row_records
        const int tr = slot >> 5, pp = slot & 31;
        for (int v = 0; v < sc.V; ++v) {
            const float* tb = sc.table + (size_t)v * KPN_TBL_STRIDE;
            const kpn_proj q = kpn_project<true>(tb, P[0], P[1], P[2], sc);
            float4 a0, a1, b0, b1;
            kpn_row_record_a<true>(sc, tb, v, q, P, D, a0, a1);
            kpn_row_record_b<true>(sc, v, q, b0, b1);
            if (e < nlive) {
                float4* rec = reinterpret_cast<float4*>(xscr) + lay.rec(tr, v) * 64;
                rec[pp] = a0; rec[32 + pp] = b0; rec[64 + pp] = a1; rec[64 + 32 + pp] = b1;
            }
        }
        raw = raw_next;
        slot = slot_next;
    }
}
