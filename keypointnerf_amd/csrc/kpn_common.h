// kpn_common.h — shared definitions of the gfx950 ray-march kernels.
//
// Written for CDNA4 only: 64-lane wavefronts, v_mfma_f32_32x32x2_f32, no CUDA/other-backend paths.
// The single build-mode switch, KPN_SIMT_EMU, selects the host wave64 emulator used by the CPU test
// suite (tests/simt/simt.h); it replaces the launch statement and a handful of intrinsics, never the
// kernel logic.
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifndef KPN_SIMT_EMU
#include <hip/hip_runtime.h>
typedef float kpn_f32x16 __attribute__((ext_vector_type(16)));
typedef float kpn_f32x4 __attribute__((ext_vector_type(4)));
#define KPN_LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, 0, (hipStream_t)(stream), __VA_ARGS__)
// raw v_exp_f32 / v_log_f32 (base 2, ~1 ulp, no denormal fix-ups: callers keep arguments in range)
__device__ __forceinline__ float kpn_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float kpn_log2(float x) { return __builtin_amdgcn_logf(x); }
typedef const __attribute__((address_space(1))) kpn_f32x4* kpn_gptr4;  // global (not flat) loads
typedef const __attribute__((address_space(3))) kpn_f32x4* kpn_lptr4;  // LDS loads (ds_read_b128)
#define KPN_GLOBAL4(p) ((kpn_gptr4)(p))
#define KPN_LDS4(p) ((kpn_lptr4)(p))
#else
#include <math.h>
static inline float kpn_exp2(float x) { return exp2f(x); }
static inline float kpn_log2(float x) { return log2f(x); }
typedef const kpn_f32x4* kpn_gptr4;
typedef const kpn_f32x4* kpn_lptr4;
#define KPN_GLOBAL4(p) ((kpn_gptr4)(p))
#define KPN_LDS4(p) ((kpn_lptr4)(p))
#endif
// lanes of one wavefront exchanging data through LDS: LDS operations of a wave execute in order, so only the
// compiler must be kept from moving accesses across the exchange point
#ifndef KPN_SIMT_EMU
#define KPN_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                             __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
#define KPN_WAVE_SYNC() simt::wave_sync()
#endif
#ifndef KPN_SIMT_EMU
// hardware global_atomic_add_f32 (no CAS loop); the sum order is not deterministic
__device__ __forceinline__ void kpn_atomic_add(float* p, float v) {
    unsafeAtomicAdd(p, v);
}
#else
static inline void kpn_atomic_add(float* p, float v) {
    uint32_t* u = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED), nw;
    do { float f; memcpy(&f, &old, 4); f += v; memcpy(&nw, &f, 4); }
    while (!__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
}
#endif
__device__ __forceinline__ float kpn_fast_exp(float x) { return kpn_exp2(x * 1.44269504088896341f); }

// ---- split-bf16 operands for v_mfma_f32_32x32x16_bf16 (k_geo_rows_h) ----
// An fp32 value x is carried as three bf16 pieces h + m + l (|x - (h+m+l)| <= 2^-24 |x|: the residuals are exact in
// fp32); a product term set  w*x ~= wh*xh + wh*xm + wm*xh + wm*xm + wh*xl + wl*xh  (everything above 2^-24 relative)
// is six MFMAs at 16x the fp32-MFMA rate.  Operand layout (verified on the MI355X with asymmetric operands):
//   A: lane l holds A[i = l&31][k = 8(l>>5) + e], e = 0..7;  B: lane l holds B[k = 8(l>>5) + e][j = l&31];  D as 32x32x2.
#ifndef KPN_SIMT_EMU
typedef __bf16 kpn_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void kpn_split3(const float (&x)[8], kpn_bf16x8& h, kpn_bf16x8& m, kpn_bf16x8& l) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 a = (__bf16)x[i];
        const float r1 = x[i] - (float)a;
        const __bf16 b = (__bf16)r1;
        h[i] = a; m[i] = b; l[i] = (__bf16)(r1 - (float)b);
    }
}
__device__ __forceinline__ kpn_bf16x8 kpn_as_bf16x8(kpn_f32x4 v) { return __builtin_bit_cast(kpn_bf16x8, v); }
typedef __bf16 kpn_bf16_t;
__device__ __forceinline__ kpn_bf16_t kpn_to_bf(float f) { return (__bf16)f; }   // round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ float kpn_bf_to_f(kpn_bf16_t b) { return (float)b; }
#define KPN_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
// v_mfma_f32_32x32x16_f16 on raw dwords (A: a weight stream's float4, B: four packed pairs); operand maps as the bf16 form
typedef uint32_t kpn_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 kpn_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ kpn_f32x16 kpn_mfma_f16(kpn_f32x4 a, kpn_u32x4 b, kpn_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(kpn_f16x8, a), __builtin_bit_cast(kpn_f16x8, b), c, 0, 0, 0);
}
// eight fp32 values -> two fp16 pieces each, x = h + l to 2^-23 relative (absolute floor 2^-24: fp16's subnormal quantum):
// h = RNE(x) (v_cvt_pk_f16_f32), x - h formed exactly by ONE v_fma_mix_f32 that reads the fp16 half in place, l = RNE(x - h)
// (scripts/f16_split_probe.hip): four instructions per pair of values.
//
// COMPILER-SELECTED instructions, not inline asm (round 5).  Rounds 3-4 wrote the four instructions as (movable) asm statements,
// and k_fuse_color_h came out WRONG AND NON-DETERMINISTIC on the MI355X in one of two equivalent product orders.  The cause
// (scripts/isa_asm_hazards.py on the two builds, scripts/repro_asm_waw_hazard.hip): hipcc's hazard recogniser treats an INLINEASM
// as one opaque instruction and applies none of its MFMA <-> VALU wait-state rules to the instructions inside the string.  The
// register allocator had put the statement's outputs ("=&v") into DEAD registers of the destination tuple of an MFMA issued three
// states earlier (a layer whose output block is only partly used: ray_encoder.2's rows 32..34 use 3 of 16 registers), and the
// MFMA's write-back, eight passes later, overwrote the freshly converted pieces — XDL write -> VALU write (WAW), a pair hipcc pads
// with wait states only when it can see the VALU instruction.  Which registers the allocator picks changes with every edit; the
// shipped order was right by luck of allocation, not by construction.  With the conversions, the fused subtract and the packing
// visible to the compiler every such pair is padded by hipcc itself.
//   * v_cvt_pk_f16_f32 is what gfx950 selects for fptrunc <2 x float> -> <2 x half>;
//   * v_fma_mix_f32 is selected for fma(fpext(half), b, c); b = -1.0 must be OPAQUE (an SGPR written by an asm s_mov: an SALU
//     instruction, no VALU hazard class applies) or the DAG combiner rewrites fma(h, -1, x) into a subtract of a separately
//     converted half (two more instructions per pair);
//   * the translation unit is compiled with -fno-slp-vectorize (build.py), or the two fused subtracts of a pair become one
//     v_pk_fma_f32 behind two v_cvt_f32_f16 (and packed fp32 beside MFMAs is an anti-lever on this chip).
typedef _Float16 kpn_h2 __attribute__((ext_vector_type(2)));
typedef float kpn_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float kpn_opaque_minus_one() {
    float m;
    asm("s_mov_b32 %0, -1.0" : "=s"(m));   // pure: hoisted out of loops and merged by CSE
    return m;
}
#ifdef KPN_PRECISION_PROBE
// PRECISION-BUDGET PROBE BUILDS ONLY (scripts/precision_budget.py, profiles/r06_precision_budget.md; never defined in the shipped
// library): bit 2 * id + 0 of the mask drops the LO piece of the B operand (activations) of layer `id`, bit 2 * id + 1 the lo piece
// of its A operand (weights) — the product hl resp. lh then multiplies by zero, i.e. that operand is carried as ONE fp16 piece.
// ids: 0..3 = layers1.0 .. 1.3 (rows kernel), 4 / 5 = the keypoint-encoding / the sampled-channel K steps of layers1.0 only,
// 8 + (SEG - SEG_G2_0) = the per-point kernel's layers.  One copy of the mask per translation unit (kpn_probe_set_mask sets both).
static __device__ unsigned long long kpn_probe_mask_dev;
#define KPN_PROBE(id, operand) (((kpn_probe_mask_dev) >> (2 * (id) + (operand))) & 1ull)
#endif
__device__ __forceinline__ void kpn_split_f16x8(const float (&x)[8], kpn_u32x4& h, kpn_u32x4& l) {
    const float m1 = kpn_opaque_minus_one();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const kpn_f2 v = {x[2 * j], x[2 * j + 1]};
        const kpn_h2 hh = __builtin_convertvector(v, kpn_h2);
        const kpn_f2 r = {__builtin_fmaf((float)hh[0], m1, v[0]), __builtin_fmaf((float)hh[1], m1, v[1])};   // exact
        const kpn_h2 ll = __builtin_convertvector(r, kpn_h2);
        h[j] = __builtin_bit_cast(uint32_t, hh);
        l[j] = __builtin_bit_cast(uint32_t, ll);
    }
}
// eight fp32 values -> three bf16 pieces each (x = h + m + l to 2^-24 relative, fp32's exponent range): the packed form of
// kpn_split3 — v_cvt_pk_bf16_f32 converts two values at once, 5.5 instead of 7.5 instructions per value — for kernels whose
// operands have a gradient's dynamic range (k_geo_rows_bwd, k_weight_grad).  Compiler-selected for the same reason as above
// (gfx950 selects v_cvt_pk_bf16_f32 for fptrunc <2 x float> -> <2 x bfloat>; a bf16's fp32 value is its bit pattern shifted).
typedef __bf16 kpn_b2 __attribute__((ext_vector_type(2)));
template <bool TO_MFMA = true>
__device__ __forceinline__ void kpn_split_bf16x8(const float (&x)[8], kpn_bf16x8& h, kpn_bf16x8& m, kpn_bf16x8& l) {
    kpn_u32x4 ph, pm, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const kpn_f2 v = {x[2 * j], x[2 * j + 1]};
        const uint32_t a = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, kpn_b2));
        const kpn_f2 r = {v[0] - __builtin_bit_cast(float, a << 16), v[1] - __builtin_bit_cast(float, a & 0xffff0000u)};
        const uint32_t b = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, kpn_b2));
        const kpn_f2 s = {r[0] - __builtin_bit_cast(float, b << 16), r[1] - __builtin_bit_cast(float, b & 0xffff0000u)};
        ph[j] = a; pm[j] = b; pl[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(s, kpn_b2));
    }
    h = __builtin_bit_cast(kpn_bf16x8, ph); m = __builtin_bit_cast(kpn_bf16x8, pm); l = __builtin_bit_cast(kpn_bf16x8, pl);
}
#else
typedef uint16_t kpn_bf16x8 __attribute__((ext_vector_type(8)));
static inline uint16_t kpn_f2bf(float f) {  // round to nearest even, as v_cvt_pk_bf16_f32
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float kpn_bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static inline void kpn_split3(const float (&x)[8], kpn_bf16x8& h, kpn_bf16x8& m, kpn_bf16x8& l) {
    for (int i = 0; i < 8; ++i) {
        const uint16_t a = kpn_f2bf(x[i]);
        const float r1 = x[i] - kpn_bf2f(a);
        const uint16_t b = kpn_f2bf(r1);
        h[i] = a; m[i] = b; l[i] = kpn_f2bf(r1 - kpn_bf2f(b));
    }
}
static inline kpn_bf16x8 kpn_as_bf16x8(kpn_f32x4 v) { kpn_bf16x8 r; memcpy(&r, &v, 16); return r; }
static inline uint16_t kpn_f2h(float f) { const _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }   // RNE, as v_cvt_pk_f16_f32
static inline float kpn_h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
typedef uint16_t kpn_bf16_t;
static inline kpn_bf16_t kpn_to_bf(float f) { return kpn_f2bf(f); }
static inline float kpn_bf_to_f(kpn_bf16_t b) { return kpn_bf2f(b); }
#define KPN_MFMA16(a, b, c) simt_mfma_f32_32x32x16_bf16((a), (b), (c))
typedef uint32_t kpn_u32x4 __attribute__((ext_vector_type(4)));
template <bool TO_MFMA = true>
static inline void kpn_split_bf16x8(const float (&x)[8], kpn_bf16x8& h, kpn_bf16x8& m, kpn_bf16x8& l) { kpn_split3(x, h, m, l); }
static inline kpn_f32x16 kpn_mfma_f16(kpn_f32x4 a, kpn_u32x4 b, kpn_f32x16 c) {
    kpn_bf16x8 av, bv; memcpy(&av, &a, 16); memcpy(&bv, &b, 16);   // eight 16-bit patterns each
    return simt_mfma_f32_32x32x16_f16(av, bv, c);
}
static inline void kpn_split_f16x8(const float (&x)[8], kpn_u32x4& h, kpn_u32x4& l) {
    for (int j = 0; j < 4; ++j) {
        const uint16_t h0 = kpn_f2h(x[2 * j]), h1 = kpn_f2h(x[2 * j + 1]);
        h[j] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        l[j] = (uint32_t)kpn_f2h(x[2 * j] - kpn_h2f(h0)) | ((uint32_t)kpn_f2h(x[2 * j + 1] - kpn_h2f(h1)) << 16);
    }
}
#endif

#define KPN_NKPT 24
#define KPN_MAXV 16
#define KPN_WAVE 64
#define KPN_TILE 32   // points per wave tile = N dimension of v_mfma_f32_32x32x2_f32

// ---------------------------------------------------------------------------------------------
// Prepared scene (device workspace), all offsets in floats from the workspace base:
//   table : V x KPN_TBL_STRIDE
//   rgbm  : V x H x W x 4      [r,g,b,fg]   (model.py:737,806 sample the mask and the image)
//   geo0  : V x g0h x g0w x 64 (channels last: one tap = one 256-B line)
//   geo1  : V x g1h x g1w x 8
//   tex   : V x th  x tw  x 8
#define KPN_TBL_STRIDE 112
#define KPN_TBL_KRT 0      // 3 rows x [m0 m1 m2 t]
#define KPN_TBL_EXT 12     // 3 rows x [r0 r1 r2 t]
#define KPN_TBL_CPOS 24    // source camera centre
#define KPN_TBL_KCAM 28    // 24 x 3 keypoints in this camera's frame

struct kpn_scene_dev {
    int32_t V, H, W, g0h, g0w, g1h, g1w, th, tw, disable_fg_mask;
    float znear, zfar, nml_scale, two_sigma2;
    uint32_t keep;  // bit v = 0: source view v switched off by train-time view dropout (model.py:742-748); eval: all ones
    const float* table;
    const float* rgbm;
    const float* geo0;
    const float* geo1;
    const float* tex;
    // [0] = max |value| over the source images and the three feature maps (set by kpn_scene_prepare; NaN if one holds a NaN):
    // the kernels that carry operands as two fp16 pieces stand aside when it is beyond fp16's range (kpn_f16_inputs_unsafe)
    const float* flags;
};
// Operands of the two-fp16-piece kernels (k_geo_rows_f2, k_fuse_color_h) must stay within fp16's range.  What is known before a
// pass — the packed weights (the packers count the ones beyond it: kpn_pack_flags_off()) and the maps (kpn_scene_prepare's
// max |value|) — is tested here, on the device, by the kernels themselves: no host round trip.  Activations are not known before
// the pass: a non-finite result is caught behind the kernels (kpn_batch::bad) and the batch evaluated again by the fp32-range
// kernels (run_field, kpn_api.hip).
// products per term set of the two-fp16-piece kernels: 4 = hh hl lh ll, 3 = without ll (<= 2^-24 of the term).
// Rows kernels (geo_rows_pair_kernels.hip): KPN_F16_PRODUCTS.  Per-point kernel (kpn_hlayer, kpn_device.h): KPN_FUSE_F16_PRODUCTS,
// three as well — in the product order given there.
#ifndef KPN_F16_PRODUCTS
#define KPN_F16_PRODUCTS 3
#endif
#ifndef KPN_FUSE_F16_PRODUCTS
#define KPN_FUSE_F16_PRODUCTS 3
#endif
#define KPN_F16_INPUT_LIMIT 60000.0f
#define KPN_SCENE_FLAG_FLOATS 16

// ---------------------------------------------------------------------------------------------
// MFMA weight segments.  One segment = one Linear layer (or a column slice of one) streamed as the
// A operand of v_mfma_f32_32x32x2_f32 in the transposed formulation  out^T = W * in^T :
//   A (32 out-rows x 2 k), lane l holds A[i = l&31][k = l>>5]
//   B (2 k x 32 points),   lane l holds B[k = l>>5][j = l&31]   <- the activations, point j = l&31
//   D (32 x 32),           lane l, reg r holds D[row = (r&3)+8(r>>2)+4(l>>5)][col = l&31]
// so after a layer, lane (p = l&31, h = l>>5) owns output rows rowmap(r,h) of point p, and the next
// layer consumes them straight from registers as its B operand at K-step s = 16*block + r: lanes<32
// supply input feature 32*block+rowmap(r,0), lanes>=32 feature 32*block+rowmap(r,1).  The host packer
// (kpn_pack_weights) permutes the weight columns to match; the K order of a dot product is free.
// Stream layout of a segment: K-steps are taken in groups of G; W part [KS/G groups][64 lanes][G*NOB]
// (a lane's A operands of one group are contiguous: G*NOB/4 dwordx4 loads, a wavefront reads one
// contiguous block per group), then the bias part [NOB][2 halves][16 regs].
enum {
    SEG_G1_0A, SEG_G1_0B, SEG_G1_1, SEG_G1_2, SEG_G1_3,  // geometry MLP layers1 (per point x view)
    SEG_G2_0, SEG_G2_1, SEG_G2_2, SEG_CMP,               // layers2 + ibr_compress_gfeat (per point)
    SEG_RE_0, SEG_RE_1, SEG_BL_0A, SEG_BL_0B, SEG_BL_1,  // IBR head
    SEG_V1_0, SEG_V1_1, SEG_V2_0, SEG_O_0, SEG_O_1,
    SEG_COUNT
};
struct kpn_seg_shape { int ks, nob, g; };
// layers1.0 is split into its keypoint-encoding part (12 groups of 7 K-steps = one keypoint pair per
// group) and its 64-channel feature part; x'-ordered 35-vectors take 20 K-steps (16 + 3 + 1 pad).
#define KPN_SEG_SHAPES                                                                                      \
    {84, 4, 7}, {32, 4, 4}, {64, 4, 4}, {68, 4, 4}, {64, 2, 4}, {64, 2, 4}, {32, 2, 4}, {32, 1, 4}, {64, 1, 4}, \
    {4, 1, 4}, {8, 2, 4}, {40, 2, 4}, {20, 2, 4}, {32, 1, 4}, {16, 1, 4}, {16, 1, 4}, {16, 1, 4}, {20, 1, 4}, {8, 1, 4}
static constexpr kpn_seg_shape kpn_seg_shapes[SEG_COUNT] = {KPN_SEG_SHAPES};

constexpr int kpn_seg_wfloats(int seg) { return kpn_seg_shapes[seg].ks * kpn_seg_shapes[seg].nob * 64; }
constexpr int kpn_seg_bfloats(int seg) { return kpn_seg_shapes[seg].nob * 32; }
constexpr int kpn_seg_woff(int seg) {
    int o = 0;
    for (int i = 0; i < seg; ++i) o += kpn_seg_wfloats(i) + kpn_seg_bfloats(i);
    return o;
}
constexpr int kpn_seg_boff(int seg) { return kpn_seg_woff(seg) + kpn_seg_wfloats(seg); }
// scalars appended after the segments: [0] = |ani_al|, [1..2] = layers2(0) = (sdf_raw, rad) of a point
// that is masked in every view, [3] pad
constexpr int kpn_scalar_off() { return kpn_seg_woff(SEG_COUNT); }
// single-output layers are VALU dot products over a lane's 16 chained features + one cross-half add:
// row vector layout [2 halves][16 regs] weights, then [bias, 0, 0, 0]
enum { ROW_V1_VIS, ROW_V2_1, ROW_O_2, ROW_COUNT };  // vis_layer1.2 row 32, vis_layer2.2, out_layer.4
constexpr int kpn_row_off(int row) { return kpn_scalar_off() + 4 + row * 36; }
constexpr int kpn_fwd_floats() { return kpn_row_off(ROW_COUNT); }
// everything k_fuse_color reads (segments SEG_G2_0.., scalars, row vectors) is one contiguous region
// of 136.8 KB: it is copied into LDS once per (persistent) workgroup
constexpr int kpn_k2_base() { return kpn_seg_woff(SEG_G2_0); }
constexpr int kpn_k2_floats() { return kpn_fwd_floats() - kpn_k2_base(); }

// Backward segments (appended after the forward region): the TRANSPOSED layers1 matrices, streamed exactly
// like forward segments (dX^T = W^T * dY^T is one more chained layer: the lane-register layout of an
// accumulator is the B operand layout of the next MFMA layer in either direction).  No bias part is used.
//   BSEG_G1_3T : dY3(64)  -> dX3(120 of 128)          BSEG_G1_2T : dA2(120 of 128) -> [dX2 chained 128 | hd 8 (block 4)]
//   BSEG_G1_1T : dA1(128) -> dX1(128)                 BSEG_G1_0T : dA0(128) -> d(geometry channels 64)
//   BSEG_G2_1T : dA(h1 pre)(64) -> d softplus(h0)(64)   BSEG_G2_0T : dA(h0 pre)(64) -> d pooled (128 = mean64 | var64)
// colour head (k_color_bwd); "x'" is the forward kernels' [lat24 | rgb3 | tex8] order of the 35-vector:
//   BSEG_CMPT  : d lat(24) -> d pooled(128)                 BSEG_RE_1T : d dir(35, x' K-steps) -> d elu(ray_encoder.0)(16)
//   BSEG_BL_0AT: sum_v dA(base_layer.0)(64) -> [d mean'(35) in blocks 0,1 | d var'(35) in blocks 2,3]
//   BSEG_BL_0BT: dA(base_layer.0)(64) -> d x'(35)           BSEG_BL_1T : dA(base_layer.2)(32) -> d elu(base_layer.0)(64)
//   BSEG_V1_0T, BSEG_V1_1T (rows 0..31), BSEG_V2_0T : 32 -> 32
//   BSEG_O_0T  : dA(out_layer.0)(16) -> [d x(32) | d vis (block 1, row 0)]      BSEG_O_1T : dA(out_layer.2)(8) -> d elu(out_layer.0)(16)
enum { BSEG_G1_3T, BSEG_G1_2T, BSEG_G1_1T, BSEG_G1_0T, BSEG_G2_1T, BSEG_G2_0T,
       BSEG_CMPT, BSEG_RE_1T, BSEG_BL_0AT, BSEG_BL_0BT, BSEG_BL_1T, BSEG_V1_0T, BSEG_V1_1T, BSEG_V2_0T, BSEG_O_0T, BSEG_O_1T,
       BSEG_COUNT };
#define KPN_BSEG_SHAPES {32, 4, 4}, {64, 5, 4}, {64, 4, 4}, {64, 2, 4}, {32, 2, 4}, {32, 4, 4}, \
    {12, 4, 4}, {20, 1, 4}, {32, 4, 4}, {32, 2, 4}, {16, 2, 4}, {16, 1, 4}, {16, 1, 4}, {16, 1, 4}, {8, 2, 4}, {4, 1, 4}
static constexpr kpn_seg_shape kpn_bseg_shapes[BSEG_COUNT] = {KPN_BSEG_SHAPES};
constexpr int kpn_bseg_wfloats(int seg) { return kpn_bseg_shapes[seg].ks * kpn_bseg_shapes[seg].nob * 64; }
constexpr int kpn_bseg_woff(int seg) {
    int o = kpn_fwd_floats();
    for (int i = 0; i < seg; ++i) o += kpn_bseg_wfloats(i);
    return o;
}
// backward row vectors: one forward OUTPUT row of a narrow layer spread over the chained layout of its 64 inputs,
// [2 blocks][2 halves][16 regs]: d in[32b + rowmap(r,h)] = row[(2b+h)*16 + r] * d out   (a rank-1 VALU update)
enum { BROW_G2_2_SDF, BROW_G2_2_RAD, BROW_COUNT };  // layers2.2 rows 0 (sdf_raw) and 1 (rad)
constexpr int kpn_brow_off(int row) { return kpn_bseg_woff(BSEG_COUNT) + row * 64; }
constexpr int kpn_bwd_end() { return kpn_brow_off(BROW_COUNT); }
// Split-bf16 segments of layers1 (k_geo_rows_h), after the backward region.  K runs in steps of 16: the h = 0 lanes of
// the B operand supply 8 values, the h = 1 lanes 8 — for chained inputs a lane's registers 8j..8j+7 of block b (step
// 2b + j), i.e. features 32b + rowmap(8j + e, h).  Stream per step: [ob][piece h,m,l][64 lanes] x 16 B (8 bf16 =
// A[i = lane&31][k = 8(lane>>5) + e]); the fp32 bias blocks of the forward segments are reused.
//   HSEG_G1_0A: step j = keypoints j (h=0) and j+12 (h=1): 7 encoding values + 1 pad;  HSEG_G1_0B: step s = geometry channels 16 s + 8 h + 0..7
//   HSEG_G1_2 : 8 chained steps + 1 step with the 4+4 hd channels (+ pads)
enum { HSEG_G1_0A, HSEG_G1_0B, HSEG_G1_1, HSEG_G1_2, HSEG_G1_3, HSEG_COUNT };
struct kpn_hseg_shape { int ks16, nob; };
#define KPN_HSEG_SHAPES {12, 4}, {4, 4}, {8, 4}, {9, 4}, {8, 2}
static constexpr kpn_hseg_shape kpn_hseg_shapes[HSEG_COUNT] = {KPN_HSEG_SHAPES};
// the same segments exist twice: with three bf16 pieces per value (np = 3, k_geo_rows_h2) and, behind them, with two fp16 pieces
// (np = 2, k_geo_rows_f2: stream per step [ob][piece h,l][64 lanes] x 16 B)
constexpr int kpn_xseg_step_floats(int seg, int np) { return np * kpn_hseg_shapes[seg].nob * 64 * 4; }
constexpr int kpn_hseg_step_floats(int seg) { return kpn_xseg_step_floats(seg, 3); }
constexpr int kpn_hseg_off(int seg) {
    int o = kpn_bwd_end();
    for (int i = 0; i < seg; ++i) o += kpn_hseg_shapes[i].ks16 * kpn_xseg_step_floats(i, 3);
    return o;
}
constexpr int kpn_fseg_off(int seg) {
    int o = kpn_hseg_off(HSEG_COUNT);
    for (int i = 0; i < seg; ++i) o += kpn_hseg_shapes[i].ks16 * kpn_xseg_step_floats(i, 2);
    return o;
}
constexpr int kpn_xseg_off(int seg, int np) { return np == 3 ? kpn_hseg_off(seg) : kpn_fseg_off(seg); }
// ---- the per-point kernel's weights with two fp16 pieces per value (k_fuse_color_h), behind the rows kernels' streams ----
// Segment SEG_x (x >= SEG_G2_0) again, its KS fp32 K-steps taken eight at a time: chunk c = K-steps 8c .. 8c+7 (pads beyond KS),
// i.e. K slot (c, h, e) of v_mfma_f32_32x32x16_f16 carries what K-step 8c + e carried for the half-h lanes — the kernel's
// B operands keep their order, eight per MFMA set instead of one.  Stream per (chunk, output block): [piece h,l][64 lanes] x 16 B
// (consecutive lanes 16 B apart: conflict-free ds_read_b128).  Then a copy of every segment's fp32 bias block, then a copy of
// the scalars and row vectors [kpn_scalar_off(), kpn_fwd_floats()): one contiguous region (141 KB) that the kernel stages in LDS.
constexpr int kpn_cseg_chunks(int seg) { return (kpn_seg_shapes[seg].ks + 7) / 8; }
constexpr int kpn_cseg_wfloats(int seg) { return kpn_cseg_chunks(seg) * kpn_seg_shapes[seg].nob * 2 * 64 * 4; }
constexpr int kpn_cseg_woff(int seg) {
    int o = kpn_fseg_off(HSEG_COUNT);
    for (int i = SEG_G2_0; i < seg; ++i) o += kpn_cseg_wfloats(i);
    return o;
}
constexpr int kpn_k2h_base() { return kpn_cseg_woff(SEG_G2_0); }
constexpr int kpn_cseg_boff(int seg) {
    int o = kpn_cseg_woff(SEG_COUNT);
    for (int i = SEG_G2_0; i < seg; ++i) o += kpn_seg_bfloats(i);
    return o;
}
constexpr int kpn_k2h_tail_off() { return kpn_cseg_boff(SEG_COUNT); }
// layers2's Softplus(beta = 100) in LOG2 UNITS in the fp16 region, like layers1's in the rows kernels (KPN_H2_LOG2ACT below): a layer
// whose OUTPUT is activated is packed times 100 log2(e) (weights and the bias copy), a layer whose INPUT is an activation times
// ln(2) / 100; for layers2.1 the two cancel.  k_fuse_color_h's activation is then max(u, 0) + log2(1 + 2^-|u|): 3 instructions + 2
// transcendentals per value instead of 7 + 2 (the natural-unit form multiplies three times and compares).
// layers2.2's weights times ln(2) / 100 would sit near fp16's subnormal floor (2^-24 absolute against inputs of a few hundred log2
// units): like layers1.3's stream (KPN_F16_ROW_SCALE) it is packed times 2^10 and the kernel multiplies its two outputs by 2^-10.
#define KPN_F16_G22_SCALE 1024.0f
constexpr float kpn_cseg_wfactor(int seg) {
    return seg == SEG_G2_0 ? 144.269504088896341f : (seg == SEG_G2_2 ? 6.93147180559945309e-3f * KPN_F16_G22_SCALE : 1.0f);
}
constexpr float kpn_cseg_bfactor(int seg) { return (seg == SEG_G2_0 || seg == SEG_G2_1) ? 144.269504088896341f : (seg == SEG_G2_2 ? KPN_F16_G22_SCALE : 1.0f); }
constexpr int kpn_k2h_floats() { return kpn_k2h_tail_off() + (kpn_fwd_floats() - kpn_scalar_off()) - kpn_k2h_base(); }
// ---- the backward chains of layers1 with three bf16 pieces per weight (k_geo_rows_bwd), behind the per-point kernel's region ----
// The forward segments SEG_G1_0A .. SEG_G1_2 (recomputed in the backward pass) and the transposed segments BSEG_G1_3T .. BSEG_G1_0T
// (dX = W^T dY), their fp32 K-steps taken CW at a time (CW = 7 for the keypoint segment, whose groups are one keypoint's seven
// values: one zero slot per chunk; 8 otherwise) = one K = 16 chunk of v_mfma_f32_32x32x16_bf16.  bf16, not fp16: gradients span
// too many decades for fp16 pieces.  Stream per (chunk, output block): [piece h,m,l][64 lanes] x 16 B.
enum { BH_G1_0A, BH_G1_0B, BH_G1_1, BH_G1_2, BH_G1_3T, BH_G1_2T, BH_G1_1T, BH_G1_0T, BH_COUNT };
struct kpn_bh_desc { int bwd, seg; };   // bwd: 0 = kpn_seg_* (forward), 1 = kpn_bseg_* (transposed)
static constexpr kpn_bh_desc kpn_bh_descs[BH_COUNT] = {{0, SEG_G1_0A}, {0, SEG_G1_0B}, {0, SEG_G1_1}, {0, SEG_G1_2},
                                                       {1, BSEG_G1_3T}, {1, BSEG_G1_2T}, {1, BSEG_G1_1T}, {1, BSEG_G1_0T}};
constexpr kpn_seg_shape kpn_bh_shape(int i) { return kpn_bh_descs[i].bwd ? kpn_bseg_shapes[kpn_bh_descs[i].seg] : kpn_seg_shapes[kpn_bh_descs[i].seg]; }
constexpr int kpn_bh_src_woff(int i) { return kpn_bh_descs[i].bwd ? kpn_bseg_woff(kpn_bh_descs[i].seg) : kpn_seg_woff(kpn_bh_descs[i].seg); }
constexpr int kpn_bh_cw(int i) { return kpn_bh_shape(i).g == 7 ? 7 : 8; }
constexpr int kpn_bh_chunks(int i) { return (kpn_bh_shape(i).ks + kpn_bh_cw(i) - 1) / kpn_bh_cw(i); }
constexpr int kpn_bh_wfloats(int i) { return kpn_bh_chunks(i) * kpn_bh_shape(i).nob * 3 * 64 * 4; }
constexpr int kpn_bh_off(int i) {
    int o = kpn_k2h_base() + kpn_k2h_floats();
    for (int k = 0; k < i; ++k) o += kpn_bh_wfloats(k);
    return o;
}
// behind everything: [0] = number of fp16-stream weights whose magnitude is beyond fp16's range (as a float; 0 = usable)
#define KPN_PACK_FLAG_FLOATS 4
constexpr int kpn_pack_flags_off() { return kpn_bh_off(BH_COUNT); }
__device__ __forceinline__ bool kpn_f16_inputs_unsafe(const kpn_scene_dev& sc, const float* __restrict__ wp) {
    return wp[kpn_pack_flags_off()] != 0.0f || !(sc.flags[0] <= KPN_F16_INPUT_LIMIT);
}
constexpr int kpn_packed_floats() { return kpn_pack_flags_off() + KPN_PACK_FLAG_FLOATS; }
// The split-bf16 streams carry the Softplus(beta = 100) of layers1 in log2 units (geo_rows_pair_kernels.hip, KPN_H2_LOG2ACT):
// a layer whose OUTPUT goes through the activation is scaled by 100 log2(e) (weights here, biases when the kernel stages
// them), a layer whose INPUT is an activation by ln(2)/100; for layers1.1 and the chained columns of layers1.2 the two cancel.
#ifndef KPN_H2_LOG2ACT
#define KPN_H2_LOG2ACT 1
#endif
#define KPN_H2_ACT_SCALE 144.269504088896341f      // 100 log2(e)
#define KPN_H2_ACT_UNSCALE 6.93147180559945309e-3f // ln(2) / 100
// factor applied to the weight of input column `col` of the layer behind segment `hseg` before it is split into bf16 pieces
constexpr float kpn_hseg_factor(int hseg, int col) {
    if (!KPN_H2_LOG2ACT) return 1.0f;
    if (hseg == HSEG_G1_0A || hseg == HSEG_G1_0B) return KPN_H2_ACT_SCALE;
    if (hseg == HSEG_G1_1) return 1.0f;
    if (hseg == HSEG_G1_2) return col < 128 ? 1.0f : KPN_H2_ACT_SCALE;   // columns 128..135: the sampled hd channels
    return KPN_H2_ACT_UNSCALE;                                           // HSEG_G1_3: no activation behind it
}
// the fp16 streams: layers1.3's weights (~1e-3 after the factor above) would sit on fp16's subnormal floor (2^-24 absolute);
// they are packed times 2^10 and k_geo_rows_f2 multiplies its rows by 2^-10 when it stores them (both exact)
#define KPN_F16_ROW_SCALE 1024.0f
constexpr float kpn_fseg_factor(int hseg, int col) { return kpn_hseg_factor(hseg, col) * (hseg == HSEG_G1_3 ? KPN_F16_ROW_SCALE : 1.0f); }

// Row scratch written by k_geo_rows and read by k_fuse_color: per work item (tile, view) KPN_ROW_SLABS
// slabs of [64 lanes] float4.  Slabs 0..7: the lane's 32 registers of the 64-vector (block b = slab/4);
// slabs 8,9: the per-(point,view) gather record of the colour head — h=0 lanes hold
// [r,g,b, pooling weight | ray_diff(3), dot], h=1 lanes hold the 8 texture channels.
#define KPN_ROW_SLABS 10

// x' order of the colour head's 35-vector: rows 0..23 = lat (orig 11..34), 24..26 = rgb (orig 0..2), 27..34 = tex (orig 3..10)
constexpr int kpn_xprime_to_orig(int q) { return q < 24 ? 11 + q : q - 24; }
// row of the 32x32 D tile held by register r of a lane in half h
#define KPN_ROWMAP(r, h) (((r) & 3) + 8 * ((r) >> 2) + 4 * (h))
