// kpn_api.hip — C ABI (include/kpnerf.h): weight packing, scene preparation, stage ops and the
// hierarchical render pipeline.  Host code only launches kernels on the caller's stream; it never
// synchronises or allocates (except kpn_selftest_mfma, a diagnostic).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kpnerf.h"
#include "kpn_device.h"

// kernels (ray_kernels.hip / field_kernels.hip)
#include "ray_kernels.hip"
#include "field_kernels.hip"
#ifdef KPN_SIMT_EMU
#include "geo_rows_pair_kernels.hip"   // the device build compiles this kernel as its own translation unit (geo_rows_pair_tu.hip)
#else
extern "C" void kpn_internal_launch_row_records(int blocks, void* stream, const kpn_scene_dev* sc, const kpn_points* ps, const float* wp, const int* list,
                                                const int* count, float* xscr, const kpn_batch* batch);
extern "C" void kpn_internal_launch_row_records_live(int blocks, void* stream, const kpn_scene_dev* sc, const kpn_points* ps, const float* wp,
                                                     const int* list, const int* count, const int* tickets, const int* live, float* xscr,
                                                     const kpn_batch* batch);
extern "C" void kpn_internal_launch_geo_rows_pair(int mode, int blocks, void* stream, const kpn_scene_dev* sc, const kpn_points* ps, const float* wp,
                                                  const int* list, const int* count, int* tickets, float* xscr, const kpn_batch* batch);
#endif
#include "field_bwd_kernels.hip"
#include "fuse_bwd_kernels.hip"

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define KPN_REQUIRE(cond, msg) do { if (!(cond)) return fail(KPN_EINVAL, std::string(msg) + " [" #cond "]"); } while (0)
static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(KPN_ELAUNCH, std::string(what) + ": " + hipGetErrorString(e));
    return KPN_OK;
}
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline dim3 grid1d(int64_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

extern "C" int kpn_abi_version(void) { return KPN_ABI_VERSION; }
extern "C" const char* kpn_last_error(void) { return g_err.c_str(); }
extern "C" int kpn_is_device_build(void) {
#ifdef KPN_SIMT_EMU
    return 0;
#else
    return 1;
#endif
}

// ---------------------------------------------------------------------------------------------
// weight packing
namespace {
enum { P_G1_0, P_G1_1, P_G1_2, P_G1_3, P_G2_0, P_G2_1, P_G2_2, P_CMP, P_RE_0, P_RE_1, P_BL_0, P_BL_1, P_V1_0, P_V1_1,
       P_V2_0, P_V2_1, P_O_0, P_O_1, P_O_2, P_COUNT };
const int plain_dims[P_COUNT][2] = {{128, 232}, {128, 128}, {120, 136}, {64, 120}, {64, 128}, {64, 64}, {2, 64},
                                    {24, 128},  {16, 4},    {35, 16},   {64, 105}, {32, 64},  {32, 32}, {33, 32},
                                    {32, 32},   {1, 32},    {16, 37},   {8, 16},   {1, 8}};
struct Plain { const float* w[P_COUNT]; const float* b[P_COUNT]; float ani_al; };
void bind_plain(const float* flat, Plain& pl) {
    const float* p = flat;
    for (int l = 0; l < P_COUNT; ++l) {
        pl.w[l] = p; p += plain_dims[l][0] * plain_dims[l][1];
        pl.b[l] = p; p += plain_dims[l][0];
    }
    pl.ani_al = *p;
}
// chained input: K-step s = 16*block + r is input feature 32*block + rowmap(r, h)
inline int chain_feature(int s, int h) { return 32 * (s / 16) + KPN_ROWMAP(s % 16, h); }
// x' order of the 35-vector: rows 0..23 = lat (orig 11..34), 24..26 = rgb (orig 0..2), 27..34 = tex (orig 3..10)
inline int xprime_to_orig(int q) { return q < 24 ? 11 + q : (q < 27 ? q - 24 : q - 24); }
// x' K-steps (20): s<16 -> row rowmap(s,h); s = 16..18 -> row 32 + rowmap(s-16, h); s = 19 -> pad
inline int xstep_row(int s, int h) { return s < 16 ? KPN_ROWMAP(s, h) : (s < 19 ? 32 + KPN_ROWMAP(s - 16, h) : 9999); }

template <class FMap, class OMap>
void pack_segment(float* packed, int seg, const float* W, const float* b, int out_dim, int in_dim, FMap fmap, OMap omap,
                  bool with_bias = true) {
    const int KS = kpn_seg_shapes[seg].ks, NOB = kpn_seg_shapes[seg].nob, G = kpn_seg_shapes[seg].g;
    const int NF = G * NOB;
    float* w = packed + kpn_seg_woff(seg);
    float* bb = packed + kpn_seg_boff(seg);
    for (int s = 0; s < KS; ++s)
        for (int ob = 0; ob < NOB; ++ob)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5;
                const int orow = omap(ob * 32 + i);
                const int f = fmap(s, h);
                float val = 0.0f;
                if (orow >= 0 && orow < out_dim && f >= 0 && f < in_dim) val = W[(size_t)orow * in_dim + f];
                w[((size_t)(s / G) * 64 + lane) * NF + (s % G) * NOB + ob] = val;
            }
    for (int ob = 0; ob < NOB; ++ob)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) {
                const int orow = omap(ob * 32 + KPN_ROWMAP(r, h));
                bb[(ob * 2 + h) * 16 + r] = (with_bias && orow >= 0 && orow < out_dim) ? b[orow] : 0.0f;
            }
}
// a transposed (backward) segment: out row R of the stream is forward INPUT feature in_of_row(R), K-step (s,h) is
// forward OUTPUT feature chain_feature(s,h); value W[o][f]
template <class RowMap, class KMap>
void pack_segment_t(float* packed, int bseg, const float* W, int out_dim, int in_dim, RowMap in_of_row, KMap out_of_kstep) {
    const int KS = kpn_bseg_shapes[bseg].ks, NOB = kpn_bseg_shapes[bseg].nob, G = kpn_bseg_shapes[bseg].g;
    const int NF = G * NOB;
    float* w = packed + kpn_bseg_woff(bseg);
    for (int s = 0; s < KS; ++s)
        for (int ob = 0; ob < NOB; ++ob)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5;
                const int f = in_of_row(ob * 32 + i);
                const int o = out_of_kstep(s, h);
                float val = 0.0f;
                if (f >= 0 && f < in_dim && o >= 0 && o < out_dim) val = W[(size_t)o * in_dim + f];
                w[((size_t)(s / G) * 64 + lane) * NF + (s % G) * NOB + ob] = val;
            }
}
template <class RowMap>
void pack_segment_t(float* packed, int bseg, const float* W, int out_dim, int in_dim, RowMap in_of_row) {
    pack_segment_t(packed, bseg, W, out_dim, in_dim, in_of_row, [](int s, int h) { return chain_feature(s, h); });
}
// split-bf16 stream of one layer (kpn_common.h HSEG_*): feat(step, h, e) = input feature of the e-th value the half-h
// lanes supply in 16-deep K-step `step`, or -1 (pad)
inline uint16_t host_f2bf(float f) {  // round to nearest even
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float host_bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
// enumerates the elements of one split-bf16 segment: emit(element index within the segment = ((s*NOB+ob)*64+lane)*8+e,
// plain-layout weight index or -1)
template <class FMap, class Emit>
void walk_hsegment(int hseg, size_t w_off, int out_dim, int in_dim, FMap feat, Emit emit) {
    const int KS = kpn_hseg_shapes[hseg].ks16, NOB = kpn_hseg_shapes[hseg].nob;
    for (int s = 0; s < KS; ++s)
        for (int ob = 0; ob < NOB; ++ob)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5, orow = ob * 32 + i;
                for (int e = 0; e < 8; ++e) {
                    const int f = feat(s, h, e);
                    const bool real = orow < out_dim && f >= 0 && f < in_dim;
                    emit((((size_t)s * NOB + ob) * 64 + lane) * 8 + e, real ? (int64_t)(w_off + (size_t)orow * in_dim + f) : (int64_t)-1);
                }
            }
}
// the five layers1 segments with their K maps; w_off[layer] = offset of the layer's W in the plain layout
template <class Emit>
void walk_hsegments(const size_t (&w_off)[4], Emit emit) {
    auto chain16 = [](int s, int h, int e) { return 32 * (s / 2) + KPN_ROWMAP(8 * (s % 2) + e, h); };
    walk_hsegment(HSEG_G1_0A, w_off[0], 128, 232, [](int s, int h, int e) { return e < 7 ? e * 24 + s + 12 * h : -1; },
                  [&](size_t el, int64_t src) { emit(HSEG_G1_0A, el, src); });
    // geo0 channels of step s: 16 s + 8 h + e — the two halves of a point read the same 64-byte piece of one cache line (32
    // distinct lines per gather instruction instead of 64)
    walk_hsegment(HSEG_G1_0B, w_off[0], 128, 232, [](int s, int h, int e) { return 168 + 16 * s + 8 * h + e; },
                  [&](size_t el, int64_t src) { emit(HSEG_G1_0B, el, src); });
    walk_hsegment(HSEG_G1_1, w_off[1], 128, 128, chain16, [&](size_t el, int64_t src) { emit(HSEG_G1_1, el, src); });
    walk_hsegment(HSEG_G1_2, w_off[2], 120, 136,
                  [&](int s, int h, int e) { return s < 8 ? chain16(s, h, e) : (e < 4 ? 128 + 4 * h + e : -1); },
                  [&](size_t el, int64_t src) { emit(HSEG_G1_2, el, src); });
    walk_hsegment(HSEG_G1_3, w_off[3], 64, 120, chain16, [&](size_t el, int64_t src) { emit(HSEG_G1_3, el, src); });
}
// u16 slot of piece pc of element el of a segment, relative to the packed buffer viewed as uint16; np = pieces per value
// (3: the bf16 streams, 2: the fp16 streams behind them)
inline size_t hseg_slot(int hseg, size_t el, int pc, int np = 3) {
    const int NOB = kpn_hseg_shapes[hseg].nob;
    const size_t e = el % 8, lane = (el / 8) % 64, ob = (el / 512) % NOB, s = el / (512 * (size_t)NOB);
    return (size_t)kpn_xseg_off(hseg, np) * 2 + ((((s * NOB + ob) * np + pc) * 64 + lane) * 8 + e);   // [step][block][piece][lane][8]
}
inline uint16_t host_f2h(float f) { const _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }   // round to nearest even
inline float host_h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
// factor folded into the weight at plain index `src` of segment `hseg` (log2-unit activations, kpn_common.h kpn_hseg_factor)
inline float hseg_weight_factor(int hseg, int64_t src, const size_t (&w_off)[4], int np = 3) {
    static const int layer_of[HSEG_COUNT] = {0, 0, 1, 2, 3}, in_dim[4] = {232, 128, 136, 120};
    if (src < 0) return 1.0f;
    const int l = layer_of[hseg];
    const int col = (int)((src - (int64_t)w_off[l]) % in_dim[l]);
    return np == 3 ? kpn_hseg_factor(hseg, col) : kpn_fseg_factor(hseg, col);
}
float softplus100_host(float x) { float t = x * 100.0f; return t > 20.0f ? x : log1pf(expf(t)) / 100.0f; }
// ---- the fp16 region of the per-point kernel (kpn_common.h kpn_cseg_*): every value is read from the ALREADY PACKED fp32
// stream of the same segment — K slot (chunk c, e) of the half-h lanes = fp32 K-step 8c + e — so host and device packer need
// no index maps of their own.  Element t of the region's stream part: ((c*NOB + ob)*64 + lane)*8 + e within its segment.
struct K2hSeg { int seg, first; };   // first element index of the segment in the concatenated element space
inline int k2h_elements(int seg) { return kpn_cseg_chunks(seg) * kpn_seg_shapes[seg].nob * 64 * 8; }
inline int k2h_total_elements() { int n = 0; for (int sg = SEG_G2_0; sg < SEG_COUNT; ++sg) n += k2h_elements(sg); return n; }
// (segment, element) -> float index of the fp32 packed weight (or -1: pad), u16 slot of the h piece; the l piece is 64*8 slots on
__host__ __device__ inline void k2h_locate(int seg, int el, int& src, int& slot) {
    const int NOB = kpn_seg_shapes[seg].nob, G = kpn_seg_shapes[seg].g, KS = kpn_seg_shapes[seg].ks;
    const int e = el % 8, lane = (el / 8) % 64, ob = (el / 512) % NOB, c = el / (512 * NOB);
    const int s = 8 * c + e;
    src = s < KS ? kpn_seg_woff(seg) + ((s / G) * 64 + lane) * (G * NOB) + (s % G) * NOB + ob : -1;
    slot = kpn_cseg_woff(seg) * 2 + (((c * NOB + ob) * 2) * 64 + lane) * 8 + e;
}
// ---- the backward chains' bf16 region (kpn_common.h BH_*): same idea, three bf16 pieces, chunk width 7 or 8 ----
inline int bh_elements(int i) { return kpn_bh_chunks(i) * kpn_bh_shape(i).nob * 64 * 8; }
inline int bh_total_elements() { int n = 0; for (int i = 0; i < BH_COUNT; ++i) n += bh_elements(i); return n; }
void pack_bh_host(float* P) {
    uint16_t* P16 = reinterpret_cast<uint16_t*>(P);
    for (int i = 0; i < BH_COUNT; ++i) {
        const int NOB = kpn_bh_shape(i).nob, G = kpn_bh_shape(i).g, KS = kpn_bh_shape(i).ks, CW = kpn_bh_cw(i);
        for (int el = 0; el < bh_elements(i); ++el) {
            const int e = el % 8, lane = (el / 8) % 64, ob = (el / 512) % NOB, c = el / (512 * NOB);
            const int s = c * CW + e;
            const float w = (e < CW && s < KS) ? P[kpn_bh_src_woff(i) + ((s / G) * 64 + lane) * (G * NOB) + (s % G) * NOB + ob] : 0.0f;
            const size_t slot = (size_t)kpn_bh_off(i) * 2 + (((size_t)(c * NOB + ob) * 3) * 64 + lane) * 8 + e;
            const uint16_t ph = host_f2bf(w);
            const float r1 = w - host_bf2f(ph);
            const uint16_t pm = host_f2bf(r1);
            P16[slot] = ph; P16[slot + 512] = pm; P16[slot + 1024] = host_f2bf(r1 - host_bf2f(pm));
        }
    }
}
int pack_k2h_host(float* P) {   // returns the number of weights beyond fp16's range
    uint16_t* P16 = reinterpret_cast<uint16_t*>(P);
    int beyond = 0;
    for (int sg = SEG_G2_0; sg < SEG_COUNT; ++sg) {
        for (int el = 0; el < k2h_elements(sg); ++el) {
            int src, slot;
            k2h_locate(sg, el, src, slot);
            const float w = (src >= 0 ? P[src] : 0.0f) * kpn_cseg_wfactor(sg);   // log2-unit activations of layers2 (kpn_common.h)
            if (!(fabsf(w) <= 65504.0f)) ++beyond;
            const uint16_t ph = host_f2h(w);
            P16[slot] = ph; P16[slot + 512] = host_f2h(w - host_h2f(ph));
        }
        for (int k = 0; k < kpn_seg_bfloats(sg); ++k) P[kpn_cseg_boff(sg) + k] = P[kpn_seg_boff(sg) + k] * kpn_cseg_bfactor(sg);
    }
    return beyond;
}
}  // namespace
// device side of the same: one thread per stream element, then the bias blocks and the scalar / row-vector tail
__global__ void k_pack_k2h(float* __restrict__ packed, int n_elem, float* __restrict__ flags) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_elem) {
        int sg = SEG_G2_0, el = t;
        for (;;) {
            const int n = kpn_cseg_chunks(sg) * kpn_seg_shapes[sg].nob * 64 * 8;
            if (el < n) break;
            el -= n; ++sg;
        }
        const int NOB = kpn_seg_shapes[sg].nob, G = kpn_seg_shapes[sg].g, KS = kpn_seg_shapes[sg].ks;
        const int e = el % 8, lane = (el / 8) % 64, ob = (el / 512) % NOB, c = el / (512 * NOB);
        const int s = 8 * c + e;
        const float w = (s < KS ? packed[kpn_seg_woff(sg) + ((s / G) * 64 + lane) * (G * NOB) + (s % G) * NOB + ob] : 0.0f) * kpn_cseg_wfactor(sg);
        const int slot = kpn_cseg_woff(sg) * 2 + (((c * NOB + ob) * 2) * 64 + lane) * 8 + e;
#ifndef KPN_SIMT_EMU
        const _Float16 h = (_Float16)w;
        const _Float16 l = (_Float16)(w - (float)h);
        uint16_t ph, pl; memcpy(&ph, &h, 2); memcpy(&pl, &l, 2);
#else
        const uint16_t ph = kpn_f2h(w), pl = kpn_f2h(w - kpn_h2f(ph));
#endif
        if (!(fabsf(w) <= 65504.0f)) kpn_atomic_add(flags, 1.0f);
        uint16_t* p16 = reinterpret_cast<uint16_t*>(packed);
        p16[slot] = ph; p16[slot + 512] = pl;
    }
    // bias blocks + tail: plain copies inside the packed buffer
    const int n_bias = kpn_k2h_tail_off() - kpn_cseg_boff(SEG_G2_0), n_tail = kpn_fwd_floats() - kpn_scalar_off();
    if (t < n_bias) {
        int sg = SEG_G2_0, k = t;
        while (k >= kpn_seg_bfloats(sg)) { k -= kpn_seg_bfloats(sg); ++sg; }
        packed[kpn_cseg_boff(sg) + k] = packed[kpn_seg_boff(sg) + k] * kpn_cseg_bfactor(sg);
    } else if (t < n_bias + n_tail) {
        packed[kpn_k2h_tail_off() + (t - n_bias)] = packed[kpn_scalar_off() + (t - n_bias)];
    }
}


extern "C" size_t kpn_plain_weight_floats(void) {
    size_t n = 1;
    for (int l = 0; l < P_COUNT; ++l) n += (size_t)plain_dims[l][0] * plain_dims[l][1] + plain_dims[l][0];
    return n;
}
extern "C" size_t kpn_packed_weight_floats(void) { return (size_t)kpn_packed_floats(); }

extern "C" int kpn_pack_weights(const float* plain_host, float* packed_host) {
    KPN_REQUIRE(plain_host && packed_host, "null pointer");
    Plain pl;
    bind_plain(plain_host, pl);
    auto ident = [](int row) { return row; };
    auto chain = [](int s, int h) { return chain_feature(s, h); };
    float* P = packed_host;
    // layers1.0: part A, K-steps 0..83 keypoint encoding (j = s/7 -> keypoint j+12h, t = s%7 -> PE block t:
    // feature t*24 + kp, spatial.py:36-39,117); part B, 32 K-steps: geometry channel 32h + s (feature 168 + c)
    pack_segment(P, SEG_G1_0A, pl.w[P_G1_0], pl.b[P_G1_0], 128, 232,
                 [](int s, int h) { return (s % 7) * 24 + (s / 7) + 12 * h; }, ident);
    pack_segment(P, SEG_G1_0B, pl.w[P_G1_0], pl.b[P_G1_0], 128, 232, [](int s, int h) { return 168 + 32 * h + s; }, ident,
                 /*with_bias=*/false);
    pack_segment(P, SEG_G1_1, pl.w[P_G1_1], pl.b[P_G1_1], 128, 128, chain, ident);
    // layers1.2: [128 chained | hd channel 4h + (s-64)]
    pack_segment(P, SEG_G1_2, pl.w[P_G1_2], pl.b[P_G1_2], 120, 136,
                 [](int s, int h) { return s < 64 ? chain_feature(s, h) : 128 + 4 * h + (s - 64); }, ident);
    pack_segment(P, SEG_G1_3, pl.w[P_G1_3], pl.b[P_G1_3], 64, 120, chain, ident);
    // layers2.0: [mean64 | var64], each in chained order
    pack_segment(P, SEG_G2_0, pl.w[P_G2_0], pl.b[P_G2_0], 64, 128,
                 [](int s, int h) { return s < 32 ? chain_feature(s, h) : 64 + chain_feature(s - 32, h); }, ident);
    pack_segment(P, SEG_G2_1, pl.w[P_G2_1], pl.b[P_G2_1], 64, 64, chain, ident);
    pack_segment(P, SEG_G2_2, pl.w[P_G2_2], pl.b[P_G2_2], 2, 64, chain, ident);
    // ibr_compress_gfeat: same input as layers2.0; output rows already in x' order (row q<24 = lat q)
    pack_segment(P, SEG_CMP, pl.w[P_CMP], pl.b[P_CMP], 24, 128,
                 [](int s, int h) { return s < 32 ? chain_feature(s, h) : 64 + chain_feature(s - 32, h); }, ident);
    pack_segment(P, SEG_RE_0, pl.w[P_RE_0], pl.b[P_RE_0], 16, 4, [](int s, int h) { return s < 2 ? 2 * s + h : -1; }, ident);
    // ray_encoder.2: output rows permuted to x' order
    pack_segment(P, SEG_RE_1, pl.w[P_RE_1], pl.b[P_RE_1], 35, 16, chain,
                 [](int q) { return q < 35 ? xprime_to_orig(q) : -1; });
    // base_layer.0 columns: [mean35 | var35 | x35] (model.py:1292)
    pack_segment(P, SEG_BL_0A, pl.w[P_BL_0], pl.b[P_BL_0], 64, 105,
                 [](int s, int h) {
                     const int q = xstep_row(s % 20, h);
                     return q < 35 ? (s / 20) * 35 + xprime_to_orig(q) : -1;
                 }, ident);
    pack_segment(P, SEG_BL_0B, pl.w[P_BL_0], pl.b[P_BL_0], 64, 105,
                 [](int s, int h) { const int q = xstep_row(s, h); return q < 35 ? 70 + xprime_to_orig(q) : -1; }, ident,
                 /*with_bias=*/false);
    pack_segment(P, SEG_BL_1, pl.w[P_BL_1], pl.b[P_BL_1], 32, 64, chain, ident);
    pack_segment(P, SEG_V1_0, pl.w[P_V1_0], pl.b[P_V1_0], 32, 32, chain, ident);
    pack_segment(P, SEG_V1_1, pl.w[P_V1_1], pl.b[P_V1_1], 32, 32, chain, ident);  // rows 0..31 (res); row 32 (vis) below
    pack_segment(P, SEG_V2_0, pl.w[P_V2_0], pl.b[P_V2_0], 32, 32, chain, ident);
    // out_layer.0 columns: [x32 | vis | ray_diff4] (model.py:1300); extra K-steps 16,17,18
    pack_segment(P, SEG_O_0, pl.w[P_O_0], pl.b[P_O_0], 16, 37,
                 [](int s, int h) {
                     if (s < 16) return chain_feature(s, h);
                     const int f = 32 + 2 * (s - 16) + h;
                     return (s < 19 && f < 37) ? f : -1;
                 }, ident);
    pack_segment(P, SEG_O_1, pl.w[P_O_1], pl.b[P_O_1], 8, 16, chain, ident);
    // single-output layers as row vectors over the chained features of one 32-row block
    auto pack_row = [&](int row, const float* W, int in_dim, float bias) {
        float* r = P + kpn_row_off(row);
        for (int h = 0; h < 2; ++h)
            for (int k = 0; k < 16; ++k) { const int f = KPN_ROWMAP(k, h); r[h * 16 + k] = f < in_dim ? W[f] : 0.0f; }
        r[32] = bias; r[33] = r[34] = r[35] = 0.0f;
    };
    pack_row(ROW_V1_VIS, pl.w[P_V1_1] + 32 * 32, 32, pl.b[P_V1_1][32]);
    pack_row(ROW_V2_1, pl.w[P_V2_1], 32, pl.b[P_V2_1][0]);
    pack_row(ROW_O_2, pl.w[P_O_2], 8, pl.b[P_O_2][0]);
    // backward of layers1 (kpn_geo_rows_backward): the transposed matrices
    pack_segment_t(P, BSEG_G1_3T, pl.w[P_G1_3], 64, 120, ident);
    pack_segment_t(P, BSEG_G1_2T, pl.w[P_G1_2], 120, 136, [](int R) { return R < 128 ? R : (R < 136 ? R : -1); });
    pack_segment_t(P, BSEG_G1_1T, pl.w[P_G1_1], 128, 128, ident);
    pack_segment_t(P, BSEG_G1_0T, pl.w[P_G1_0], 128, 232, [](int R) { return R < 64 ? 168 + R : -1; });
    // backward of layers2 (kpn_query_backward)
    pack_segment_t(P, BSEG_G2_1T, pl.w[P_G2_1], 64, 64, ident);
    pack_segment_t(P, BSEG_G2_0T, pl.w[P_G2_0], 64, 128, ident);
    // backward of the colour head (k_color_bwd)
    pack_segment_t(P, BSEG_CMPT, pl.w[P_CMP], 24, 128, ident);
    pack_segment_t(P, BSEG_RE_1T, pl.w[P_RE_1], 35, 16, [](int R) { return R < 16 ? R : -1; },
                   [](int s, int h) { const int q = xstep_row(s, h); return q < 35 ? xprime_to_orig(q) : -1; });
    pack_segment_t(P, BSEG_BL_0AT, pl.w[P_BL_0], 64, 105, [](int R) {
        const int part = R / 64, q = R % 64;
        return q < 35 ? part * 35 + xprime_to_orig(q) : -1;
    });
    pack_segment_t(P, BSEG_BL_0BT, pl.w[P_BL_0], 64, 105, [](int R) { return R < 35 ? 70 + xprime_to_orig(R) : -1; });
    pack_segment_t(P, BSEG_BL_1T, pl.w[P_BL_1], 32, 64, ident);
    pack_segment_t(P, BSEG_V1_0T, pl.w[P_V1_0], 32, 32, ident);
    pack_segment_t(P, BSEG_V1_1T, pl.w[P_V1_1], 32, 32, ident);  // the vis row (32) is a rank-1 VALU update (ROW_V1_VIS)
    pack_segment_t(P, BSEG_V2_0T, pl.w[P_V2_0], 32, 32, ident);
    pack_segment_t(P, BSEG_O_0T, pl.w[P_O_0], 16, 37, [](int R) { return R <= 32 ? R : -1; });
    pack_segment_t(P, BSEG_O_1T, pl.w[P_O_1], 8, 16, [](int R) { return R < 16 ? R : -1; });
    for (int o = 0; o < 2; ++o)
        for (int b = 0; b < 2; ++b)
            for (int h = 0; h < 2; ++h)
                for (int r = 0; r < 16; ++r)
                    P[kpn_brow_off(BROW_G2_2_SDF + o) + (2 * b + h) * 16 + r] = pl.w[P_G2_2][o * 64 + 32 * b + KPN_ROWMAP(r, h)];
    // split-bf16 streams of layers1 (k_geo_rows_h)
    {
        size_t w_off[4];
        for (int l = 0; l < 4; ++l) w_off[l] = (size_t)(pl.w[P_G1_0 + l] - plain_host);
        uint16_t* P16 = reinterpret_cast<uint16_t*>(P);
        walk_hsegments(w_off, [&](int hseg, size_t el, int64_t src) {
            const float w = src >= 0 ? plain_host[src] * hseg_weight_factor(hseg, src, w_off) : 0.0f;
            const uint16_t ph = host_f2bf(w);
            const float r1 = w - host_bf2f(ph);
            const uint16_t pm = host_f2bf(r1);
            P16[hseg_slot(hseg, el, 0)] = ph; P16[hseg_slot(hseg, el, 1)] = pm; P16[hseg_slot(hseg, el, 2)] = host_f2bf(r1 - host_bf2f(pm));
        });
        // fp16 double-split streams of layers1 (k_geo_rows_f2) and the count of weights beyond fp16's range
        int beyond = 0;
        walk_hsegments(w_off, [&](int hseg, size_t el, int64_t src) {
            const float w = src >= 0 ? plain_host[src] * hseg_weight_factor(hseg, src, w_off, 2) : 0.0f;
            if (!(fabsf(w) <= 65504.0f)) ++beyond;
            const uint16_t ph = host_f2h(w);
            P16[hseg_slot(hseg, el, 0, 2)] = ph; P16[hseg_slot(hseg, el, 1, 2)] = host_f2h(w - host_h2f(ph));
        });
        // the per-point kernel's region with two fp16 pieces per value (k_fuse_color_h): derived from the fp32 streams packed above
        beyond += pack_k2h_host(P);
        pack_bh_host(P);
        float* fl = P + kpn_pack_flags_off();
        fl[0] = (float)beyond; fl[1] = fl[2] = fl[3] = 0.0f;
    }
    // scalars: |ani_al| (model.py:1287) and layers2(0), the query() result of a fully masked point
    float* sc = P + kpn_scalar_off();
    sc[0] = fabsf(pl.ani_al);
    {
        float a[64], b2[64], o2[2];
        for (int o = 0; o < 64; ++o) a[o] = softplus100_host(pl.b[P_G2_0][o]);
        for (int o = 0; o < 64; ++o) {
            float acc = 0.0f;
            for (int i = 0; i < 64; ++i) acc += pl.w[P_G2_1][o * 64 + i] * a[i];
            b2[o] = softplus100_host(acc + pl.b[P_G2_1][o]);
        }
        for (int o = 0; o < 2; ++o) {
            float acc = 0.0f;
            for (int i = 0; i < 64; ++i) acc += pl.w[P_G2_2][o * 64 + i] * b2[i];
            o2[o] = acc + pl.b[P_G2_2][o];
        }
        sc[1] = o2[0]; sc[2] = o2[1];
        sc[3] = pl.ani_al > 0.0f ? 1.0f : (pl.ani_al < 0.0f ? -1.0f : 0.0f);  // d|a|/da for the colour-head reverse
    }
    // k_fuse_color_h's copy of the scalars and row vectors (the tail of its LDS region)
    memcpy(P + kpn_k2h_tail_off(), P + kpn_scalar_off(), sizeof(float) * (size_t)(kpn_fwd_floats() - kpn_scalar_off()));
    return KPN_OK;
}

// ---------------------------------------------------------------------------------------------
// device-side packing: the host packer above is a pure gather apart from four derived scalars, so its index map is
// taken once (by packing a ramp) and applied on the device — a training loop re-packs after every optimizer step
// split-bf16 region: element t of the concatenated segments -> three bf16 pieces at their slots
__global__ void k_pack_hseg(const float* __restrict__ plain, const int32_t* __restrict__ src, const int32_t* __restrict__ slot0,
                            const int32_t* __restrict__ pstride, const float* __restrict__ factor, int n,
                            uint16_t* __restrict__ packed16) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float w = src[t] >= 0 ? kpn_mul_nofma(plain[src[t]], factor[t]) : 0.0f;   // the host packer's product, bit for bit
    float one[8] = {w, 0, 0, 0, 0, 0, 0, 0};
    kpn_bf16x8 h, m, l;
    kpn_split3(one, h, m, l);
    uint16_t ph, pm, plo;
    { const auto hv = h[0]; const auto mv = m[0]; const auto lv = l[0]; memcpy(&ph, &hv, 2); memcpy(&pm, &mv, 2); memcpy(&plo, &lv, 2); }
    packed16[slot0[t]] = ph; packed16[slot0[t] + pstride[t]] = pm; packed16[slot0[t] + 2 * pstride[t]] = plo;
}
// fp16 region: two pieces; flag[0] counts the weights beyond fp16's range
__global__ void k_pack_fseg(const float* __restrict__ plain, const int32_t* __restrict__ src, const int32_t* __restrict__ slot0,
                            const int32_t* __restrict__ pstride, const float* __restrict__ factor, int n,
                            uint16_t* __restrict__ packed16, float* __restrict__ flags) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float w = src[t] >= 0 ? kpn_mul_nofma(plain[src[t]], factor[t]) : 0.0f;
#ifndef KPN_SIMT_EMU
    const _Float16 h = (_Float16)w;
    const _Float16 l = (_Float16)(w - (float)h);
    uint16_t ph, pl; memcpy(&ph, &h, 2); memcpy(&pl, &l, 2);
#else
    const uint16_t ph = kpn_f2h(w), pl = kpn_f2h(w - kpn_h2f(ph));
#endif
    if (!(fabsf(w) <= 65504.0f)) kpn_atomic_add(flags, 1.0f);
    packed16[slot0[t]] = ph; packed16[slot0[t] + pstride[t]] = pl;
}
__global__ void k_pack_gather(const float* __restrict__ plain, const int32_t* __restrict__ map, int n, float* __restrict__ packed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int m = map[i];
    packed[i] = m >= 0 ? plain[m] : 0.0f;
}
// scalars: |ani_al|, layers2(0) = query()'s [sdf_raw, rad] of a point masked in every view, sign(ani_al)
__global__ void k_pack_scalars(const float* __restrict__ plain, size_t w0, size_t b0, size_t w1, size_t b1, size_t w2, size_t b2,
                               size_t ani, float* __restrict__ sc) {
    __shared__ float a[64], b[64];
    const int o = threadIdx.x;  // 64 threads
    auto sp = [](float x) { const float t = x * 100.0f; return t > 20.0f ? x : log1pf(expf(t)) / 100.0f; };
    a[o] = sp(plain[b0 + o]);   // layers2.0 on pooled = 0
    __syncthreads();
    float acc = 0.0f;
    for (int i = 0; i < 64; ++i) acc += plain[w1 + (size_t)o * 64 + i] * a[i];
    b[o] = sp(acc + plain[b1 + o]);
    __syncthreads();
    if (o < 2) {
        float s = 0.0f;
        for (int i = 0; i < 64; ++i) s += plain[w2 + (size_t)o * 64 + i] * b[i];
        sc[1 + o] = s + plain[b2 + o];
    }
    if (o == 2) {
        const float al = plain[ani];
        sc[0] = fabsf(al);
        sc[3] = al > 0.0f ? 1.0f : (al < 0.0f ? -1.0f : 0.0f);
    }
    (void)w0;
}

// The gather maps of the device packer live in device memory, so they are kept PER DEVICE (a process that renders on two
// GPUs packs on both); built on first use for the device that is current at the call.
__global__ void k_pack_bh(float* __restrict__ packed, int n_elem) {
    int el = blockIdx.x * blockDim.x + threadIdx.x;
    if (el >= n_elem) return;
    int i = 0;
    for (;;) {
        const int n = kpn_bh_chunks(i) * kpn_bh_shape(i).nob * 64 * 8;
        if (el < n) break;
        el -= n; ++i;
    }
    const int NOB = kpn_bh_shape(i).nob, G = kpn_bh_shape(i).g, KS = kpn_bh_shape(i).ks, CW = kpn_bh_cw(i);
    const int e = el % 8, lane = (el / 8) % 64, ob = (el / 512) % NOB, c = el / (512 * NOB);
    const int s = c * CW + e;
    const float w = (e < CW && s < KS) ? packed[kpn_bh_src_woff(i) + ((s / G) * 64 + lane) * (G * NOB) + (s % G) * NOB + ob] : 0.0f;
    float one[8] = {w, 0, 0, 0, 0, 0, 0, 0};
    kpn_bf16x8 h, m, l;
    kpn_split3(one, h, m, l);
    uint16_t ph, pm, plo;
    { const auto hv = h[0]; const auto mv = m[0]; const auto lv = l[0]; memcpy(&ph, &hv, 2); memcpy(&pm, &mv, 2); memcpy(&plo, &lv, 2); }
    uint16_t* p16 = reinterpret_cast<uint16_t*>(packed);
    const size_t slot = (size_t)kpn_bh_off(i) * 2 + (((size_t)(c * NOB + ob) * 3) * 64 + lane) * 8 + e;
    p16[slot] = ph; p16[slot + 512] = pm; p16[slot + 1024] = plo;
}
namespace {
struct DevicePackMaps {
    int32_t* map = nullptr;
    int32_t *hsrc = nullptr, *hslot = nullptr, *hstride = nullptr;
    float* hfactor = nullptr;
    int32_t *fslot = nullptr, *fstride = nullptr;   // the fp16 streams: same sources, their own slots and factors
    float* ffactor = nullptr;
    int n_helem = 0;
    int rc = KPN_OK;
};
DevicePackMaps* device_pack_maps() {
    static std::mutex mtx;
    static std::vector<DevicePackMaps*> per_device;   // index = HIP device ordinal
    int dev = 0;
#ifndef KPN_SIMT_EMU
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
#endif
    std::lock_guard<std::mutex> lock(mtx);
    if ((size_t)dev >= per_device.size()) per_device.resize((size_t)dev + 1, nullptr);
    if (per_device[dev]) return per_device[dev];
    DevicePackMaps* M = per_device[dev] = new DevicePackMaps();
    const size_t np = kpn_plain_weight_floats(), nk = kpn_packed_weight_floats();
    std::vector<float> ramp(np), pk(nk, 0.0f);
    for (size_t i = 0; i < np; ++i) ramp[i] = (float)(i + 1);  // exact in fp32 (np < 2^24)
    if (kpn_pack_weights(ramp.data(), pk.data()) != KPN_OK) { M->rc = KPN_EINVAL; return M; }
    std::vector<int32_t> map(nk);
    // (the four derived scalars are not gathers: k_pack_scalars writes them; the split-bf16 region has its own maps)
    for (size_t i = 0; i < nk; ++i) map[i] = (pk[i] >= 1.0f && pk[i] <= (float)np) ? (int32_t)pk[i] - 1 : -1;
    for (int i = 0; i < 4; ++i) map[kpn_scalar_off() + i] = -1;
    auto up = [&](auto** d, const auto& h) {
        if (hipMalloc((void**)d, h.size() * sizeof(h[0])) != hipSuccess ||
            hipMemcpy(*d, h.data(), h.size() * sizeof(h[0]), hipMemcpyHostToDevice) != hipSuccess) M->rc = KPN_ELAUNCH;
    };
    up(&M->map, map);
    // split-bf16 region: per element its source weight, the folded factor, the u16 slot of its first piece and the piece stride
    std::vector<int32_t> hsrc, hslot, hstride;
    std::vector<float> hfactor;
    size_t w_off[4];
    for (int l = 0; l < 4; ++l) { size_t o = 0; for (int k = 0; k < P_G1_0 + l; ++k) o += (size_t)plain_dims[k][0] * plain_dims[k][1] + plain_dims[k][0]; w_off[l] = o; }
    walk_hsegments(w_off, [&](int hseg, size_t el, int64_t src) {
        hsrc.push_back((int32_t)src);
        hfactor.push_back(hseg_weight_factor(hseg, src, w_off));
        hslot.push_back((int32_t)hseg_slot(hseg, el, 0));
        hstride.push_back((int32_t)(hseg_slot(hseg, el, 1) - hseg_slot(hseg, el, 0)));
    });
    M->n_helem = (int)hsrc.size();
    up(&M->hsrc, hsrc); up(&M->hslot, hslot); up(&M->hstride, hstride); up(&M->hfactor, hfactor);
    std::vector<int32_t> fslot, fstride;
    std::vector<float> ffactor;
    walk_hsegments(w_off, [&](int hseg, size_t el, int64_t src) {
        ffactor.push_back(hseg_weight_factor(hseg, src, w_off, 2));
        fslot.push_back((int32_t)hseg_slot(hseg, el, 0, 2));
        fstride.push_back((int32_t)(hseg_slot(hseg, el, 1, 2) - hseg_slot(hseg, el, 0, 2)));
    });
    up(&M->fslot, fslot); up(&M->fstride, fstride); up(&M->ffactor, ffactor);
    return M;
}
}  // namespace

extern "C" int kpn_pack_weights_device(const float* plain_dev, float* packed_dev, void* stream) {
    KPN_REQUIRE(plain_dev && packed_dev, "null pointer");
    const size_t np = kpn_plain_weight_floats();
    const DevicePackMaps* M = device_pack_maps();
    if (!M || M->rc != KPN_OK || !M->map) return fail(KPN_ELAUNCH, "could not build the device pack map");
    const int n_gather = kpn_bwd_end();  // everything before the split-bf16 region is a gather
    KPN_LAUNCH(k_pack_gather, grid1d((int64_t)n_gather, 256), dim3(256), stream, plain_dev, (const int32_t*)M->map, n_gather, packed_dev);
    KPN_LAUNCH(k_pack_hseg, grid1d((int64_t)M->n_helem, 256), dim3(256), stream, plain_dev, (const int32_t*)M->hsrc,
               (const int32_t*)M->hslot, (const int32_t*)M->hstride, (const float*)M->hfactor, M->n_helem,
               reinterpret_cast<uint16_t*>(packed_dev));
    (void)hipMemsetAsync(packed_dev + kpn_pack_flags_off(), 0, KPN_PACK_FLAG_FLOATS * sizeof(float), (hipStream_t)stream);
    KPN_LAUNCH(k_pack_fseg, grid1d((int64_t)M->n_helem, 256), dim3(256), stream, plain_dev, (const int32_t*)M->hsrc,
               (const int32_t*)M->fslot, (const int32_t*)M->fstride, (const float*)M->ffactor, M->n_helem,
               reinterpret_cast<uint16_t*>(packed_dev), packed_dev + kpn_pack_flags_off());
    auto woff = [](int layer) { size_t o = 0; for (int l = 0; l < layer; ++l) o += (size_t)plain_dims[l][0] * plain_dims[l][1] + plain_dims[l][0]; return o; };
    auto boff = [&](int layer) { return woff(layer) + (size_t)plain_dims[layer][0] * plain_dims[layer][1]; };
    KPN_LAUNCH(k_pack_scalars, dim3(1), dim3(64), stream, plain_dev, woff(P_G2_0), boff(P_G2_0), woff(P_G2_1), boff(P_G2_1),
               woff(P_G2_2), boff(P_G2_2), np - 1, packed_dev + kpn_scalar_off());
    // the per-point kernel's fp16 region from the fp32 streams, biases, scalars and row vectors written above (same stream: ordered)
    const int n_k2h = k2h_total_elements();
    KPN_LAUNCH(k_pack_k2h, grid1d((int64_t)n_k2h, 256), dim3(256), stream, packed_dev, n_k2h, packed_dev + kpn_pack_flags_off());
    const int n_bh = bh_total_elements();
    KPN_LAUNCH(k_pack_bh, grid1d((int64_t)n_bh, 256), dim3(256), stream, packed_dev, n_bh);
    return check_launch("kpn_pack_weights_device");
}

// ---------------------------------------------------------------------------------------------
// scene
namespace {
struct SceneLayout { size_t table, rgbm, geo0, geo1, tex, flags, total; };  // float offsets
SceneLayout scene_layout(const kpn_scene_desc* d) {
    SceneLayout L;
    size_t o = 0;
    L.table = o; o += align_up((size_t)d->n_views * KPN_TBL_STRIDE, 64);
    L.rgbm = o; o += align_up((size_t)d->n_views * d->src_h * d->src_w * 4, 64);
    L.geo0 = o; o += align_up((size_t)d->n_views * d->geo0_h * d->geo0_w * 64, 64);
    L.geo1 = o; o += align_up((size_t)d->n_views * d->geo1_h * d->geo1_w * 8, 64);
    L.tex = o; o += align_up((size_t)d->n_views * d->tex_h * d->tex_w * 8, 64);
    L.flags = o; o += align_up((size_t)KPN_SCENE_FLAG_FLOATS, 64);   // [0] = max |value| of the images and maps (kpn_common.h)
    L.total = o;
    return L;
}
int check_desc(const kpn_scene_desc* d) {
    KPN_REQUIRE(d != nullptr, "scene desc is null");
    KPN_REQUIRE(d->n_views >= 1 && d->n_views <= KPN_MAX_VIEWS, "n_views out of range");
    KPN_REQUIRE(d->src_h > 1 && d->src_w > 1 && d->geo0_h > 1 && d->geo0_w > 1 && d->geo1_h > 1 && d->geo1_w > 1 &&
                d->tex_h > 1 && d->tex_w > 1, "map sizes must be > 1");
    KPN_REQUIRE(d->zfar > d->znear && d->nml_scale > 0.0f && d->sigma > 0.0f, "bad scalar parameters");
    KPN_REQUIRE(d->KRT && d->extrin && d->kpt3d && d->img && d->geo0 && d->geo1 && d->tex, "null scene tensor");
    KPN_REQUIRE(d->disable_fg_mask || d->fg_mask, "fg_mask is null");
    return KPN_OK;
}
kpn_scene_dev scene_dev(const kpn_scene_desc* d, const void* ws) {
    const SceneLayout L = scene_layout(d);
    const float* base = static_cast<const float*>(ws);
    kpn_scene_dev s;
    s.V = d->n_views; s.H = d->src_h; s.W = d->src_w;
    s.g0h = d->geo0_h; s.g0w = d->geo0_w; s.g1h = d->geo1_h; s.g1w = d->geo1_w; s.th = d->tex_h; s.tw = d->tex_w;
    s.disable_fg_mask = d->disable_fg_mask;
    s.znear = d->znear; s.zfar = d->zfar; s.nml_scale = d->nml_scale;
    s.two_sigma2 = (float)(2.0 * ((double)d->sigma * (double)d->sigma));  // spatial.py:114
    s.keep = 0xFFFFFFFFu;
    s.table = base + L.table; s.rgbm = base + L.rgbm; s.geo0 = base + L.geo0; s.geo1 = base + L.geo1; s.tex = base + L.tex;
    s.flags = base + L.flags;
    return s;
}
}  // namespace

extern "C" size_t kpn_scene_workspace_bytes(const kpn_scene_desc* d) {
    if (check_desc(d) != KPN_OK) return 0;
    return scene_layout(d).total * sizeof(float);
}

extern "C" int kpn_scene_prepare(const kpn_scene_desc* d, void* scene_ws, void* stream) {
    if (int e = check_desc(d)) return e;
    KPN_REQUIRE(scene_ws != nullptr, "scene workspace is null");
    const SceneLayout L = scene_layout(d);
    float* base = static_cast<float*>(scene_ws);
    const int V = d->n_views;
    float* flags = base + L.flags;   // zeroed by k_scene_table (first on the stream), raised by the copies behind it
    KPN_LAUNCH(k_scene_table, dim3(1), dim3(64), stream, V, d->KRT, d->extrin, d->kpt3d, base + L.table, flags);
    const int64_t HW = (int64_t)d->src_h * d->src_w;
    KPN_LAUNCH(k_pack_rgbm, grid1d(V * HW, 256), dim3(256), stream, (int64_t)(V * HW), HW, d->img,
               d->disable_fg_mask ? (const uint8_t*)nullptr : d->fg_mask, base + L.rgbm, flags);
    const int64_t hw0 = (int64_t)d->geo0_h * d->geo0_w, hw1 = (int64_t)d->geo1_h * d->geo1_w, hwt = (int64_t)d->tex_h * d->tex_w;
    KPN_LAUNCH(k_nchw_to_nhwc, grid1d(V * hw0 * 64, 256), dim3(256), stream, (int64_t)(V * hw0 * 64), 64, hw0, d->geo0, base + L.geo0, flags);
    KPN_LAUNCH(k_nchw_to_nhwc, grid1d(V * hw1 * 8, 256), dim3(256), stream, (int64_t)(V * hw1 * 8), 8, hw1, d->geo1, base + L.geo1, flags);
    KPN_LAUNCH(k_nchw_to_nhwc, grid1d(V * hwt * 8, 256), dim3(256), stream, (int64_t)(V * hwt * 8), 8, hwt, d->tex, base + L.tex, flags);
    return check_launch("kpn_scene_prepare");
}

// ---------------------------------------------------------------------------------------------
// stage ops
extern "C" int kpn_ray_bbox_intersection(const float* bounds, const float* orig, const float* direct, int64_t R,
                                         float* near_o, float* far_o, uint8_t* hit_o, void* stream) {
    KPN_REQUIRE(bounds && orig && direct && near_o && far_o && hit_o, "null pointer");
    KPN_REQUIRE(R >= 0, "negative ray count");
    if (R == 0) return KPN_OK;
    KPN_LAUNCH(k_ray_bbox, grid1d(R, 256), dim3(256), stream, R, bounds, orig, direct, near_o, far_o, hit_o);
    return check_launch("kpn_ray_bbox_intersection");
}

extern "C" int kpn_make_rays(const float* K, const float* RT, float znear, float zfar, const float* bounds, int32_t x0,
                             int32_t y0, int32_t step, int32_t nx, int32_t ny, float* dirs, float* cam_pos,
                             float* near_o, float* far_o, void* stream) {
    KPN_REQUIRE(K && RT && bounds && dirs && cam_pos && near_o && far_o, "null pointer");
    KPN_REQUIRE(nx > 0 && ny > 0 && step > 0, "bad pixel grid");
    KPN_LAUNCH(k_make_rays, grid1d((int64_t)nx * ny, 256), dim3(256), stream, K, RT, znear, zfar, bounds, (int)x0, (int)y0,
               (int)step, (int)step, (int)nx, (int)ny, (const int*)nullptr, dirs, cam_pos, near_o, far_o);
    return check_launch("kpn_make_rays");
}

extern "C" int kpn_importance_sample(const float* contrib, const float* z, const float* u, int64_t R, int32_t Dm2,
                                     int32_t n, float* out, void* stream) {
    KPN_REQUIRE(contrib && z && out, "null pointer");
    KPN_REQUIRE(Dm2 >= 1 && Dm2 + 1 <= KPN_IS_MAXD, "bin count out of range (<= 128)");
    KPN_REQUIRE(n >= 1 && R >= 0, "bad sizes");
    if (R == 0) return KPN_OK;
    KPN_LAUNCH(k_importance, grid1d(R, 64), dim3(64), stream, R, (int)Dm2, (int)n, contrib, z, u, out);
    return check_launch("kpn_importance_sample");
}

// compositor launch: the kernel is specialised by samples per lane (ceil(S / 64)) so that the double-buffered ray
// state stays in few registers
static void launch_rgba2out(void* stream, int64_t R, int S, const float* rgba, const float* z, float* color, float* depth,
                            float* alpha, float* contrib, float* sdf, const int16_t* src, const float* rgba_new, int Sc) {
    const int64_t blocks = (R + 3) / 4;  // 4 waves per block, one ray per wave per iteration
    const dim3 grid((unsigned)(blocks < 8192 ? blocks : 8192));
    const int per = (S + 63) / 64;
    if (per <= 1) KPN_LAUNCH(k_rgba2out<1>, grid, dim3(256), stream, R, S, rgba, z, color, depth, alpha, contrib, sdf, src, rgba_new, Sc);
    else if (per <= 2) KPN_LAUNCH(k_rgba2out<2>, grid, dim3(256), stream, R, S, rgba, z, color, depth, alpha, contrib, sdf, src, rgba_new, Sc);
    else if (per <= 4) KPN_LAUNCH(k_rgba2out<4>, grid, dim3(256), stream, R, S, rgba, z, color, depth, alpha, contrib, sdf, src, rgba_new, Sc);
    else KPN_LAUNCH(k_rgba2out<KPN_MAX_PER_LANE>, grid, dim3(256), stream, R, S, rgba, z, color, depth, alpha, contrib, sdf, src, rgba_new, Sc);
}

extern "C" int kpn_rgba2out(const float* rgba, const float* z, int64_t R, int32_t S, float* color, float* depth,
                            float* alpha, float* contrib, float* sdf, void* stream) {
    KPN_REQUIRE(rgba && z && color && depth && alpha && sdf, "null pointer");
    KPN_REQUIRE(S >= 1 && S <= 64 * KPN_MAX_PER_LANE, "samples per ray out of range (<= 512)");
    if (R <= 0) return R == 0 ? KPN_OK : fail(KPN_EINVAL, "negative ray count");
    launch_rgba2out(stream, R, (int)S, rgba, z, color, depth, alpha, contrib, sdf, nullptr, nullptr, 0);
    return check_launch("kpn_rgba2out");
}
// compositor over the merged list of the fine pass, read in place from the coarse and the new samples' records
static int rgba2out_merged(const float* rgba_c, const float* rgba_n, const int16_t* src, const float* z, int64_t R, int Sc, int Sf,
                           float* color, float* depth, float* alpha, float* sdf, void* stream) {
    launch_rgba2out(stream, R, Sc + Sf, rgba_c, z, color, depth, alpha, nullptr, sdf, src, rgba_n, Sc);
    return check_launch("kpn_render_rays");
}

extern "C" int kpn_rgba2out_backward(const float* rgba, const float* z, int64_t R, int32_t S, const float* d_color,
                                     const float* d_depth, const float* d_alpha, const float* d_sdf, float* d_rgba, void* stream) {
    KPN_REQUIRE(rgba && z && d_rgba, "null pointer");
    KPN_REQUIRE(S >= 1, "bad sample count");
    if (R <= 0) return R == 0 ? KPN_OK : fail(KPN_EINVAL, "negative ray count");
    static const int serial = [] { const char* e = getenv("KPN_RGBA2OUT_BWD_SERIAL"); return e ? atoi(e) : 0; }();   // A/B knob
    const int per = serial ? 9 : (int)((S + 63) / 64);
    const dim3 wgrid = grid1d(R * 64, 256);   // one wavefront per ray
    if (per <= 1) KPN_LAUNCH(k_rgba2out_bwd_w<1>, wgrid, dim3(256), stream, R, (int)S, rgba, z, d_color, d_depth, d_alpha, d_sdf, d_rgba);
    else if (per <= 2) KPN_LAUNCH(k_rgba2out_bwd_w<2>, wgrid, dim3(256), stream, R, (int)S, rgba, z, d_color, d_depth, d_alpha, d_sdf, d_rgba);
    else if (per <= 4) KPN_LAUNCH(k_rgba2out_bwd_w<4>, wgrid, dim3(256), stream, R, (int)S, rgba, z, d_color, d_depth, d_alpha, d_sdf, d_rgba);
    else if (per <= 8) KPN_LAUNCH(k_rgba2out_bwd_w<8>, wgrid, dim3(256), stream, R, (int)S, rgba, z, d_color, d_depth, d_alpha, d_sdf, d_rgba);
    else KPN_LAUNCH(k_rgba2out_bwd, grid1d(R, 64), dim3(64), stream, R, (int)S, rgba, z, d_color, d_depth, d_alpha, d_sdf, d_rgba);
    return check_launch("kpn_rgba2out_backward");
}

// ---------------------------------------------------------------------------------------------
// field query
namespace {
// The row scratch (10 KB per valid tile and view) is capped: a pass with more valid tiles than fit is evaluated in
// batches that reuse it (kpn_batch, field_kernels.hip).  Sized for the worst case it was 32 GiB for a 512^2 frame at
// 64 + 64 samples — of which a scene uses the valid third; the cap keeps one pass per frame (one launch ramp, one
// weight staging) at a fixed, small footprint.  KPN_ROW_SCRATCH_MIB overrides the default of 3 GiB.
size_t g_row_scratch_cap = 0;   // 0 = not set yet: KPN_ROW_SCRATCH_MIB or the default
size_t row_scratch_cap_bytes() {
    if (g_row_scratch_cap == 0) {
        const char* e = getenv("KPN_ROW_SCRATCH_MIB");
#ifdef KPN_SIMT_EMU
        g_row_scratch_cap = e ? (size_t)atoll(e) << 20 : (size_t)1 << 20;
#else
        g_row_scratch_cap = (e ? (size_t)atoll(e) : (size_t)3072) << 20;
#endif
    }
    return g_row_scratch_cap;
}
const int kMaxBatches = 60;
const size_t kCounterBytes = 2048;   // (8 + 8 * kMaxBatches) ints
static_assert((8 + 8 * kMaxBatches) * sizeof(int) <= kCounterBytes, "counter block");
// passes of at most this many points always get their worst-case scratch (never batched): the backward entry points
// read a pass's rows again and work in passes of kBwdChunk points
#ifdef KPN_SIMT_EMU
const int64_t kUncappedPoints = 2048;
#else
const int64_t kUncappedPoints = 262144;
#endif
struct QueryLayout { size_t count, list, live, xscr, total; int tiles_cap, nbatch; };  // byte offsets
// pool: the POOL layout of the scratch (kpn_field_shared.h): the render / query passes with the pair-tile rows kernels
int geo_rows_mode();
// (the A/B knob KPN_NO_POOL is read once per process)
bool pool_layout_selected() {
    static const bool no_pool = [] { const char* e = getenv("KPN_NO_POOL"); return e && atoi(e) != 0; }();
    return geo_rows_mode() >= 2 && !no_pool;
}
QueryLayout query_layout(int64_t N, int V, bool pool = false) {
    QueryLayout L;
    size_t o = 0;
    // [0] valid count; batch b owns ints [8 + 8b, 16 + 8b): [0] rows ticket, [1] per-point ticket, [2] live points of the batch and
    // [3] pass B's ticket (density-first render passes), [4] / [5] the tickets of the fp32-range kernels launched behind them (range
    // guard), [6] the batch's "non-finite result" flag
    L.count = o; o += kCounterBytes;
    L.list = o; o += align_up((size_t)N * sizeof(int), 256);
    const size_t ntiles = (size_t)(N + KPN_TILE - 1) / KPN_TILE;
    const size_t tile_bytes = (size_t)kpn_tile_slabs(pool ? 1 : 0, V) * 64 * sizeof(float4);
    // monotone in N (a render workspace is laid out for its largest pass and used by smaller ones): never fewer tiles
    // than an uncapped pass of kUncappedPoints points needs
    size_t cap = row_scratch_cap_bytes() / tile_bytes;
    const size_t floor_tiles = (size_t)(kUncappedPoints + KPN_TILE - 1) / KPN_TILE;
    if (cap < floor_tiles) cap = floor_tiles;
    if ((ntiles + cap - 1) / cap > (size_t)kMaxBatches) cap = (ntiles + kMaxBatches - 1) / kMaxBatches;
    if (cap > ntiles) cap = ntiles ? ntiles : 1;
    L.tiles_cap = (int)cap;
    L.nbatch = (int)((ntiles + cap - 1) / cap);
    // the live list of ONE batch (density-first render passes, field_kernels.hip PHASE): scratch slots tile * 32 + point
    L.live = o; o += align_up(cap * KPN_TILE * sizeof(int), 256);
    L.xscr = o; o += align_up(cap * tile_bytes, 256);
    L.total = o;
    return L;
}
int field_grid_blocks() {
    // persistent grid: 256 CUs x 2 blocks of 256 threads (launch_bounds(256,2) -> 8 waves per CU)
#ifdef KPN_SIMT_EMU
    return 8;
#else
    static int blocks = [] { const char* e = getenv("KPN_GEO_BLOCKS"); return e ? atoi(e) : 512; }();  // tuning knob
    return blocks;
#endif
}
// ---- measurement hooks ----
#ifndef KPN_SIMT_EMU
struct ProfState {
    bool on = false;
    std::vector<hipEvent_t> ev;   // pairs
    int* counts_host = nullptr;   // pinned: the pass's valid count, one copy per recorded launch
    unsigned long long* clk_dev = nullptr;    // two shader-clock stamps per recorded launch (kpn_batch::clk), and their pinned copy
    unsigned long long* clk_host = nullptr;
    std::vector<kpn_batch> batch; // which batch of the pass the launch was
    size_t used = 0, cap = 0;
    int V = 0;
};
static ProfState g_prof;
#endif

// Rows kernel of layers1 (kpn_set_geo_rows_mode):
// 0: fp32 MFMA (v_mfma_f32_32x32x2_f32, k_geo_rows)
// 2: three bf16 pieces per operand, six products, two tiles per wave and ONE wave per SIMD (k_geo_rows_h2): fp32-class results
//    (every product term above 2^-24 relative is kept) in fp32's exponent range
// 3: two fp16 pieces per operand, three products (hh hl lh), same kernel structure (k_geo_rows_f2): the default — the same accuracy class
//    with 1.5x fewer MFMAs and a third of the split instructions; operands must stay within fp16's range, which the range guard
//    below takes care of
// (1 was the one-tile-per-wave split-bf16 kernel of round 1: not part of the library, scripts/mode1_investigation/)
#ifndef KPN_DEFAULT_GEO_ROWS_MODE
#define KPN_DEFAULT_GEO_ROWS_MODE 3
#endif
// The per-point kernel: 1 = k_fuse_color_h (weights as two fp16 pieces per value on v_mfma_f32_32x32x16_f16: the default),
// 0 = k_fuse_color (fp32 weights on v_mfma_f32_32x32x2_f32).  Process-wide (kpn_set_fuse_mode) or per call (kpn_render_args.fuse_kernel):
// two mechanisms, no environment variable (round 5).
int g_fuse_mode = -1;
int fuse_mode() {
    if (g_fuse_mode < 0) g_fuse_mode = 1;
    return g_fuse_mode;
}
// Density first (field_kernels.hip, PHASE): the per-point work of a render pass as pass A (density of every listed point + the
// batch's live list) and pass B (colour of the live points) instead of the fused per-point kernel — where it applies (lean render
// passes, POOL layout, fuse mode 1).  kpn_set_density_first: 0 = never (the fused kernel), 1 = always, 2 = AUTO (the default;
// KPN_DENSITY_FIRST=0/1/2 sets the initial value): the pair wins when enough of the hull is empty and loses 0.16 ms per launch when
// nothing is (field_kernels.hip), so each render pass takes the form the dead fraction of the EARLIER passes calls for — the
// per-point kernels count listed / live points on the device (kpn_density_counts), a 16-byte copy into pinned host memory is queued
// behind every pass, and the next pass looks at whatever has arrived: no synchronisation, and since both forms give the same bits
// the choice never shows in a frame.  A stream that is being captured neither allocates nor copies (the captured graph keeps the
// form chosen at capture time).
int g_density_first = -1;
int density_first() {
    if (g_density_first < 0) {
        const char* e = getenv("KPN_DENSITY_FIRST");
        g_density_first = (e && e[0] >= '0' && e[0] <= '2' && e[1] == 0) ? e[0] - '0' : 2;
    }
    return g_density_first;
}
const float kDensityFirstDeadFraction = 0.20f;   // AUTO: density first when at least this fraction of the hull's points was dead
struct DensityHint {
    unsigned long long* pinned = nullptr;   // [listed, live] as last copied from the device
    unsigned long long seen[2] = {0, 0};    // the snapshot the current decision was taken from
    double avg[2] = {0.0, 0.0};             // moving sums of listed / live points over the looks
    bool split = false;                     // nothing measured yet: the fused kernel
};
DensityHint g_density_hint[16];
DensityHint* density_hint() {
#ifndef KPN_SIMT_EMU
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    return &g_density_hint[dev];
#else
    return &g_density_hint[0];
#endif
}
bool stream_is_capturing(void* stream) {
#ifndef KPN_SIMT_EMU
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
#else
    (void)stream; return false;
#endif
}
long long g_density_first_passes[2] = {0, 0};   // eligible passes run density first / on the fused kernel (kpn_density_first_passes)
// the form of THIS pass
bool density_first_now() {
    const int m = density_first();
    if (m != 2) return m == 1;
    DensityHint* hnt = density_hint();
    if (!hnt) return false;
#ifndef KPN_SIMT_EMU
    if (!hnt->pinned) return hnt->split;
    const unsigned long long now[2] = {hnt->pinned[0], hnt->pinned[1]};
#else
    const unsigned long long now[2] = {kpn_density_counts[0], kpn_density_counts[1]};
#endif
    const bool was_reset = now[0] < hnt->seen[0];
    const unsigned long long d0 = was_reset ? now[0] : now[0] - hnt->seen[0], d1 = was_reset ? now[1] : now[1] - hnt->seen[1];
    if (d0 > 0 && d1 <= d0) {
        // a look may cover one pass only (a coarse pass's hull is emptier than a fine pass's): the decision follows a moving
        // average over the last few looks, not the last one
        hnt->avg[0] = 0.5 * hnt->avg[0] + (double)d0;
        hnt->avg[1] = 0.5 * hnt->avg[1] + (double)d1;
        hnt->split = (hnt->avg[0] - hnt->avg[1]) >= (double)kDensityFirstDeadFraction * hnt->avg[0];
        hnt->seen[0] = now[0]; hnt->seen[1] = now[1];
    }
    return hnt->split;
}
// behind a pass: the counters on their way to the host
void density_hint_refresh(void* stream) {
#ifndef KPN_SIMT_EMU
    if (density_first() != 2 || stream_is_capturing(stream)) return;
    DensityHint* hnt = density_hint();
    if (!hnt) return;
    if (!hnt->pinned) {
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 2 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return; }
        hnt->pinned = static_cast<unsigned long long*>(hp);
        hnt->pinned[0] = hnt->pinned[1] = 0;
    }
    void* dp = nullptr;
    if (hipGetSymbolAddress(&dp, HIP_SYMBOL(kpn_density_counts)) != hipSuccess) { (void)hipGetLastError(); return; }
    (void)hipMemcpyAsync(hnt->pinned, dp, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, (hipStream_t)stream);
#else
    (void)stream;
#endif
}
int g_geo_rows_mode = -1;
int geo_rows_mode() {
    if (g_geo_rows_mode < 0) g_geo_rows_mode = KPN_DEFAULT_GEO_ROWS_MODE;
    return g_geo_rows_mode;
}
int pair_grid_blocks() {   // k_geo_rows_h2: one 256-thread workgroup per CU = one wave per SIMD
#ifdef KPN_SIMT_EMU
    return 8;
#else
    static int blocks = [] {
        const char* e = getenv("KPN_H2_BLOCKS");
        if (e) return atoi(e);
        int dev = 0, cus = 0;   // one workgroup per compute unit of the current device (256 on an MI355X)
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus;
    }();
    return blocks;
#endif
}
int fuse_grid_blocks() {
#ifdef KPN_SIMT_EMU
    return 4;
#else
    return 256;
#endif
}
// points per k_mask_compact thread: as many as keep >= 8 workgroups per CU in flight
static inline int mask_points_per_thread(int64_t N) {
#ifndef KPN_SIMT_EMU
    const int64_t min_groups = 2048;
#else
    const int64_t min_groups = 2;  // so that the emulator tests walk the multi-point loop
#endif
    int ppt = KPN_MASK_PPT;
    while (ppt > 1 && N / (256 * (int64_t)ppt) < min_groups) ppt >>= 1;
    return ppt;
}
// ---- the range guard of the two-fp16-piece kernels (kpn_field_shared.h kpn_batch) ----
// On (the default) whenever rows mode 3 or fuse mode 1 is selected: those kernels stand aside on the device when the weights or
// the maps are beyond fp16's range, the per-point kernel flags a batch with a non-finite result, and the fp32-range kernels (rows
// mode 2, or 0 if selected; fuse mode 0) launched behind them evaluate such a batch again — two launches per batch that return
// at once otherwise.  KPN_NO_RANGE_GUARD=1 / kpn_set_range_guard(0): the round-3 behaviour (an operand beyond fp16's range makes
// the point NaN), for timing comparisons.
int g_range_guard = -1;
int range_guard() {
    if (g_range_guard < 0) { const char* e = getenv("KPN_NO_RANGE_GUARD"); g_range_guard = (e && atoi(e) != 0) ? 0 : 1; }
    return g_range_guard;
}
// batches evaluated again by the fp32-range kernels since the library was loaded, per device: a device GLOBAL of this module (one
// instance per device, zero-initialised when the module is loaded) that the per-point kernel of such a launch increments and
// kpn_range_guard_count reads.  No allocation, no synchronisation on the render path (round 4 hipMalloc'ed the counter inside the
// first guarded render: an advisor finding — a first frame captured into a HIP graph would have been invalidated).
#ifndef KPN_SIMT_EMU
__device__ int kpn_redone_batches;
#else
static int kpn_redone_batches;
#endif
int* redone_counter() {
#ifndef KPN_SIMT_EMU
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(kpn_redone_batches)) != hipSuccess) return nullptr;
    return static_cast<int*>(p);
#else
    return &kpn_redone_batches;
#endif
}

void launch_rows(int rows_mode, const kpn_scene_dev& sc, const kpn_points& ps, const float* wp, const int* list, const int* count,
                 int* tickets, float* xscr, const kpn_batch& batch, void* stream) {
    if (rows_mode >= 2) {
#ifdef KPN_SIMT_EMU
        if (batch.pool) {
            if (rows_mode == 3) KPN_LAUNCH(k_geo_rows_f2p, dim3(pair_grid_blocks()), dim3(256), stream, sc, ps, wp, list, count, tickets, xscr, batch);
            else KPN_LAUNCH(k_geo_rows_h2p, dim3(pair_grid_blocks()), dim3(256), stream, sc, ps, wp, list, count, tickets, xscr, batch);
        } else if (rows_mode == 3) KPN_LAUNCH(k_geo_rows_f2, dim3(pair_grid_blocks()), dim3(256), stream, sc, ps, wp, list, count, tickets, xscr, batch);
        else KPN_LAUNCH(k_geo_rows_h2, dim3(pair_grid_blocks()), dim3(256), stream, sc, ps, wp, list, count, tickets, xscr, batch);
#else
        kpn_internal_launch_geo_rows_pair(rows_mode, pair_grid_blocks(), stream, &sc, &ps, wp, list, count, tickets, xscr, &batch);
#endif
    } else {
        KPN_LAUNCH(k_geo_rows, dim3(field_grid_blocks()), dim3(256), stream, sc, ps, wp, list, count, tickets, xscr, batch);
    }
}
void launch_fuse(int fmode, const kpn_scene_dev& sc, const kpn_points& ps, const float* wp, const int* list, const int* count,
                 int* tickets, const float* xscr, int mode, int park_x, float* out, const kpn_batch& batch, int zero_skip, void* stream) {
    const int fblocks = fuse_grid_blocks();  // one 512-thread workgroup per CU: its 137 / 141 KB of weights sit in LDS
    static const int fthreads = [] { const char* e = getenv("KPN_FUSE_THREADS"); return e ? atoi(e) : 512; }();  // tuning knob
    const char* no_h3 = getenv("KPN_NO_FUSE_H3");   // A/B and test knob, read per call: the generic kernel for V = 3 as well
    if (fmode == 1 && sc.V == 3 && (sc.keep & 7u) == 7u && !(no_h3 && atoi(no_h3)))   // the shipped view count, no view dropped: the unrolled variant
        KPN_LAUNCH(k_fuse_color_h3, dim3(fblocks), dim3(fthreads), stream, sc, ps, wp, list, count, tickets, xscr, mode, park_x, out, batch, zero_skip);
    else if (fmode == 1)
        KPN_LAUNCH(k_fuse_color_h, dim3(fblocks), dim3(fthreads), stream, sc, ps, wp, list, count, tickets, xscr, mode, park_x, out, batch, zero_skip);
    else
        KPN_LAUNCH(k_fuse_color, dim3(fblocks), dim3(fthreads), stream, sc, ps, wp, list, count, tickets, xscr, mode, park_x, out, batch, zero_skip);
}
// the density-first pair of a render pass's batch: pass A, the gather records of the live points, pass B
void launch_density_first(const kpn_scene_dev& sc, const kpn_points& ps, const float* wp, const int* list, const int* count, int* tickets,
                          float* xscr, int* live, float* out, const kpn_batch& batch, void* stream) {
    const int fblocks = fuse_grid_blocks();
    KPN_LAUNCH(k_density_h, dim3(fblocks), dim3(512), stream, sc, ps, wp, list, count, tickets, xscr, live, out, batch);
#ifdef KPN_SIMT_EMU
    KPN_LAUNCH(k_row_records_live, dim3(8), dim3(256), stream, sc, ps, wp, list, count, (const int*)tickets, (const int*)live, xscr, batch);
#else
    kpn_internal_launch_row_records_live(2048, stream, &sc, &ps, wp, list, count, tickets, live, xscr, &batch);
#endif
    const char* no_h3 = getenv("KPN_NO_FUSE_H3");
    if (sc.V == 3 && (sc.keep & 7u) == 7u && !(no_h3 && atoi(no_h3)))
        KPN_LAUNCH(k_colour_h3, dim3(fblocks), dim3(512), stream, sc, ps, wp, list, count, tickets, (const float*)xscr, live, out, batch);
    else
        KPN_LAUNCH(k_colour_h, dim3(fblocks), dim3(512), stream, sc, ps, wp, list, count, tickets, (const float*)xscr, live, out, batch);
}

// rows_sel / fuse_sel: KPN_ROWS_* / KPN_FUSE_* of kpn_render_args (0 = the process-wide selection)
int run_field(const kpn_scene_dev& sc, const kpn_points& ps, const float* wp, int64_t N, int mode, float* out,
              uint8_t* valid, void* ws, void* stream, int lean = 0, int keep_rows = 0, int allow_pool = 0, int rows_sel = 0, int fuse_sel = 0) {
    // POOL layout of the scratch (kpn_field_shared.h): the pair-tile rows kernels pool over the views themselves — the eval render
    // passes and kpn_query; never when a backward pass reads the per-view rows again, never in the train branch (whose kept and
    // not-kept forward must stay bit-identical)
    const int rmode = rows_sel == KPN_ROWS_F32 ? 0 : (rows_sel == KPN_ROWS_BF16X3 ? 2 : (rows_sel == KPN_ROWS_F16X2 ? 3 : geo_rows_mode()));
    const int fmode = !out ? 0 : (fuse_sel == KPN_FUSE_F32 ? 0 : (fuse_sel == KPN_FUSE_F16X2 ? 1 : fuse_mode()));
    // (the workspace is laid out for the process-wide selection: a per-call rows kernel without the POOL layout uses the ROWS one)
    const int pool = (allow_pool && !keep_rows && out && pool_layout_selected() && rmode >= 2) ? 1 : 0;
    const QueryLayout L = query_layout(N, sc.V, pool != 0);
    char* base = static_cast<char*>(ws);
    int* count = reinterpret_cast<int*>(base + L.count);
    int* list = reinterpret_cast<int*>(base + L.list);
    int* live = reinterpret_cast<int*>(base + L.live);
    float* xscr = reinterpret_cast<float*>(base + L.xscr);
    hipMemsetAsync(count, 0, kCounterBytes, (hipStream_t)stream);
    const int ppt = mask_points_per_thread(N);
    KPN_LAUNCH(k_mask_compact, grid1d(N, 256 * ppt), dim3(256), stream, sc, ps, N, mode, lean, ppt, wp + kpn_scalar_off(), out, valid, list, count);
    if (keep_rows && L.nbatch > 1) return fail(KPN_EWORKSPACE, "a pass whose rows a backward call reads again must fit the row scratch");
    // which launches stand under the range guard: the two-fp16-piece ones; `redo`: the fp32-range pair behind them
    const bool guard = range_guard() && out && (rmode == 3 || fmode == 1);
    int* redone = guard ? redone_counter() : nullptr;
    const int safe_rmode = rmode == 3 ? 2 : rmode;
    // zero-density short path: render passes only (lean), never when a backward pass reads the rows again
    const char* zs = getenv("KPN_NO_ZERO_SKIP");   // A/B knob, read per call
    const int zero_skip = (lean && !keep_rows && !(zs && atoi(zs))) ? 1 : 0;
    const int park_x = 0;   // (rounds 2-3: x' parked in the row scratch between the per-point kernel's passes; gone with the one-pass statistics)
    // density first: where the exact short path applies (render passes: eval_func, no density noise), on the POOL layout with the
    // two-fp16-piece per-point arithmetic — the shipped configuration
    const bool eligible = zero_skip && mode == 1 && ps.noise == nullptr && pool && fmode == 1;
    const bool split = eligible && density_first_now();
    if (eligible) ++g_density_first_passes[split ? 0 : 1];
    for (int b = 0; b < L.nbatch; ++b) {
        int* slots = count + 8 + 8 * b;
        int* bad = guard ? slots + 6 : nullptr;
        kpn_batch b_rows{b, L.tiles_cap, (guard && rmode == 3) ? KPN_RUN_IF_SAFE : KPN_RUN_ALWAYS, bad, nullptr, pool, nullptr};
        const kpn_batch b_rec{b, L.tiles_cap, KPN_RUN_ALWAYS, nullptr, nullptr, pool, nullptr};
        const kpn_batch b_fuse{b, L.tiles_cap, (guard && fmode == 1) ? KPN_RUN_IF_SAFE : KPN_RUN_ALWAYS, bad, nullptr, pool, nullptr};
#ifndef KPN_SIMT_EMU
        const bool prof = g_prof.on && g_prof.used < g_prof.cap;
        if (prof) {
            if (rmode >= 2) b_rows.clk = g_prof.clk_dev + 2 * g_prof.used;
            (void)hipEventRecord(g_prof.ev[2 * g_prof.used], (hipStream_t)stream);
        }
#endif
        launch_rows(rmode, sc, ps, wp, list, count, slots + 0, xscr, b_rows, stream);
#ifndef KPN_SIMT_EMU
        if (prof) {
            (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], (hipStream_t)stream);
            (void)hipMemcpyAsync(g_prof.counts_host + g_prof.used, count, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
            g_prof.batch[g_prof.used] = b_rows;
            g_prof.V = sc.V;
            ++g_prof.used;
        }
#endif
        auto launch_records = [&](const kpn_batch& bb) {
#ifdef KPN_SIMT_EMU
            KPN_LAUNCH(k_row_records, dim3(8), dim3(256), stream, sc, ps, wp, (const int*)list, (const int*)count, xscr, bb);
#else
            kpn_internal_launch_row_records(2048, stream, &sc, &ps, wp, list, count, xscr, &bb);
#endif
        };
        // the colour head's gather records (the pair-tile rows kernels leave them to k_row_records; a density-first pass forms
        // them for its live points only)
        if (rmode >= 2 && !split) launch_records(b_rec);
        if (!out) continue;   // rows only (the backward entry points run their own per-point kernels)
        if (split) launch_density_first(sc, ps, wp, list, count, slots, xscr, live, out, b_fuse, stream);
        else launch_fuse(fmode, sc, ps, wp, list, count, slots, xscr, mode, park_x, out, b_fuse, zero_skip, stream);
        if (guard) {
            // The same batch again in fp32's exponent range, IF the kernels above stood aside or flagged it: the rows first (the
            // non-finite value may have come from either kernel; the gather records are intact — after a density-first pass they
            // exist for its live points only, so every point's are formed here), then the fused per-point kernel.
            const kpn_batch r_rows{b, L.tiles_cap, KPN_RUN_IF_UNSAFE, bad, nullptr, pool, nullptr};
            const kpn_batch r_fuse{b, L.tiles_cap, KPN_RUN_IF_UNSAFE, bad, redone, pool, nullptr};
            launch_rows(safe_rmode, sc, ps, wp, list, count, slots + 4, xscr, r_rows, stream);
            if (split) launch_records(r_rows);
            launch_fuse(0, sc, ps, wp, list, count, slots + 4, xscr, mode, park_x, out, r_fuse, zero_skip, stream);
        }
    }
    if (eligible) density_hint_refresh(stream);
    return check_launch("field query");
}
}  // namespace

extern "C" int kpn_set_geo_rows_mode(int32_t mode) {
    KPN_REQUIRE(mode == 0 || mode == 2 || mode == 3, "mode must be 0 (fp32 MFMA), 2 (three bf16 pieces) or 3 (two fp16 pieces); mode 1 is not part of the library");
    g_geo_rows_mode = mode;
    return KPN_OK;
}
extern "C" int kpn_get_geo_rows_mode(void) { return geo_rows_mode(); }
extern "C" int kpn_set_fuse_mode(int32_t mode) {
    KPN_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (fp32 MFMA) or 1 (two fp16 pieces per operand)");
    g_fuse_mode = mode;
    return KPN_OK;
}
extern "C" int kpn_get_fuse_mode(void) { return fuse_mode(); }
// points whose density the render passes' per-point kernels looked at since the last reset, and how many of them were live
extern "C" int kpn_density_stats(void* stream, int64_t* listed_host, int64_t* live_host, int32_t reset) {
    KPN_REQUIRE(listed_host && live_host, "null pointer");
    unsigned long long v[2] = {0, 0};
#ifndef KPN_SIMT_EMU
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(kpn_density_counts)) != hipSuccess) return fail(KPN_ELAUNCH, "no density counters in this module");
    if (hipMemcpyAsync(v, p, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        (reset && hipMemsetAsync(p, 0, sizeof(v), (hipStream_t)stream) != hipSuccess) ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return fail(KPN_ELAUNCH, "could not read the density counters");
#else
    (void)stream;
    v[0] = kpn_density_counts[0]; v[1] = kpn_density_counts[1];
    if (reset) kpn_density_counts[0] = kpn_density_counts[1] = 0;
#endif
    *listed_host = (int64_t)v[0];
    *live_host = (int64_t)v[1];
    return KPN_OK;
}
#if defined(KPN_PRECISION_PROBE) && !defined(KPN_SIMT_EMU)
// probe builds only (scripts/precision_budget.py): which lo pieces the two-fp16-piece kernels replace by zero (kpn_common.h)
extern "C" int kpn_internal_probe_set_mask_pair(unsigned long long m);
extern "C" int kpn_probe_set_mask(unsigned long long m) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(kpn_probe_mask_dev), &m, sizeof(m)) != hipSuccess) return fail(KPN_ELAUNCH, "probe mask");
    if (kpn_internal_probe_set_mask_pair(m)) return fail(KPN_ELAUNCH, "probe mask (pair unit)");
    return hipDeviceSynchronize() == hipSuccess ? KPN_OK : KPN_ELAUNCH;
}
#endif
extern "C" int kpn_density_first_passes(int64_t* density_first_host, int64_t* fused_host, int32_t reset) {
    KPN_REQUIRE(density_first_host && fused_host, "null pointer");
    *density_first_host = g_density_first_passes[0];
    *fused_host = g_density_first_passes[1];
    if (reset) g_density_first_passes[0] = g_density_first_passes[1] = 0;
    return KPN_OK;
}
extern "C" int kpn_set_density_first(int32_t mode) {
    KPN_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (fused per-point kernel), 1 (density first) or 2 (auto)");
    g_density_first = mode;
    return KPN_OK;
}
extern "C" int kpn_get_density_first(void) { return density_first(); }
extern "C" int kpn_set_range_guard(int32_t on) { g_range_guard = on ? 1 : 0; return KPN_OK; }
extern "C" int kpn_get_range_guard(void) { return range_guard(); }
// Batches of (point, view) rows that the fp32-range kernels evaluated again on the current device since the library was loaded
// (0 = every pass ran on the two-fp16-piece kernels).  Synchronises `stream`.
extern "C" int kpn_range_guard_count(void* stream, int64_t* batches_host) {
    KPN_REQUIRE(batches_host != nullptr, "null pointer");
    *batches_host = 0;
    int* c = redone_counter();
    if (!c) return fail(KPN_ELAUNCH, "could not allocate the range guard's counter");
    int v = 0;
#ifndef KPN_SIMT_EMU
    if (hipMemcpyAsync(&v, c, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return fail(KPN_ELAUNCH, "could not read the range guard's counter");
#else
    (void)stream; v = *c;
#endif
    *batches_host = v;
    return KPN_OK;
}

// Number of packed layers1 weights whose magnitude (after the folded activation scale) is beyond fp16's range, i.e. that rows
// mode 3 cannot represent (use mode 2 or 0 for such weights).  Reads four floats back from the device: synchronises `stream`.
extern "C" int kpn_packed_f16_range_check(const float* packed_dev, void* stream, int32_t* beyond) {
    KPN_REQUIRE(packed_dev && beyond, "null pointer");
    float fl[KPN_PACK_FLAG_FLOATS] = {0};
#ifndef KPN_SIMT_EMU
    if (hipMemcpyAsync(fl, packed_dev + kpn_pack_flags_off(), sizeof(fl), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return fail(KPN_ELAUNCH, "could not read the pack flags");
#else
    memcpy(fl, packed_dev + kpn_pack_flags_off(), sizeof(fl));
#endif
    *beyond = (int32_t)fl[0];
    return KPN_OK;
}

extern "C" size_t kpn_query_workspace_bytes(int64_t N, int32_t V) {
    if (N <= 0 || V <= 0) return 0;
    const size_t qa = query_layout(N, V, false).total, qb = query_layout(N, V, pool_layout_selected()).total;
    return qa > qb ? qa : qb;
}

extern "C" int kpn_query(const kpn_scene_desc* d, const void* scene_ws, const float* wp, int64_t N, const float* pts,
                         const float* view, int32_t mode, float* out, uint8_t* valid, void* ws, size_t ws_bytes,
                         void* stream) {
    if (int e = check_desc(d)) return e;
    KPN_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (raw query) or 1 (eval_func)");
    KPN_REQUIRE(N >= 0 && N < (1ll << 31), "point count out of range");
    if (N == 0) return KPN_OK;  // empty input: nothing to do (pointers may be null)
    KPN_REQUIRE(scene_ws && wp && pts && view && out && ws, "null pointer");
    if (ws_bytes < query_layout(N, d->n_views, pool_layout_selected()).total) return fail(KPN_EWORKSPACE, "query workspace too small");
    kpn_points ps{pts, view, nullptr, nullptr, nullptr, 1};
    return run_field(scene_dev(d, scene_ws), ps, wp, N, mode, out, valid, ws, stream, 0, 0, 1);
}

// ---------------------------------------------------------------------------------------------
// backward of the field evaluation
namespace {
const int64_t kBwdChunk = 262144;  // points per pass: V=3 -> 786432 rows x 4.3 KB of dumps = 3.4 GB
static_assert(kBwdChunk <= 262144, "kUncappedPoints (query_layout) must cover a backward pass");
#ifdef KPN_SIMT_EMU
const int kGradWorkers = 3;     // row workers (one workgroup each; its waves are the column groups)
#else
#ifndef KPN_GRAD_WORKERS
#define KPN_GRAD_WORKERS 512    // 2 workgroups per CU
#endif
const int kGradWorkers = KPN_GRAD_WORKERS;
#endif
const int kPartialUnits = 72;   // capacity of the partial-tile scratch in units of (workers x 2048 floats)
// full = 1: the whole-query reverse (adds the forward row scratch, the per-point dumps and the d x_view rows)
// colour-head dumps, floats per (point, view) row, in kpn_color_bufs order (X buffers then dA buffers)
const int kColorLd[25] = {4, 16, KPN_LD_XDIR, KPN_LD_XBL, 64, 32, 32, 32, 2, 32, 32, KPN_LD_XO0, 16, 8,
                          2, 8, 16, 2, 32, KPN_LD_DV11, 32, 32, 64, KPN_LD_XDIR, 16};
struct BwdLayout { size_t count, list, X0, X1, X2, X3, D0, D1, D2, D3, partial, dbp, xscr, Xp, Xh0, Xh1, D20, D21, D22, dxrows,
                   color, color_bytes, Dcmp, total; int64_t chunk; };
// full: 0 = geometry rows only; 1 = whole query, geometry outputs; 2 = whole query incl. the colour head
BwdLayout bwd_layout(int64_t N, int V, int full) {
    BwdLayout L{};
    L.chunk = N < kBwdChunk ? N : kBwdChunk;
    const size_t ntiles = (size_t)((L.chunk + KPN_TILE - 1) / KPN_TILE);
    const size_t npts = ntiles * KPN_TILE, rows = npts * V;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes, 256); return r; };
    L.count = take(256);
    L.list = take((size_t)L.chunk * sizeof(int));
    L.X0 = take(rows * KPN_LDX0 * 4); L.X1 = take(rows * 128 * 4); L.X2 = take(rows * KPN_LDX2 * 4); L.X3 = take(rows * 128 * 4);
    L.D0 = take(rows * 128 * 4); L.D1 = take(rows * 128 * 4); L.D2 = take(rows * 128 * 4); L.D3 = take(rows * 64 * 4);
    // partial tile blocks of all weight-gradient jobs of a pass (they run in shared launches): sum over the 19 layers of
    // column groups x MV = 65 -> 72 x workers x 8 KB (checked when the jobs are queued); bias partials per job
    L.partial = take((size_t)kPartialUnits * kGradWorkers * (2 * 16 * 64) * 4);
    L.dbp = take((size_t)KPN_WGRAD_MAX_JOBS * kGradWorkers * 4 * 64 * 4);
    if (full) {
        L.xscr = take(ntiles * (size_t)V * KPN_ROW_SLABS * 64 * sizeof(float4));
        L.Xp = take(npts * 128 * 4); L.Xh0 = take(npts * 64 * 4); L.Xh1 = take(npts * 64 * 4);
        L.D20 = take(npts * 64 * 4); L.D21 = take(npts * 64 * 4); L.D22 = take(npts * 2 * 4);
        L.dxrows = take(rows * 64 * 4);
    }
    if (full == 2) {
        size_t per_row = 0;
        for (int i = 0; i < 25; ++i) per_row += kColorLd[i];
        L.color_bytes = rows * per_row * 4;
        L.color = take(L.color_bytes);
        L.Dcmp = take(npts * 24 * 4);
    }
    L.total = o;
    return L;
}
size_t plain_w_off(int layer) {
    size_t o = 0;
    for (int l = 0; l < layer; ++l) o += (size_t)plain_dims[l][0] * plain_dims[l][1] + plain_dims[l][0];
    return o;
}
}  // namespace

// rows[0] = (point, view) rows of the current pass, rows[1] = points, both padded to whole tiles
__global__ void k_bwd_rows(const int* __restrict__ count, int V, int64_t* __restrict__ rows) {
    const int64_t npts = (int64_t)((*count + KPN_TILE - 1) / KPN_TILE) * KPN_TILE;
    rows[0] = npts * V;
    rows[1] = npts;
}

namespace {
// ---- measurement hooks of the backward (kpn_bwd_profile_enable / _collect): HIP events around each kernel group of a pass ----
enum { BP_ROWS_FWD = 0, BP_COLOR_BWD, BP_FUSE_BWD, BP_ROWS_BWD, BP_WGRAD, BP_KINDS };
#ifndef KPN_SIMT_EMU
struct BwdProf {
    bool on = false;
    std::vector<hipEvent_t> ev;        // pairs
    std::vector<int> kind;
    int* counts_host = nullptr;        // pinned: per recorded PASS the valid count and the view count / keep mask
    size_t used = 0, cap = 0, passes = 0;
};
static BwdProf g_bprof;
struct BwdProfScope {                  // brackets the launches made while it lives
    int slot = -1;
    void* stream;
    BwdProfScope(int kind, void* st) : stream(st) {
        if (!g_bprof.on || g_bprof.used >= g_bprof.cap) return;
        slot = (int)g_bprof.used++;
        g_bprof.kind[slot] = kind;
        (void)hipEventRecord(g_bprof.ev[2 * slot], (hipStream_t)stream);
    }
    ~BwdProfScope() { if (slot >= 0) (void)hipEventRecord(g_bprof.ev[2 * slot + 1], (hipStream_t)stream); }
};
#define KPN_BPROF(kind) BwdProfScope bprof_scope_(kind, stream)
#else
#define KPN_BPROF(kind) ((void)0)
#endif
// d_x != nullptr: geometry rows only, upstream gradient given per (point, view).  Otherwise the whole-query reverse from
// d_out (N,5): its geometry columns only (d_tex == nullptr) or all five incl. the colour head (d_tex != nullptr).
int run_backward(const kpn_scene_desc* d, const void* scene_ws, const float* wp, int64_t N, const float* pts, const float* view,
                 int mode, uint32_t keep_mask, const float* noise, float noise_std, const float* d_x, const float* d_out,
                 float* d_plain, float* d_geo0, float* d_geo1, float* d_tex, void* ws, size_t ws_bytes, void* stream,
                 const kpn_points* marched = nullptr, void* fwd_query_ws = nullptr) {
    // fwd_query_ws: the workspace a run_field() call on the SAME points just used: its valid list and row scratch are
    // reused instead of being recomputed (the train-branch backward runs the forward anyway to get rgba)
    // marched: the points are ray-marched (cam_pos + dirs * z, as kpn_render_rays evaluates them) instead of explicit;
    // one pass only (N <= kBwdChunk)
    const int V = d->n_views;
    const int full = d_x != nullptr ? 0 : (d_tex ? 2 : 1);
    const BwdLayout L = bwd_layout(N, V, full);
    if (ws_bytes < L.total) return fail(KPN_EWORKSPACE, "backward workspace too small");
    if (marched && N > L.chunk) return fail(KPN_EINVAL, "ray-marched backward pass too large");
    kpn_scene_dev sc = scene_dev(d, scene_ws);
    sc.keep = keep_mask;
    char* base = static_cast<char*>(ws);
    auto fp = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    int* count = reinterpret_cast<int*>(base + L.count);  // [0] valid count, [1..3] work tickets of the three persistent kernels
    int64_t* rows_dev = reinterpret_cast<int64_t*>(base + L.count + 64);
    int* list = reinterpret_cast<int*>(base + L.list);
    kpn_bwd_bufs B;
    B.X0 = fp(L.X0); B.X1 = fp(L.X1); B.X2 = fp(L.X2); B.X3 = fp(L.X3);
    B.D0 = fp(L.D0); B.D1 = fp(L.D1); B.D2 = fp(L.D2); B.D3 = fp(L.D3);
    B.dgeo0 = d_geo0; B.dgeo1 = d_geo1;
    kpn_fuse_bwd_bufs F{};
    if (full) {
        F.Xp = fp(L.Xp); F.Xh0 = fp(L.Xh0); F.Xh1 = fp(L.Xh1); F.D20 = fp(L.D20); F.D21 = fp(L.D21); F.D22 = fp(L.D22);
        F.dxrows = fp(L.dxrows);
    }
    kpn_color_bufs C{};
    if (full == 2) {
        const size_t rows = (size_t)((L.chunk + KPN_TILE - 1) / KPN_TILE) * KPN_TILE * V;
        float* q = fp(L.color);
        float** slots[25] = {&C.Xrd, &C.Xe1, &C.Xdir, &C.Xbl, &C.Xb1, &C.Xa, &C.Xv10, &C.Xv11, &C.Xt33, &C.Xv20, &C.Xv21, &C.Xo0,
                             &C.Xo1, &C.Xo2, &C.Do2, &C.Do1, &C.Do0, &C.Dv21, &C.Dv20, &C.Dv11, &C.Dv10, &C.Dbl1, &C.Dbl0,
                             &C.Dre1, &C.Dre0};
        for (int i = 0; i < 25; ++i) { *slots[i] = q; q += rows * kColorLd[i]; }
        C.Dcmp = fp(L.Dcmp);
        C.dtex = d_tex;
        C.dani = d_plain + kpn_plain_weight_floats() - 1;
        F.Dcmp = C.Dcmp;
    }
    float* partial = fp(L.partial);
    float* dbp = fp(L.dbp);
    const int blocks = field_grid_blocks();
    // rows of views switched off by the train-time dropout are skipped by k_color_bwd (their dumps are never written) and carry a
    // zero upstream gradient in k_geo_rows_bwd: k_weight_grad reads them as zeros by the same mask (no memset of the dumps)
    const uint32_t wgrad_keep = keep_mask | ~((V >= 32) ? 0xFFFFFFFFu : ((1u << V) - 1u));
    // dW[layer] += dY^T X over the rows (which = 0) or points (which = 1) of this pass
    // weight-gradient jobs of a pass: queued while the producers are launched, then run in one launch per MV class
    // and one reduce launch
    static const int wgrad_f32 = [] { const char* e = getenv("KPN_WGRAD_F32"); return e ? atoi(e) : 0; }();   // A/B knob: the fp32-MFMA form
    kpn_wgrad_jobs jobs[3], all;  // MV = 1, 2, 4
    int gzmax[3];
    size_t partial_used = 0;
    bool overflow = false;
    auto reset_jobs = [&]() { jobs[0].n = jobs[1].n = jobs[2].n = all.n = 0; gzmax[0] = gzmax[1] = gzmax[2] = 0; partial_used = 0; };
    // which: 0 = (point, view) rows, 1 = points.  Kc: columns of the X dump read (even); Kt: real input features
    auto wgrad = [&](int mv, int which, const float* dY, int ldy, int M, const float* X, int ldx, int Kc, int Kt, int layer,
                     int cmap, int omap) {
        const int cls = mv == 1 ? 0 : (mv == 2 ? 1 : 2);
        const int gz = (Kc + 63) / 64;
        kpn_wgrad_job j;
        j.dY = dY; j.X = X; j.ldy = ldy; j.M = M; j.ldx = ldx; j.Kc = Kc; j.Kt = Kt; j.cmap = cmap; j.omap = omap; j.mv = mv; j.which = which;
        j.V = which == 0 ? V : 1; j.keep = which == 0 ? wgrad_keep : 0xFFFFFFFFu;
        j.olab = wgrad_f32 ? 0 : 1;
        j.partial = partial + partial_used;
        partial_used += (size_t)gz * kGradWorkers * mv * 2048;
        if (partial_used > (size_t)kPartialUnits * kGradWorkers * 2048 || all.n >= KPN_WGRAD_MAX_JOBS) { overflow = true; return; }
        j.dbp = dbp + (size_t)all.n * kGradWorkers * 4 * 64;
        j.dW = d_plain + plain_w_off(layer);
        j.dB = j.dW + (size_t)plain_dims[layer][0] * plain_dims[layer][1];
        j.in_dim = plain_dims[layer][1];
        jobs[cls].j[jobs[cls].n++] = j;
        all.j[all.n++] = j;
        if (gz > gzmax[cls]) gzmax[cls] = gz;
    };
    auto run_jobs = [&]() {
        if (wgrad_f32) {
            if (jobs[0].n) KPN_LAUNCH(k_weight_grad_f32<1>, dim3(kGradWorkers, jobs[0].n), dim3(64 * gzmax[0]), stream, jobs[0], (const int64_t*)rows_dev);
            if (jobs[1].n) KPN_LAUNCH(k_weight_grad_f32<2>, dim3(kGradWorkers, jobs[1].n), dim3(64 * gzmax[1]), stream, jobs[1], (const int64_t*)rows_dev);
            if (jobs[2].n) KPN_LAUNCH(k_weight_grad_f32<4>, dim3(kGradWorkers, jobs[2].n), dim3(64 * gzmax[2]), stream, jobs[2], (const int64_t*)rows_dev);
        } else {
            if (jobs[0].n) KPN_LAUNCH(k_weight_grad<1>, dim3(kGradWorkers, jobs[0].n), dim3(64 * gzmax[0]), stream, jobs[0], (const int64_t*)rows_dev);
            if (jobs[1].n) KPN_LAUNCH(k_weight_grad<2>, dim3(kGradWorkers, jobs[1].n), dim3(64 * gzmax[1]), stream, jobs[1], (const int64_t*)rows_dev);
            if (jobs[2].n) KPN_LAUNCH(k_weight_grad<4>, dim3(kGradWorkers, jobs[2].n), dim3(64 * gzmax[2]), stream, jobs[2], (const int64_t*)rows_dev);
        }
        int gz_all = gzmax[0] > gzmax[1] ? gzmax[0] : gzmax[1];
        if (gzmax[2] > gz_all) gz_all = gzmax[2];
        const int mv_all = jobs[2].n ? 4 : (jobs[1].n ? 2 : 1);
        if (all.n) KPN_LAUNCH(k_weight_grad_reduce, dim3((mv_all * 2048 + mv_all * 32 + 31) / 32, gz_all, all.n), dim3(256), stream, all,
                              (int)kGradWorkers);
    };
    for (int64_t c0 = 0; c0 < N; c0 += L.chunk) {
        const int64_t n = (N - c0 < L.chunk) ? (N - c0) : L.chunk;
        const kpn_points ps = marched ? *marched
                                      : kpn_points{pts + c0 * 3, (view ? view : pts) + c0 * 3, nullptr, nullptr, nullptr, 1,
                                                   noise ? noise + c0 : nullptr, noise_std};
        reset_jobs();
        hipMemsetAsync(count, 0, 8 * sizeof(int), (hipStream_t)stream);
        const int* vcount = count;  // valid count of this pass
        float* xscr = full ? fp(L.xscr) : nullptr;
        if (fwd_query_ws && full) {
            const QueryLayout Q = query_layout(n, V);
            char* qb = static_cast<char*>(fwd_query_ws);
            vcount = reinterpret_cast<const int*>(qb + Q.count);
            list = reinterpret_cast<int*>(qb + Q.list);
            xscr = reinterpret_cast<float*>(qb + Q.xscr);
        } else {
            const int ppt = mask_points_per_thread(n);
            KPN_LAUNCH(k_mask_compact, grid1d(n, 256 * ppt), dim3(256), stream, sc, ps, n, 0, 1, ppt, wp + kpn_scalar_off(), (float*)nullptr,
                       (uint8_t*)nullptr, list, count);
        }
        KPN_LAUNCH(k_bwd_rows, dim3(1), dim3(1), stream, vcount, V, rows_dev);
#ifndef KPN_SIMT_EMU
        if (g_bprof.on && g_bprof.passes < g_bprof.cap) {   // the pass's valid count, for the FLOP models of kpn_bwd_profile_collect
            (void)hipMemcpyAsync(g_bprof.counts_host + 3 * g_bprof.passes, vcount, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
            g_bprof.counts_host[3 * g_bprof.passes + 1] = V;
            g_bprof.counts_host[3 * g_bprof.passes + 2] = (int)(keep_mask & ((V >= 31) ? 0x7FFFFFFFu : ((1u << V) - 1u)));
            ++g_bprof.passes;
        }
#endif
        if (full) {
            if (!fwd_query_ws) {
                KPN_BPROF(BP_ROWS_FWD);
                KPN_LAUNCH(k_geo_rows, dim3(blocks), dim3(256), stream, sc, ps, wp, (const int*)list, vcount, count + 1, xscr,
                           kpn_batch{0, 1 << 30});
            }
            if (full == 2) {
                KPN_BPROF(BP_COLOR_BWD);
                if (V <= 3)
                    KPN_LAUNCH(k_color_bwd<3>, dim3(blocks), dim3(256), stream, sc, ps, wp, (const int*)list, vcount, count + 4,
                               (const float*)xscr, d_out + c0 * 5, C);
                else
                    KPN_LAUNCH(k_color_bwd<KPN_MAXV>, dim3(blocks), dim3(256), stream, sc, ps, wp, (const int*)list, vcount, count + 4,
                               (const float*)xscr, d_out + c0 * 5, C);
            }
            if (full == 2) {
                wgrad(1, 0, C.Do2, 2, 1, C.Xo2, 8, 8, 8, P_O_2, 0, 0);
                wgrad(1, 0, C.Do1, 8, 8, C.Xo1, 16, 16, 16, P_O_1, 0, 0);
                wgrad(1, 0, C.Do0, 16, 16, C.Xo0, KPN_LD_XO0, KPN_LD_XO0, 37, P_O_0, 0, 0);
                wgrad(1, 0, C.Dv21, 2, 1, C.Xv21, 32, 32, 32, P_V2_1, 0, 0);
                wgrad(1, 0, C.Dv20, 32, 32, C.Xv20, 32, 32, 32, P_V2_0, 0, 0);
                wgrad(2, 0, C.Dv11, KPN_LD_DV11, 33, C.Xv11, 32, 32, 32, P_V1_1, 0, 0);
                wgrad(1, 0, C.Dv10, 32, 32, C.Xv10, 32, 32, 32, P_V1_0, 0, 0);
                wgrad(1, 0, C.Dbl1, 32, 32, C.Xb1, 64, 64, 64, P_BL_1, 0, 0);
                wgrad(2, 0, C.Dbl0, 64, 64, C.Xbl, KPN_LD_XBL, KPN_LD_XBL, KPN_LD_XBL, P_BL_0, 2, 0);
                wgrad(2, 0, C.Dre1, KPN_LD_XDIR, 35, C.Xe1, 16, 16, 16, P_RE_1, 0, 1);
                wgrad(1, 0, C.Dre0, 16, 16, C.Xrd, 4, 4, 4, P_RE_0, 0, 0);
            }
            {
                KPN_BPROF(BP_FUSE_BWD);
                KPN_LAUNCH(k_fuse_bwd, dim3(blocks), dim3(256), stream, sc, ps, wp, (const int*)list, vcount, count + 2,
                           (const float*)xscr, mode, d_out + c0 * 5, F);
            }
            wgrad(2, 1, F.D20, 64, 64, F.Xp, 128, 128, 128, P_G2_0, 0, 0);
            wgrad(2, 1, F.D21, 64, 64, F.Xh0, 64, 64, 64, P_G2_1, 0, 0);
            wgrad(1, 1, F.D22, 2, 2, F.Xh1, 64, 64, 64, P_G2_2, 0, 0);
            if (full == 2) wgrad(1, 1, C.Dcmp, 24, 24, F.Xp, 128, 128, 128, P_CMP, 0, 0);
        }
        {
            KPN_BPROF(BP_ROWS_BWD);
            KPN_LAUNCH(k_geo_rows_bwd, dim3(blocks), dim3(256), stream, sc, ps, wp, (const int*)list, vcount, count + 3,
                       full ? (const float*)F.dxrows : d_x + c0 * V * 64, full ? 1 : 0, B);
        }
        wgrad(4, 0, B.D0, 128, 128, B.X0, KPN_LDX0, 232, 232, P_G1_0, 1, 0);
        wgrad(4, 0, B.D1, 128, 128, B.X1, 128, 128, 128, P_G1_1, 0, 0);
        wgrad(4, 0, B.D2, 128, 120, B.X2, KPN_LDX2, 136, 136, P_G1_2, 0, 0);
        wgrad(2, 0, B.D3, 64, 64, B.X3, 128, 120, 120, P_G1_3, 0, 0);
        if (overflow) return fail(KPN_EWORKSPACE, "weight-gradient scratch too small (internal)");
        {
            KPN_BPROF(BP_WGRAD);
            run_jobs();
        }
    }
    return check_launch("field backward");
}
}  // namespace

#ifdef KPN_BWD_TIMING
extern "C" int kpn_bwd_timing(unsigned long long* out16) {   // read and clear (debug builds only; not part of the ABI)
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(kpn_bwd_cycles), 128) != hipSuccess) return 1;
    const unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(kpn_bwd_cycles), z, 128) != hipSuccess;
}
#endif
extern "C" size_t kpn_geo_rows_backward_workspace_bytes(int64_t N, int32_t V) {
    if (N <= 0 || V <= 0) return 0;
    return bwd_layout(N, V, 0).total;
}

extern "C" int kpn_geo_rows_backward(const kpn_scene_desc* d, const void* scene_ws, const float* wp, int64_t N,
                                     const float* pts, uint32_t keep_mask, const float* d_x, float* d_plain, float* d_geo0,
                                     float* d_geo1, void* ws, size_t ws_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    KPN_REQUIRE(N >= 0 && N < (1ll << 31), "point count out of range");
    if (N == 0) return KPN_OK;
    KPN_REQUIRE(scene_ws && wp && pts && d_x && d_plain && d_geo0 && d_geo1 && ws, "null pointer");
    return run_backward(d, scene_ws, wp, N, pts, nullptr, 0, keep_mask, nullptr, 0.0f, d_x, nullptr, d_plain, d_geo0, d_geo1, nullptr,
                        ws, ws_bytes, stream);
}

extern "C" size_t kpn_query_backward_geometry_workspace_bytes(int64_t N, int32_t V) {
    if (N <= 0 || V <= 0) return 0;
    return bwd_layout(N, V, 1).total;
}

extern "C" int kpn_query_backward_geometry(const kpn_scene_desc* d, const void* scene_ws, const float* wp, int64_t N,
                                           const float* pts, int32_t mode, uint32_t keep_mask, const float* noise,
                                           float noise_std, const float* d_out, float* d_plain, float* d_geo0, float* d_geo1,
                                           void* ws, size_t ws_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    KPN_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (raw query) or 1 (eval_func)");
    KPN_REQUIRE(N >= 0 && N < (1ll << 31), "point count out of range");
    if (N == 0) return KPN_OK;
    KPN_REQUIRE(scene_ws && wp && pts && d_out && d_plain && d_geo0 && d_geo1 && ws, "null pointer");
    return run_backward(d, scene_ws, wp, N, pts, nullptr, mode, keep_mask, noise, noise_std, nullptr, d_out, d_plain, d_geo0,
                        d_geo1, nullptr, ws, ws_bytes, stream);
}

extern "C" size_t kpn_query_backward_workspace_bytes(int64_t N, int32_t V) {
    if (N <= 0 || V <= 0) return 0;
    return bwd_layout(N, V, 2).total;
}

extern "C" int kpn_query_backward(const kpn_scene_desc* d, const void* scene_ws, const float* wp, int64_t N, const float* pts,
                                  const float* view, int32_t mode, uint32_t keep_mask, const float* noise, float noise_std,
                                  const float* d_out, float* d_plain, float* d_geo0, float* d_geo1, float* d_tex, void* ws,
                                  size_t ws_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    KPN_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (raw query) or 1 (eval_func)");
    KPN_REQUIRE(N >= 0 && N < (1ll << 31), "point count out of range");
    if (N == 0) return KPN_OK;
    KPN_REQUIRE(scene_ws && wp && pts && view && d_out && d_plain && d_geo0 && d_geo1 && d_tex && ws, "null pointer");
    return run_backward(d, scene_ws, wp, N, pts, view, mode, keep_mask, noise, noise_std, nullptr, d_out, d_plain, d_geo0, d_geo1,
                        d_tex, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------
// hierarchical render
namespace {
struct RenderLayout { size_t cam_pos, dirs, nearv, farv, zc, zf, zn, src, rgba, rgba_c, rgba_n, contrib, color, depth, alpha, sdf, query, total; int64_t chunk; };
int64_t pick_chunk(const kpn_scene_desc* d, const kpn_render_args* a) {
    // default: passes of up to 262144 rays (one per 512^2 frame) — large passes amortise launch ramps and the per-workgroup
    // weight staging of the persistent field kernels (measured per 512^2 frame: 65536 rays/pass 33.6 ms, 131072 33.3 ms,
    // 262144 32.2 ms).  The row scratch no longer scales with the pass: it is capped (query_layout) and reused by batches.
    const int64_t R = (int64_t)a->nx * a->ny;
    int64_t c = a->chunk_rays;
    if (c <= 0) {
        const int64_t cmax = 262144;
        const int64_t npass = (R + cmax - 1) / cmax;
        c = ((R + npass - 1) / npass + 63) / 64 * 64;
    }
    return c < R ? c : R;
}
RenderLayout render_layout(const kpn_scene_desc* d, const kpn_render_args* a) {
    RenderLayout L;
    const int64_t R = (int64_t)a->nx * a->ny;
    const int64_t C = pick_chunk(d, a);
    const int64_t Sfull = a->n_coarse + (a->fine ? a->n_fine : 0);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes, 256); return r; };
    L.chunk = C;
    L.cam_pos = take(64);
    L.dirs = take((size_t)R * 3 * 4);
    L.nearv = take((size_t)R * 4);
    L.farv = take((size_t)R * 4);
    L.zc = take((size_t)C * a->n_coarse * 4);
    L.zf = take((size_t)C * Sfull * 4);
    L.rgba = take((size_t)C * Sfull * 5 * 4);
    // eval with coarse re-use (the default): the coarse values kept for the fine pass and the values at the new samples share the
    // block a pass without re-use fills as a whole — never both in one call (round 6: 671 MB less per 512 x 512 plan)
    L.rgba_c = L.rgba;
    L.rgba_n = L.rgba + align_up((size_t)C * a->n_coarse * 5 * 4, 256);
    o += 256;                                                          // (the alignment of rgba_n inside the block)
    L.zn = take((size_t)C * (a->fine ? a->n_fine : 0) * 4);
    L.src = take((size_t)C * Sfull * sizeof(int16_t));
    L.contrib = take((size_t)C * a->n_coarse * 4);                    // the coarse compositor's weights (the fine one writes none)
    L.color = take((size_t)C * 3 * 4);
    L.depth = take((size_t)C * 4);
    L.alpha = take((size_t)C * 4);
    L.sdf = take((size_t)C * 4);
    {   // eval passes use the POOL layout of the scratch, the train branch (same workspace) the ROWS layout: room for either
        const size_t qa = query_layout(C * Sfull, d->n_views, false).total, qb = query_layout(C * Sfull, d->n_views, pool_layout_selected()).total;
        L.query = take(qa > qb ? qa : qb);
    }
    L.total = o;
    return L;
}
int check_render(const kpn_render_args* a) {
    KPN_REQUIRE(a != nullptr, "render args null");
    KPN_REQUIRE(a->K && a->RT && a->bounds, "null camera/bounds");
    KPN_REQUIRE(a->nx > 0 && a->ny > 0 && a->step > 0 && a->step_y >= 0, "bad pixel grid");
    KPN_REQUIRE(a->rows_kernel >= 0 && a->rows_kernel <= KPN_ROWS_F16X2 && a->fuse_kernel >= 0 && a->fuse_kernel <= KPN_FUSE_F16X2, "bad kernel selection");
    KPN_REQUIRE(a->n_coarse >= 3 && a->n_coarse <= 128, "sample_per_ray_c must be in [3,128]");
    KPN_REQUIRE(!a->fine || (a->n_fine >= 1 && a->n_fine <= 128), "sample_per_ray_f must be in [1,128]");
    KPN_REQUIRE((int64_t)a->nx * a->ny < (1ll << 31), "too many rays");
    return KPN_OK;
}
}  // namespace

// scatter of per-chunk (rays, C) results into the planar (C, ny*nx) outputs
__global__ void k_store_planar(int64_t r0, int64_t n, int64_t R, int C, const float* __restrict__ src, float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * C) return;
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    dst[(int64_t)c * R + r0 + r] = src[i];
}

extern "C" size_t kpn_render_workspace_bytes(const kpn_scene_desc* d, const kpn_render_args* a) {
    if (check_desc(d) != KPN_OK || check_render(a) != KPN_OK) return 0;
    return render_layout(d, a).total;
}

// shared implementation: t == nullptr -> eval branch (model.py:1019-1022, uniform=True); otherwise the train
// branch with explicit random draws
// importance samples of the fine pass + the merged depth list (k_fine_samples_w), Sc, Sf <= 128
static void launch_fine_samples(void* stream, int64_t n, int Sc, int Sf, const float* zc, const float* contrib,
                                const float* u, float* zf, float* znew, int16_t* src) {
    const dim3 grid((unsigned)(n + 3 < 4 * 8192 ? (n + 3) / 4 : 8192));
    if (Sc <= 64 && Sf <= 64)
        KPN_LAUNCH(k_fine_samples_w<true>, grid, dim3(256), stream, n, Sc, Sf, zc, contrib, u, zf, znew, src);
    else
        KPN_LAUNCH(k_fine_samples_w<false>, grid, dim3(256), stream, n, Sc, Sf, zc, contrib, u, zf, znew, src);
}

static int render_impl(const kpn_scene_desc* d, const void* scene_ws, const float* wp, const kpn_render_args* a,
                       const kpn_train_args* t, void* ws, size_t ws_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    if (int e = check_render(a)) return e;
    KPN_REQUIRE(scene_ws && wp && ws, "null pointer");
    if (t) {
        KPN_REQUIRE(t->pix && t->u_coarse, "train args: pix and u_coarse are required");
        KPN_REQUIRE(!a->fine || t->u_fine, "train args: u_fine is required when fine");
        KPN_REQUIRE(t->rand_noise_std == 0.0f || (t->noise_coarse && (!a->fine || t->noise_fine)), "train args: noise tensors missing");
        KPN_REQUIRE((t->keep_coarse & ((1u << d->n_views) - 1u)) && (t->keep_fine & ((1u << d->n_views) - 1u)),
                    "train args: view dropout must keep at least one view (reference src/model.py:744)");
    }
    const RenderLayout L = render_layout(d, a);
    if (ws_bytes < L.total) return fail(KPN_EWORKSPACE, "render workspace too small");
    char* base = static_cast<char*>(ws);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    kpn_scene_dev sc = scene_dev(d, scene_ws);
    const int64_t R = (int64_t)a->nx * a->ny;
    const int Sc = a->n_coarse, Sf = a->fine ? a->n_fine : 0, Sfull = Sc + Sf;
    KPN_LAUNCH(k_make_rays, grid1d(R, 256), dim3(256), stream, a->K, a->RT, a->znear, a->zfar, a->bounds, (int)a->x0, (int)a->y0,
               (int)a->step, (int)(a->step_y > 0 ? a->step_y : a->step), (int)a->nx, (int)a->ny, t ? (const int*)t->pix : (const int*)nullptr, F(L.dirs), F(L.cam_pos),
               F(L.nearv), F(L.farv));
    for (int64_t r0 = 0; r0 < R; r0 += L.chunk) {
        const int64_t n = (R - r0) < L.chunk ? (R - r0) : L.chunk;
        const float* dirs = F(L.dirs) + r0 * 3;
        KPN_LAUNCH(k_coarse_z, grid1d(n * Sc, 256), dim3(256), stream, n, Sc, (const float*)(F(L.nearv) + r0),
                   (const float*)(F(L.farv) + r0), t ? t->u_coarse + r0 * Sc : (const float*)nullptr, F(L.zc));
        kpn_points ps{nullptr, nullptr, F(L.cam_pos), dirs, F(L.zc), Sc,
                      (t && t->rand_noise_std != 0.0f) ? t->noise_coarse + r0 * Sc : nullptr, t ? t->rand_noise_std : 0.0f};
        sc.keep = t ? t->keep_coarse : 0xFFFFFFFFu;
        // eval: the fine pass re-uses the coarse samples' field values (identical points, no dropout, no noise: identical
        // deterministic results) and evaluates the field at the new samples only — 128 instead of 192 evaluations per ray
        // at 64 + 64 samples.  The train branch draws fresh dropout masks and noise for the fine query and cannot.
        const char* nr = getenv("KPN_NO_COARSE_REUSE");  // A/B knob (read per call: tests flip it)
        const bool no_reuse = nr && atoi(nr);
        const bool reuse = (t == nullptr) && a->fine && !no_reuse;
        float* rgba_coarse = reuse ? F(L.rgba_c) : F(L.rgba);
        const int allow_pool = t == nullptr;   // eval: pooled inside the rows kernel; the train branch keeps the per-view rows
        if (int e = run_field(sc, ps, wp, n * Sc, 1, rgba_coarse, nullptr, base + L.query, stream, 1, 0, allow_pool, a->rows_kernel, a->fuse_kernel)) return e;   // model.py:1062
        if (int e = kpn_rgba2out(rgba_coarse, F(L.zc), n, Sc, F(L.color), F(L.depth), F(L.alpha), F(L.contrib), F(L.sdf), stream)) return e;
        const kpn_render_stages* st = a->stages;
        auto copy_out = [&](float* dst, const float* src_, size_t floats) {   // device to device, on the call's stream
#ifndef KPN_SIMT_EMU
            (void)hipMemcpyAsync(dst, src_, floats * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
#else
            memcpy(dst, src_, floats * sizeof(float));
#endif
        };
        if (st && st->dirs) copy_out(st->dirs + r0 * 3, dirs, (size_t)n * 3);
        if (st && st->cam_pos && r0 == 0) copy_out(st->cam_pos, F(L.cam_pos), 3);
        if (st && st->z_coarse) copy_out(st->z_coarse + r0 * Sc, F(L.zc), (size_t)n * Sc);
        if (st && st->rgba_coarse) copy_out(st->rgba_coarse + r0 * Sc * 5, rgba_coarse, (size_t)n * Sc * 5);
        if (a->tex_fg) KPN_LAUNCH(k_store_planar, grid1d(n * 3, 256), dim3(256), stream, r0, n, R, 3, (const float*)F(L.color), a->tex_fg);
        if (a->depth) KPN_LAUNCH(k_store_planar, grid1d(n, 256), dim3(256), stream, r0, n, R, 1, (const float*)F(L.depth), a->depth);
        if (a->alpha) KPN_LAUNCH(k_store_planar, grid1d(n, 256), dim3(256), stream, r0, n, R, 1, (const float*)F(L.alpha), a->alpha);
        if (a->fine) {
            const float* uf = t ? t->u_fine + r0 * Sf : (const float*)nullptr;
            float* zn = reuse ? F(L.zn) : nullptr;
            int16_t* src = reuse ? reinterpret_cast<int16_t*>(base + L.src) : nullptr;
            launch_fine_samples(stream, n, Sc, Sf, F(L.zc), F(L.contrib), uf, F(L.zf), zn, src);
            sc.keep = t ? t->keep_fine : 0xFFFFFFFFu;
            if (reuse) {
                kpn_points pn{nullptr, nullptr, F(L.cam_pos), dirs, F(L.zn), Sf, nullptr, 0.0f};
                if (int e = run_field(sc, pn, wp, n * Sf, 1, F(L.rgba_n), nullptr, base + L.query, stream, 1, 0, allow_pool, a->rows_kernel, a->fuse_kernel)) return e;  // :1082, new samples
                if (int e = rgba2out_merged(F(L.rgba_c), F(L.rgba_n), src, F(L.zf), n, Sc, Sf, F(L.color), F(L.depth), F(L.alpha), F(L.sdf), stream)) return e;
            } else {
                kpn_points pf{nullptr, nullptr, F(L.cam_pos), dirs, F(L.zf), Sfull,
                              (t && t->rand_noise_std != 0.0f) ? t->noise_fine + r0 * Sfull : nullptr, t ? t->rand_noise_std : 0.0f};
                if (int e = run_field(sc, pf, wp, n * Sfull, 1, F(L.rgba), nullptr, base + L.query, stream, 1, 0, allow_pool, a->rows_kernel, a->fuse_kernel)) return e;  // :1082
                if (int e = kpn_rgba2out(F(L.rgba), F(L.zf), n, Sfull, F(L.color), F(L.depth), F(L.alpha), nullptr, F(L.sdf), stream)) return e;
            }
            if (st && st->z_fine) copy_out(st->z_fine + r0 * Sfull, F(L.zf), (size_t)n * Sfull);
            if (st && st->rgba_fine) {
                if (reuse) KPN_LAUNCH(k_merge_rgba, grid1d(n * Sfull, 256), dim3(256), stream, n, Sfull, Sc, (const float*)F(L.rgba_c), (const float*)F(L.rgba_n),
                                      (const int16_t*)src, st->rgba_fine + r0 * Sfull * 5);
                else copy_out(st->rgba_fine + r0 * Sfull * 5, F(L.rgba), (size_t)n * Sfull * 5);
            }
            if (a->tex_fg_fine) KPN_LAUNCH(k_store_planar, grid1d(n * 3, 256), dim3(256), stream, r0, n, R, 3, (const float*)F(L.color), a->tex_fg_fine);
            if (a->depth_fine) KPN_LAUNCH(k_store_planar, grid1d(n, 256), dim3(256), stream, r0, n, R, 1, (const float*)F(L.depth), a->depth_fine);
            if (a->alpha_fine) KPN_LAUNCH(k_store_planar, grid1d(n, 256), dim3(256), stream, r0, n, R, 1, (const float*)F(L.alpha), a->alpha_fine);
            if (a->sdf) KPN_LAUNCH(k_store_planar, grid1d(n, 256), dim3(256), stream, r0, n, R, 1, (const float*)F(L.sdf), a->sdf);
        }
    }
    return check_launch("kpn_render_rays");
}

extern "C" int kpn_render_rays(const kpn_scene_desc* d, const void* scene_ws, const float* wp, const kpn_render_args* a,
                               void* ws, size_t ws_bytes, void* stream) {
    return render_impl(d, scene_ws, wp, a, nullptr, ws, ws_bytes, stream);
}
extern "C" int kpn_render_rays_train(const kpn_scene_desc* d, const void* scene_ws, const float* wp, const kpn_render_args* a,
                                     const kpn_train_args* t, void* ws, size_t ws_bytes, void* stream) {
    KPN_REQUIRE(t != nullptr, "train args null");
    return render_impl(d, scene_ws, wp, a, t, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------
// backward of the train-branch render
namespace {
struct TrainBwdLayout { size_t cam_pos, dirs, nearv, farv, zc, zf, rgba_c, rgba_f, contrib, scratch, g3, g1a, g1b, g1c, drgba_c, drgba_f,
                        query, bwd, total; int64_t chunk; };
TrainBwdLayout train_bwd_layout(const kpn_scene_desc* d, const kpn_render_args* a) {
    TrainBwdLayout L;
    const int64_t R = (int64_t)a->nx * a->ny;
    const int64_t Sfull = a->n_coarse + a->n_fine;
    int64_t C = a->chunk_rays > 0 ? a->chunk_rays : kBwdChunk / Sfull;   // one backward pass per point set
    if (C * Sfull > kBwdChunk) C = kBwdChunk / Sfull;
    if (C > R) C = R;
    L.chunk = C;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes, 256); return r; };
    L.cam_pos = take(64);
    L.dirs = take((size_t)R * 3 * 4); L.nearv = take((size_t)R * 4); L.farv = take((size_t)R * 4);
    L.zc = take((size_t)C * a->n_coarse * 4); L.zf = take((size_t)C * Sfull * 4);
    L.rgba_c = take((size_t)C * a->n_coarse * 5 * 4); L.rgba_f = take((size_t)C * Sfull * 5 * 4);
    L.contrib = take((size_t)C * (Sfull > 8 ? Sfull : 8) * 4);
    L.scratch = take((size_t)C * 8 * 4);  // colour / depth / alpha / sdf of the recomputed forward (unused results)
    L.g3 = take((size_t)C * 3 * 4); L.g1a = take((size_t)C * 4); L.g1b = take((size_t)C * 4); L.g1c = take((size_t)C * 4);
    L.drgba_c = take((size_t)C * a->n_coarse * 5 * 4); L.drgba_f = take((size_t)C * Sfull * 5 * 4);
    L.query = take(query_layout(C * Sfull, d->n_views).total);
    L.bwd = take(bwd_layout(C * Sfull, d->n_views, 2).total);
    L.total = o;
    return L;
}
}  // namespace

// gather of a chunk's upstream gradients from the planar (C, R) layout the outputs use; src == nullptr -> zeros
__global__ void k_load_planar(int64_t r0, int64_t n, int64_t R, int C, const float* __restrict__ src, float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * C) return;
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    dst[i] = src ? src[(int64_t)c * R + r0 + r] : 0.0f;
}

// State a train-branch forward call can leave behind for its backward call (kpn_render_rays_train_keep): rays, and per
// chunk of rays the depths, the field values and — the expensive part — the valid lists and (point, view) rows of both
// field passes.  With it the backward call skips the forward it otherwise repeats (k_make_rays, k_coarse_z, 2 x
// k_mask_compact + k_geo_rows + k_fuse_color, compositor, sampler: 1.05 of 9.1 ms at 1024 rays x 192 samples).
namespace {
struct TrainStateLayout { size_t cam_pos, dirs, nearv, farv, chunk0, zc, zf, rgba_c, rgba_f, contrib, query_c, query_f, chunk_bytes, total;
                          int64_t chunk, nchunks; };
TrainStateLayout train_state_layout(const kpn_scene_desc* d, const kpn_render_args* a) {
    TrainStateLayout S;
    const TrainBwdLayout L = train_bwd_layout(d, a);
    const int64_t R = (int64_t)a->nx * a->ny, C = L.chunk;
    const int64_t Sfull = a->n_coarse + a->n_fine;
    S.chunk = C; S.nchunks = (R + C - 1) / C;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes, 256); return r; };
    S.cam_pos = take(64);
    S.dirs = take((size_t)R * 3 * 4); S.nearv = take((size_t)R * 4); S.farv = take((size_t)R * 4);
    S.chunk0 = o;
    o = 0;   // offsets inside one chunk block
    S.zc = take((size_t)C * a->n_coarse * 4); S.zf = take((size_t)C * Sfull * 4);
    S.rgba_c = take((size_t)C * a->n_coarse * 5 * 4); S.rgba_f = take((size_t)C * Sfull * 5 * 4);
    S.contrib = take((size_t)C * (Sfull > 8 ? Sfull : 8) * 4);   // also stages the fine composite of a forward call (6 floats per ray)
    S.query_c = take(query_layout(C * a->n_coarse, d->n_views).total);
    S.query_f = take(query_layout(C * Sfull, d->n_views).total);
    S.chunk_bytes = o;
    S.total = S.chunk0 + S.chunk_bytes * (size_t)S.nchunks;
    return S;
}

// one implementation, three uses: state == nullptr: the classic backward (forward repeated inside `ws`);
// forward_only: fills `state` and writes the outputs of `a` (kpn_render_rays_train_keep); otherwise: backward from `state`
int train_impl(const kpn_scene_desc* d, const void* scene_ws, const float* wp, const kpn_render_args* a, const kpn_train_args* t,
               const kpn_render_grads* g, float* d_plain, float* d_geo0, float* d_geo1, float* d_tex, void* state, size_t state_bytes,
               bool forward_only, void* ws, size_t ws_bytes, void* stream) {
    if (int e = check_desc(d)) return e;
    if (int e = check_render(a)) return e;
    KPN_REQUIRE(t != nullptr, "train args null");
    // the forward, its kept state and the backward's recompute must run the SAME kernels: the train branch follows the process-wide
    // selection only (kpn_set_geo_rows_mode / kpn_set_fuse_mode); a per-call selection is refused rather than silently replaced
    KPN_REQUIRE(a->rows_kernel == KPN_ROWS_DEFAULT && a->fuse_kernel == KPN_FUSE_DEFAULT,
                "the train branch takes the process-wide kernel selection: rows_kernel / fuse_kernel must be 0");
    KPN_REQUIRE(a->fine, "the train branch renders coarse + fine (dr_kwargs.fine)");
    KPN_REQUIRE(scene_ws && wp, "null pointer");
    KPN_REQUIRE(t->pix && t->u_coarse && t->u_fine, "train args: pix, u_coarse, u_fine are required");
    KPN_REQUIRE(t->rand_noise_std == 0.0f || (t->noise_coarse && t->noise_fine), "train args: noise tensors missing");
    KPN_REQUIRE((t->keep_coarse & ((1u << d->n_views) - 1u)) && (t->keep_fine & ((1u << d->n_views) - 1u)),
                "train args: view dropout must keep at least one view (reference src/model.py:744)");
    const bool backward = !forward_only;
    if (backward) KPN_REQUIRE(g && d_plain && d_geo0 && d_geo1 && d_tex && ws, "gradient pointers / workspace null");
    if (forward_only) KPN_REQUIRE(state != nullptr, "state null");
    const TrainBwdLayout L = train_bwd_layout(d, a);
    const TrainStateLayout S = train_state_layout(d, a);
    if (backward && ws_bytes < L.total) return fail(KPN_EWORKSPACE, "train backward workspace too small");
    if (state && state_bytes < S.total) return fail(KPN_EWORKSPACE, "train state too small");
    char* base = static_cast<char*>(ws);
    char* sbase = static_cast<char*>(state);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    kpn_scene_dev sc = scene_dev(d, scene_ws);
    const int64_t R = (int64_t)a->nx * a->ny;
    const int Sc = a->n_coarse, Sf = a->n_fine, Sfull = Sc + Sf;
    const bool run_forward = forward_only || state == nullptr;
    // ray set-up lives in the state when there is one
    float* cam_pos = state ? reinterpret_cast<float*>(sbase + S.cam_pos) : F(L.cam_pos);
    float* dirs_all = state ? reinterpret_cast<float*>(sbase + S.dirs) : F(L.dirs);
    float* nearv = state ? reinterpret_cast<float*>(sbase + S.nearv) : F(L.nearv);
    float* farv = state ? reinterpret_cast<float*>(sbase + S.farv) : F(L.farv);
    if (run_forward)
        KPN_LAUNCH(k_make_rays, grid1d(R, 256), dim3(256), stream, a->K, a->RT, a->znear, a->zfar, a->bounds, (int)a->x0, (int)a->y0,
                   (int)a->step, (int)a->step, (int)a->nx, (int)a->ny, (const int*)t->pix, dirs_all, cam_pos, nearv, farv);
    const float std_ = t->rand_noise_std;
    int64_t ci = 0;
    for (int64_t r0 = 0; r0 < R; r0 += L.chunk, ++ci) {
        const int64_t n = (R - r0) < L.chunk ? (R - r0) : L.chunk;
        const float* dirs = dirs_all + r0 * 3;
        char* cb = state ? sbase + S.chunk0 + S.chunk_bytes * (size_t)ci : nullptr;
        auto CS = [&](size_t s_off, size_t l_off) { return state ? reinterpret_cast<float*>(cb + s_off) : F(l_off); };
        float *zc = CS(S.zc, L.zc), *zf = CS(S.zf, L.zf), *rgba_c = CS(S.rgba_c, L.rgba_c), *rgba_f = CS(S.rgba_f, L.rgba_f);
        float* contrib = CS(S.contrib, L.contrib);
        char* query_c = state ? cb + S.query_c : base + L.query;
        char* query_f = state ? cb + S.query_f : base + L.query;
        kpn_points pc{nullptr, nullptr, cam_pos, dirs, zc, Sc, std_ != 0.0f ? t->noise_coarse + r0 * Sc : nullptr, std_};
        kpn_points pf{nullptr, nullptr, cam_pos, dirs, zf, Sfull, std_ != 0.0f ? t->noise_fine + r0 * Sfull : nullptr, std_};
        float* sc4 = forward_only ? nullptr : F(L.scratch);   // per-ray results of a repeated forward (unused)
        if (run_forward) {
            // ---- forward: z, rgba of the coarse pass ----
            KPN_LAUNCH(k_coarse_z, grid1d(n * Sc, 256), dim3(256), stream, n, Sc, (const float*)(nearv + r0), (const float*)(farv + r0),
                       t->u_coarse + r0 * Sc, zc);
            sc.keep = t->keep_coarse;
            if (int e = run_field(sc, pc, wp, n * Sc, 1, rgba_c, nullptr, query_c, stream, 1, 1)) return e;
        }
        if (forward_only) {
            // per-ray composites are staged in the (still unused) head of this chunk's fine rgba buffer: 8 floats per ray
            float* st = rgba_f;
            if (int e = kpn_rgba2out(rgba_c, zc, n, Sc, st, st + 3 * n, st + 4 * n, contrib, st + 5 * n, stream)) return e;
            if (a->tex_fg) KPN_LAUNCH(k_store_planar, grid1d(n * 3, 256), dim3(256), stream, r0, n, R, 3, (const float*)st, a->tex_fg);
            if (a->depth) KPN_LAUNCH(k_store_planar, grid1d(n, 256), dim3(256), stream, r0, n, R, 1, (const float*)(st + 3 * n), a->depth);
            if (a->alpha) KPN_LAUNCH(k_store_planar, grid1d(n, 256), dim3(256), stream, r0, n, R, 1, (const float*)(st + 4 * n), a->alpha);
        } else if (run_forward) {
            if (int e = kpn_rgba2out(rgba_c, zc, n, Sc, sc4, sc4 + 3 * n, sc4 + 4 * n, contrib, sc4 + 5 * n, stream)) return e;
        }
        const size_t bwd_bytes = backward ? L.total - L.bwd : 0;
        if (backward) {
            // ---- coarse pass reverse (in the classic call: before the fine forward overwrites the query workspace whose
            //      valid list and row scratch it reuses); sample positions carry no gradient (model.py:1038,1118) ----
            KPN_LAUNCH(k_load_planar, grid1d(n * 3, 256), dim3(256), stream, r0, n, R, 3, g->d_tex_fg, F(L.g3));
            KPN_LAUNCH(k_load_planar, grid1d(n, 256), dim3(256), stream, r0, n, R, 1, g->d_depth, F(L.g1a));
            KPN_LAUNCH(k_load_planar, grid1d(n, 256), dim3(256), stream, r0, n, R, 1, g->d_alpha, F(L.g1b));
            if (int e = kpn_rgba2out_backward(rgba_c, zc, n, Sc, F(L.g3), F(L.g1a), F(L.g1b), nullptr, F(L.drgba_c), stream)) return e;
            if (int e = run_backward(d, scene_ws, wp, n * Sc, nullptr, nullptr, 1, t->keep_coarse, nullptr, 0.0f, nullptr, F(L.drgba_c),
                                     d_plain, d_geo0, d_geo1, d_tex, base + L.bwd, bwd_bytes, stream, &pc, query_c)) return e;
        }
        if (run_forward) {
            // ---- fine pass: samples, forward ----
            launch_fine_samples(stream, n, Sc, Sf, zc, contrib, t->u_fine + r0 * Sf, zf, nullptr, nullptr);
            sc.keep = t->keep_fine;
            if (int e = run_field(sc, pf, wp, n * Sfull, 1, rgba_f, nullptr, query_f, stream, 1, 1)) return e;
        }
        if (forward_only) {
            float* st = contrib;   // the coarse contributions have been consumed by the sampler: 8 floats per ray of staging
            if (int e = kpn_rgba2out(rgba_f, zf, n, Sfull, st, st + 3 * n, st + 4 * n, nullptr, st + 5 * n, stream)) return e;
            if (a->tex_fg_fine) KPN_LAUNCH(k_store_planar, grid1d(n * 3, 256), dim3(256), stream, r0, n, R, 3, (const float*)st, a->tex_fg_fine);
            if (a->depth_fine) KPN_LAUNCH(k_store_planar, grid1d(n, 256), dim3(256), stream, r0, n, R, 1, (const float*)(st + 3 * n), a->depth_fine);
            if (a->alpha_fine) KPN_LAUNCH(k_store_planar, grid1d(n, 256), dim3(256), stream, r0, n, R, 1, (const float*)(st + 4 * n), a->alpha_fine);
            if (a->sdf) KPN_LAUNCH(k_store_planar, grid1d(n, 256), dim3(256), stream, r0, n, R, 1, (const float*)(st + 5 * n), a->sdf);
        }
        if (backward) {
            // ---- fine pass reverse ----
            KPN_LAUNCH(k_load_planar, grid1d(n * 3, 256), dim3(256), stream, r0, n, R, 3, g->d_tex_fg_fine, F(L.g3));
            KPN_LAUNCH(k_load_planar, grid1d(n, 256), dim3(256), stream, r0, n, R, 1, g->d_depth_fine, F(L.g1a));
            KPN_LAUNCH(k_load_planar, grid1d(n, 256), dim3(256), stream, r0, n, R, 1, g->d_alpha_fine, F(L.g1b));
            KPN_LAUNCH(k_load_planar, grid1d(n, 256), dim3(256), stream, r0, n, R, 1, g->d_sdf, F(L.g1c));
            if (int e = kpn_rgba2out_backward(rgba_f, zf, n, Sfull, F(L.g3), F(L.g1a), F(L.g1b), F(L.g1c), F(L.drgba_f), stream)) return e;
            if (int e = run_backward(d, scene_ws, wp, n * Sfull, nullptr, nullptr, 1, t->keep_fine, nullptr, 0.0f, nullptr, F(L.drgba_f),
                                     d_plain, d_geo0, d_geo1, d_tex, base + L.bwd, bwd_bytes, stream, &pf, query_f)) return e;
        }
    }
    return check_launch(forward_only ? "kpn_render_rays_train_keep" : "kpn_render_rays_train_backward");
}
}  // namespace

extern "C" size_t kpn_render_rays_train_backward_workspace_bytes(const kpn_scene_desc* d, const kpn_render_args* a) {
    if (check_desc(d) != KPN_OK || check_render(a) != KPN_OK || !a->fine) return 0;
    return train_bwd_layout(d, a).total;
}

extern "C" int kpn_render_rays_train_backward(const kpn_scene_desc* d, const void* scene_ws, const float* wp,
                                              const kpn_render_args* a, const kpn_train_args* t, const kpn_render_grads* g,
                                              float* d_plain, float* d_geo0, float* d_geo1, float* d_tex, void* ws,
                                              size_t ws_bytes, void* stream) {
    return train_impl(d, scene_ws, wp, a, t, g, d_plain, d_geo0, d_geo1, d_tex, nullptr, 0, false, ws, ws_bytes, stream);
}

extern "C" size_t kpn_render_rays_train_state_bytes(const kpn_scene_desc* d, const kpn_render_args* a) {
    if (check_desc(d) != KPN_OK || check_render(a) != KPN_OK || !a->fine) return 0;
    return train_state_layout(d, a).total;
}
extern "C" int kpn_render_rays_train_keep(const kpn_scene_desc* d, const void* scene_ws, const float* wp, const kpn_render_args* a,
                                          const kpn_train_args* t, void* state, size_t state_bytes, void* stream) {
    return train_impl(d, scene_ws, wp, a, t, nullptr, nullptr, nullptr, nullptr, nullptr, state, state_bytes, true, nullptr, 0, stream);
}
extern "C" int kpn_render_rays_train_backward_kept(const kpn_scene_desc* d, const void* scene_ws, const float* wp,
                                                   const kpn_render_args* a, const kpn_train_args* t, const kpn_render_grads* g,
                                                   float* d_plain, float* d_geo0, float* d_geo1, float* d_tex, void* state,
                                                   size_t state_bytes, void* ws, size_t ws_bytes, void* stream) {
    KPN_REQUIRE(state != nullptr, "state null");
    return train_impl(d, scene_ws, wp, a, t, g, d_plain, d_geo0, d_geo1, d_tex, state, state_bytes, false, ws, ws_bytes, stream);
}

extern "C" int kpn_frame_to_rgb8(const float* chw, int32_t H, int32_t W, int32_t bgr, uint8_t* hwc_out, void* stream) {
    KPN_REQUIRE(chw && hwc_out, "null pointer");
    KPN_REQUIRE(H > 0 && W > 0 && (int64_t)H * W < (1ll << 31), "bad frame size");
    KPN_LAUNCH(k_frame_to_rgb8, grid1d((int64_t)H * W, 256), dim3(256), stream, (int)(H * W), (int)bgr, chw, hwc_out);
    return check_launch("kpn_frame_to_rgb8");
}
extern "C" int kpn_mse_psnr(const float* pred, const float* gt, int64_t n, double* out2, void* scratch, void* stream) {
    KPN_REQUIRE(pred && gt && out2 && scratch, "null pointer");
    KPN_REQUIRE(n > 0, "empty image");
    double* partial = static_cast<double*>(scratch);
    int* ticket = reinterpret_cast<int*>(partial + 2048);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    (void)hipMemsetAsync(ticket, 0, sizeof(int), (hipStream_t)stream);
    KPN_LAUNCH(k_mse_psnr, dim3((unsigned)blocks), dim3(256), stream, n, pred, gt, partial, ticket, out2);
    return check_launch("kpn_mse_psnr");
}

extern "C" int kpn_pix_l1_loss(const float* src, const float* tar, int64_t n, float lambda, float* loss, float* d_src, void* scratch,
                               void* stream) {
    KPN_REQUIRE(src && tar && loss && scratch, "null pointer");
    KPN_REQUIRE(n > 0, "empty image");
    double* partial = static_cast<double*>(scratch);
    int* ticket = reinterpret_cast<int*>(partial + 2048);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    (void)hipMemsetAsync(ticket, 0, sizeof(int), (hipStream_t)stream);
    KPN_LAUNCH(k_pix_l1, dim3((unsigned)blocks), dim3(256), stream, n, lambda, src, tar, partial, ticket, loss, d_src);
    return check_launch("kpn_pix_l1_loss");
}

extern "C" size_t kpn_ssim_scratch_bytes(int32_t w, int32_t h) {
    if (w < 7 || h < 7) return 0;
    return align_up((size_t)5 * 3 * (h - 6) * w * sizeof(float), 256) + 2048 * sizeof(double) + 256;
}
extern "C" int kpn_ssim(const float* pred_chw, const float* gt_chw, int32_t H, int32_t W, int32_t x0, int32_t y0, int32_t w,
                        int32_t h, double* out, void* scratch, void* stream) {
    KPN_REQUIRE(pred_chw && gt_chw && out && scratch, "null pointer");
    KPN_REQUIRE(x0 >= 0 && y0 >= 0 && w >= 7 && h >= 7 && x0 + w <= W && y0 + h <= H, "crop must lie inside the image and be at least 7x7 (win_size)");
    char* base = static_cast<char*>(scratch);
    float* tmp = reinterpret_cast<float*>(base);
    double* partial = reinterpret_cast<double*>(base + align_up((size_t)5 * 3 * (h - 6) * w * sizeof(float), 256));
    int* ticket = reinterpret_cast<int*>(partial + 2048);
    (void)hipMemsetAsync(ticket, 0, sizeof(int), (hipStream_t)stream);
    KPN_LAUNCH(k_ssim_vertical, grid1d((int64_t)3 * (h - 6) * w, 256), dim3(256), stream, pred_chw, gt_chw, (int)H, (int)W, (int)x0, (int)y0,
               (int)w, (int)h, tmp);
    int64_t blocks = ((int64_t)3 * (h - 6) * (w - 6) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    KPN_LAUNCH(k_ssim_map, dim3((unsigned)blocks), dim3(256), stream, (const float*)tmp, (int)w, (int)h, partial, ticket, out);
    return check_launch("kpn_ssim");
}

extern "C" int kpn_bwd_profile_enable(int32_t on) {
#ifndef KPN_SIMT_EMU
    if (on && g_bprof.cap == 0) {
        g_bprof.cap = 4096;
        g_bprof.ev.resize(2 * g_bprof.cap);
        g_bprof.kind.resize(g_bprof.cap);
        for (auto& e : g_bprof.ev) if (hipEventCreate(&e) != hipSuccess) return fail(KPN_ELAUNCH, "hipEventCreate failed");
        if (hipHostMalloc((void**)&g_bprof.counts_host, g_bprof.cap * 3 * sizeof(int), 0) != hipSuccess) return fail(KPN_ELAUNCH, "hipHostMalloc failed");
    }
    g_bprof.on = on != 0;
    g_bprof.used = 0;
    g_bprof.passes = 0;
#endif
    return KPN_OK;
}
extern "C" int kpn_bwd_profile_collect(double* ms5, int64_t* launches5, int64_t* rows_host, int64_t* kept_rows_host, int64_t* points_host) {
    KPN_REQUIRE(ms5 && launches5 && rows_host && kept_rows_host && points_host, "null pointer");
    for (int i = 0; i < BP_KINDS; ++i) { ms5[i] = 0.0; launches5[i] = 0; }
    *rows_host = *kept_rows_host = *points_host = 0;
#ifndef KPN_SIMT_EMU
    for (size_t i = 0; i < g_bprof.used; ++i) {
        if (hipEventSynchronize(g_bprof.ev[2 * i + 1]) != hipSuccess) return fail(KPN_ELAUNCH, "hipEventSynchronize failed");
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, g_bprof.ev[2 * i], g_bprof.ev[2 * i + 1]) != hipSuccess) return fail(KPN_ELAUNCH, "hipEventElapsedTime failed");
        ms5[g_bprof.kind[i]] += ms;
        ++launches5[g_bprof.kind[i]];
    }
    if (hipDeviceSynchronize() != hipSuccess) return fail(KPN_ELAUNCH, "hipDeviceSynchronize failed");
    for (size_t p_ = 0; p_ < g_bprof.passes; ++p_) {
        const int64_t cnt = g_bprof.counts_host[3 * p_], V = g_bprof.counts_host[3 * p_ + 1];
        *points_host += cnt;
        *rows_host += cnt * V;
        *kept_rows_host += cnt * __builtin_popcount((unsigned)g_bprof.counts_host[3 * p_ + 2]);
    }
    g_bprof.used = 0;
    g_bprof.passes = 0;
#endif
    return KPN_OK;
}
extern "C" int kpn_profile_enable(int32_t on) {
#ifndef KPN_SIMT_EMU
    if (on && g_prof.cap == 0) {
        g_prof.cap = 8192;
        g_prof.ev.resize(2 * g_prof.cap);
        g_prof.batch.resize(g_prof.cap);
        for (auto& e : g_prof.ev) if (hipEventCreate(&e) != hipSuccess) return fail(KPN_ELAUNCH, "hipEventCreate failed");
        if (hipHostMalloc((void**)&g_prof.counts_host, g_prof.cap * sizeof(int), 0) != hipSuccess)
            return fail(KPN_ELAUNCH, "hipHostMalloc failed");
        if (hipMalloc((void**)&g_prof.clk_dev, g_prof.cap * 2 * sizeof(unsigned long long)) != hipSuccess ||
            hipHostMalloc((void**)&g_prof.clk_host, g_prof.cap * 2 * sizeof(unsigned long long), 0) != hipSuccess)
            return fail(KPN_ELAUNCH, "hipMalloc failed");
    }
    if (on && hipMemset(g_prof.clk_dev, 0, g_prof.cap * 2 * sizeof(unsigned long long)) != hipSuccess) return fail(KPN_ELAUNCH, "hipMemset failed");
    g_prof.on = on != 0;
    g_prof.used = 0;
#endif
    return KPN_OK;
}
extern "C" int kpn_profile_collect3(double* ms_out, int64_t* launches_out, int64_t* rows_out, int64_t* surplus_out, double* clock_ghz_out) {
    KPN_REQUIRE(ms_out && launches_out && rows_out && surplus_out, "null pointer");
    *ms_out = 0.0; *launches_out = 0; *rows_out = 0; *surplus_out = 0;
    if (clock_ghz_out) *clock_ghz_out = 0.0;
#ifndef KPN_SIMT_EMU
    double cyc = 0.0, cyc_ms = 0.0;
    if (g_prof.used && g_prof.clk_dev) {
        if (hipEventSynchronize(g_prof.ev[2 * (g_prof.used - 1) + 1]) != hipSuccess ||
            hipMemcpy(g_prof.clk_host, g_prof.clk_dev, g_prof.used * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(KPN_ELAUNCH, "could not read the clock stamps");
    }
    for (size_t i = 0; i < g_prof.used; ++i) {
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return fail(KPN_ELAUNCH, "hipEventSynchronize failed");
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return fail(KPN_ELAUNCH, "hipEventElapsedTime failed");
        // rows of this launch: the same arithmetic as kpn_batch_range (field_kernels.hip) on the pass's valid count
        const int64_t count = g_prof.counts_host[i];
        const kpn_batch b = g_prof.batch[i];
        const int64_t ntiles = (count + KPN_TILE - 1) / KPN_TILE;
        const int64_t nb = (ntiles + b.tiles_cap - 1) / b.tiles_cap;
        if (b.index >= nb) { ++*surplus_out; continue; }     // surplus batch: returned at once, nothing processed
        const int64_t p0 = ntiles * b.index / nb * KPN_TILE, p1 = ntiles * (b.index + 1) / nb * KPN_TILE;
        *rows_out += ((p1 < count ? p1 : count) - p0) * g_prof.V;
        *ms_out += ms;
        ++*launches_out;
        if (g_prof.clk_host && g_prof.clk_host[2 * i + 1] > g_prof.clk_host[2 * i]) {   // pair-tile kernels only
            cyc += (double)(g_prof.clk_host[2 * i + 1] - g_prof.clk_host[2 * i]);
            cyc_ms += ms;
        }
    }
    // shader cycles the first workgroup spent in the launches / the launches' event time: a lower bound of the sustained clock
    // (the workgroup ends a little before its launch does)
    if (clock_ghz_out && cyc_ms > 0.0) *clock_ghz_out = cyc / (cyc_ms * 1e6);
    g_prof.used = 0;
#endif
    return KPN_OK;
}
extern "C" int kpn_profile_collect2(double* ms_out, int64_t* launches_out, int64_t* rows_out, int64_t* surplus_out) {
    return kpn_profile_collect3(ms_out, launches_out, rows_out, surplus_out, nullptr);
}
extern "C" int kpn_profile_collect(double* ms_out, int64_t* launches_out, int64_t* rows_out) {
    int64_t surplus = 0;
    return kpn_profile_collect2(ms_out, launches_out, rows_out, &surplus);
}
extern "C" size_t kpn_row_scratch_cap_bytes(void) { return row_scratch_cap_bytes(); }
extern "C" int kpn_set_row_scratch_cap_bytes(size_t bytes) {
    KPN_REQUIRE(bytes >= ((size_t)1 << 20), "the row scratch cap must be at least 1 MiB");
    g_row_scratch_cap = bytes;   // workspaces sized before the change must be re-queried (kpn_*_workspace_bytes)
    return KPN_OK;
}
extern "C" double kpn_flops_per_row(void) { return 2.0 * 70080.0; }

extern "C" double kpn_flops_per_point(int32_t V) {
    // algorithmic MACs: per (point,view) 70,080 (layers1) + 13,256 (IBR head);
    // per point 12,416 (layers2) + 3,072 (compress)
    return 2.0 * ((70080.0 + 13256.0) * V + 12416.0 + 3072.0);
}

extern "C" int kpn_selftest_mfma(float* scratch, void* stream, float* max_err_host) {
    KPN_REQUIRE(scratch && max_err_host, "null pointer");
    float A[64], B[64], Dm[1024];
    for (int i = 0; i < 64; ++i) { A[i] = 0.37f * i - 7.0f + 0.011f * i * i; B[i] = 3.0f - 0.23f * i + (i % 5) * 0.7f; }
    // diagnostic: fully synchronous (pageable host buffers), every runtime call checked
#define KPN_HIP_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(KPN_ELAUNCH, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
    for (int i = 0; i < 1024; ++i) Dm[i] = -12345.0f;
    KPN_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    KPN_HIP_TRY(hipMemcpy(scratch, A, sizeof(A), hipMemcpyHostToDevice));
    KPN_HIP_TRY(hipMemcpy(scratch + 64, B, sizeof(B), hipMemcpyHostToDevice));
    KPN_LAUNCH(k_selftest_mfma, dim3(1), dim3(64), stream, (const float*)scratch, (const float*)(scratch + 64), scratch + 128);
    if (int e = check_launch("k_selftest_mfma launch")) return e;
    KPN_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    KPN_HIP_TRY(hipMemcpy(Dm, scratch + 128, sizeof(Dm), hipMemcpyDeviceToHost));
    float me = 0.0f;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            const float ref = fmaf(A[i * 2 + 1], B[32 + j], A[i * 2] * B[j]);
            me = fmaxf(me, fabsf(ref - Dm[i * 32 + j]));
        }
    *max_err_host = me;
    if (int e = check_launch("kpn_selftest_mfma")) return e;
    if (!(me < 1e-3f)) {
        char buf[256];
        snprintf(buf, sizeof(buf), "MFMA lane map mismatch: max err %g, D[0][0]=%g (ref %g), D[5][7]=%g (ref %g)", (double)me,
                 (double)Dm[0], (double)fmaf(A[1], B[32], A[0] * B[0]), (double)Dm[5 * 32 + 7],
                 (double)fmaf(A[11], B[39], A[10] * B[7]));
        return fail(KPN_ELAUNCH, buf);
    }
    return KPN_OK;
}

#if defined(KPN_FUSE_TIMING) && !defined(KPN_SIMT_EMU)
// debug builds only: read (and clear) the per-phase cycle sums of k_fuse_color
extern "C" int kpn_fuse_timing(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(kpn_fuse_cycles), 64) != hipSuccess) return 1;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(kpn_fuse_cycles), z, 64) != hipSuccess;
}
#endif
