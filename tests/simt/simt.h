// simt.h — a wave64 SIMT emulator for running the HIP kernels of keypointnerf_amd/csrc on the host.
//
// TEST INFRASTRUCTURE ONLY.  The build container has no GPU and GPU minutes are scarce, so the
// kernels' index arithmetic, MFMA operand layouts, weight packing and control flow are exercised on
// the CPU first: the very same kernel sources are compiled by host clang++ with -DKPN_SIMT_EMU and
// this header force-included, and every workgroup is executed by cooperative fibers (one per lane,
// ucontext) that rendezvous at barriers and cross-lane operations.  The result is a shared library
// with the same C-ABI (include/kpnerf.h) operating on host memory; only tests/ load it.  It is NOT
// a fallback: keypointnerf_amd never loads it and fails loudly without the real gfx950 library.
//
// Emulated semantics (gfx950 / CDNA4, per /opt/skills/guides/cdna_hip_programming.md §1-§3):
//   * wavefront = 64 lanes; threadIdx/blockIdx/blockDim/gridDim; __syncthreads(); static __shared__;
//   * __shfl / __shfl_xor / __shfl_up / __shfl_down / __ballot / __any / __all (64 wide);
//   * __builtin_amdgcn_mfma_f32_32x32x2f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
//     D[row=(r&3)+8(r>>2)+4(l>>5)][col=l&31], D = fma(A[i][1],B[1][j], fma(A[i][0],B[0][j], C));
//   * __builtin_amdgcn_mfma_f32_16x16x4f32: A[l&15][k=l>>4], B[k=l>>4][l&15], D[row=4(l>>4)+r][col=l&15];
//   * atomicAdd on global memory (blocks may run on several host threads).
#pragma once
#ifndef KPN_SIMT_EMU
#error "simt.h is only for the -DKPN_SIMT_EMU host build"
#endif
#include <ucontext.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__ __restrict

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct int4 { int x, y, z, w; };
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
typedef void* hipStream_t;

namespace simt {
constexpr int WAVE = 64;
struct Fiber {
    ucontext_t ctx;
    std::vector<char> stack;
    bool done = false;
    unsigned tid = 0;
};
struct WaveShared {
    float a8[WAVE][8], b8[WAVE][8];
    float a[WAVE], b[WAVE];
    uint32_t u[WAVE];
    unsigned arrived = 0, gen = 0;
};
struct Block {
    std::vector<Fiber> fibers;
    std::vector<WaveShared> waves;
    ucontext_t sched;
    unsigned cur = 0, nthreads = 0;
    unsigned bar_arrived = 0, bar_gen = 0;
    std::function<void()> body;
};
extern thread_local Block* g_block;
extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

inline void yield() {
    Block* b = g_block;
    swapcontext(&b->fibers[b->cur].ctx, &b->sched);
}
inline void block_barrier() {
    Block* b = g_block;
    unsigned g = b->bar_gen;
    if (++b->bar_arrived == b->nthreads) { b->bar_arrived = 0; ++b->bar_gen; }
    while (b->bar_gen == g) yield();
}
inline WaveShared& wave_shared() { return g_block->waves[g_block->cur / WAVE]; }
inline unsigned wave_width() {
    Block* b = g_block;
    unsigned w = b->cur / WAVE;
    unsigned n = b->nthreads - w * WAVE;
    return n < (unsigned)WAVE ? n : (unsigned)WAVE;
}
inline void wave_sync() {
    WaveShared& w = wave_shared();
    unsigned g = w.gen, n = wave_width();
    if (++w.arrived == n) { w.arrived = 0; ++w.gen; }
    while (w.gen == g) yield();
}
inline int lane_id() { return (int)(g_block->cur % WAVE); }

void launch(dim3 grid, dim3 block, const std::function<void()>& body);
}  // namespace simt

#define threadIdx (simt::t_threadIdx)
#define blockIdx (simt::t_blockIdx)
#define blockDim (simt::t_blockDim)
#define gridDim (simt::t_gridDim)

static inline void __syncthreads() { simt::block_barrier(); }

// ---- cross-lane ----
static inline float __shfl(float v, int src, int width = 64) {
    auto& w = simt::wave_shared();
    int l = simt::lane_id();
    w.a[l] = v;
    simt::wave_sync();
    int base = l & ~(width - 1);
    float r = w.a[base + (src & (width - 1))];
    simt::wave_sync();
    return r;
}
static inline int __shfl(int v, int src, int width = 64) {
    float f; memcpy(&f, &v, 4); f = __shfl(f, src, width); int r; memcpy(&r, &f, 4); return r;
}
static inline float __shfl_xor(float v, int mask, int width = 64) { return __shfl(v, (simt::lane_id() ^ mask), width); }
static inline int __shfl_xor(int v, int mask, int width = 64) { return __shfl(v, (simt::lane_id() ^ mask), width); }
static inline float __shfl_up(float v, unsigned d, int width = 64) {
    int l = simt::lane_id(); int s = (l & (width - 1)) - (int)d;
    float r = __shfl(v, s < 0 ? l : l - (int)d, width); return r;
}
static inline float __shfl_down(float v, unsigned d, int width = 64) {
    int l = simt::lane_id(); int s = (l & (width - 1)) + (int)d;
    float r = __shfl(v, s >= width ? l : l + (int)d, width); return r;
}
static inline unsigned long long __ballot(int pred) {
    auto& w = simt::wave_shared();
    int l = simt::lane_id();
    w.u[l] = pred ? 1u : 0u;
    simt::wave_sync();
    unsigned long long m = 0;
    unsigned n = simt::wave_width();
    for (unsigned i = 0; i < n; ++i) m |= (unsigned long long)(w.u[i] & 1u) << i;
    simt::wave_sync();
    return m;
}
static inline int __any(int p) { return __ballot(p) != 0ull; }
static inline int __all(int p) {
    unsigned n = simt::wave_width();
    unsigned long long full = n == 64 ? ~0ull : ((1ull << n) - 1);
    return __ballot(p) == full;
}
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return __shfl(v, 0); }

// ---- MFMA ----
typedef float kpn_f32x16 __attribute__((ext_vector_type(16)));
typedef float kpn_f32x4 __attribute__((ext_vector_type(4)));
static inline kpn_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, kpn_f32x16 c, int, int, int) {
    auto& w = simt::wave_shared();
    int l = simt::lane_id();
    w.a[l] = a; w.b[l] = b;
    simt::wave_sync();
    kpn_f32x16 d;
    int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = std::fmaf(w.a[i], w.b[j], c[r]);         // k = 0
        acc = std::fmaf(w.a[i + 32], w.b[j + 32], acc);      // k = 1
        d[r] = acc;
    }
    simt::wave_sync();
    return d;
}
// v_mfma_f32_32x32x16_bf16: A[i=l&31][k=8(l>>5)+e], B[k=8(l>>5)+e][j=l&31], D as the 32x32x2 form; bf16 carried as uint16
typedef uint16_t simt_bf16x8 __attribute__((ext_vector_type(8)));
static inline kpn_f32x16 simt_mfma_f32_32x32x16_bf16(simt_bf16x8 a, simt_bf16x8 b, kpn_f32x16 c) {
    auto& w = simt::wave_shared();
    int l = simt::lane_id();
    for (int e = 0; e < 8; ++e) {
        uint32_t ua = (uint32_t)a[e] << 16, ub = (uint32_t)b[e] << 16;
        memcpy(&w.a8[l][e], &ua, 4); memcpy(&w.b8[l][e], &ub, 4);
    }
    simt::wave_sync();
    kpn_f32x16 d;
    int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int kh = 0; kh < 2; ++kh)
            for (int e = 0; e < 8; ++e) acc = std::fmaf(w.a8[i + 32 * kh][e], w.b8[j + 32 * kh][e], acc);
        d[r] = acc;
    }
    simt::wave_sync();
    return d;
}
// v_mfma_f32_32x32x16_f16: same operand maps; fp16 carried as its 16-bit pattern (subnormal inputs are honoured by the
// hardware: scripts/f16_split_probe.hip)
static inline kpn_f32x16 simt_mfma_f32_32x32x16_f16(simt_bf16x8 a, simt_bf16x8 b, kpn_f32x16 c) {
    auto& w = simt::wave_shared();
    int l = simt::lane_id();
    for (int e = 0; e < 8; ++e) {
        _Float16 ha, hb; uint16_t ua = a[e], ub = b[e];
        memcpy(&ha, &ua, 2); memcpy(&hb, &ub, 2);
        w.a8[l][e] = (float)ha; w.b8[l][e] = (float)hb;
    }
    simt::wave_sync();
    kpn_f32x16 d;
    int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int kh = 0; kh < 2; ++kh)
            for (int e = 0; e < 8; ++e) acc = std::fmaf(w.a8[i + 32 * kh][e], w.b8[j + 32 * kh][e], acc);
        d[r] = acc;
    }
    simt::wave_sync();
    return d;
}
static inline kpn_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, kpn_f32x4 c, int, int, int) {
    auto& w = simt::wave_shared();
    int l = simt::lane_id();
    w.a[l] = a; w.b[l] = b;
    simt::wave_sync();
    kpn_f32x4 d;
    int j = l & 15, g = l >> 4;
    for (int r = 0; r < 4; ++r) {
        int i = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = std::fmaf(w.a[i + 16 * k], w.b[j + 16 * k], acc);
        d[r] = acc;
    }
    simt::wave_sync();
    return d;
}

// ---- atomics / math ----
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
// (fast-math intrinsics such as __expf are wrapped by kpn_common.h: glibc owns those names on the host)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }

// ---- runtime shims used by kpn_api ----
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToDevice 3
#define hipMemcpyDeviceToHost 2
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? 0 : 1; }
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }

#define KPN_LAUNCH(kernel, grid, block, stream, ...) \
    simt::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
