// Probe: dependent v_mfma_f32_32x32x16_bf16 chains on TWO accumulators, issued interleaved (a0, a1, a0, a1, ... — each
// MFMA's C operand is the result of the MFMA two instructions earlier) versus one chain after the other.  Both orders
// compute the same two sums bit for bit; the asm-free build of k_geo_rows_h interleaves like this (hipcc's choice).
// Compared at 1 and 2 waves per SIMD; different operands per step so that a skipped or stale accumulate shows.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma16_chain_probe.hip -o /tmp/chain && /tmp/chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define M(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)

template <int MODE>   // 0: sequential chains, 1: interleaved (sched_barrier keeps the written order), 2: interleaved with VALU between
__global__ __launch_bounds__(256, 2) void k_probe(int iters, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; }
    bf16x8 w[6], x[3];
    float extra = 0.0f;
    for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < 6; ++k) for (int i = 0; i < 8; ++i) w[k][i] = (__bf16)(0.01f * (float)((lane + i + 3 * k + it) % 7 - 3));
        for (int k = 0; k < 3; ++k) for (int i = 0; i < 8; ++i) x[k][i] = (__bf16)(0.02f * (float)((lane * 3 + i + k + it) % 5 - 2));
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) {
            M(a0, w[0], x[0]); M(a0, w[0], x[1]); M(a0, w[1], x[0]); M(a0, w[1], x[1]); M(a0, w[0], x[2]); M(a0, w[2], x[0]);
            M(a1, w[3], x[0]); M(a1, w[3], x[1]); M(a1, w[4], x[0]); M(a1, w[4], x[1]); M(a1, w[3], x[2]); M(a1, w[5], x[0]);
        } else {
#define STEP(wa, xa, wb, xb) M(a0, wa, xa); __builtin_amdgcn_sched_barrier(0); if (MODE == 2) { extra = extra * 1.0001f + 0.5f; __builtin_amdgcn_sched_barrier(0); } \
                             M(a1, wb, xb); __builtin_amdgcn_sched_barrier(0);
            STEP(w[0], x[0], w[3], x[0]) STEP(w[0], x[1], w[3], x[1]) STEP(w[1], x[0], w[4], x[0])
            STEP(w[1], x[1], w[4], x[1]) STEP(w[0], x[2], w[3], x[2]) STEP(w[2], x[0], w[5], x[0])
        }
        __builtin_amdgcn_sched_barrier(0);
        for (int r = 0; r < 16; ++r) { a0[r] *= 0.5f; a1[r] *= 0.5f; }
    }
    float* o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 32;
    for (int r = 0; r < 16; ++r) { o[r] = a0[r]; o[16 + r] = a1[r] + 0.0f * extra; }
}
int main() {
    float* d; (void)hipMalloc(&d, (size_t)512 * 256 * 32 * 4);
    const size_t per = 256 * 32;
    std::vector<float> ref(per), h((size_t)512 * per);
    k_probe<0><<<256, 256>>>(200, d);
    (void)hipMemcpy(ref.data(), d, per * 4, hipMemcpyDeviceToHost);
    printf("reference sample %g %g\n", ref[5 * 32 + 3], ref[70 * 32 + 20]);
    for (int mode = 0; mode < 3; ++mode)
        for (int blocks : {256, 512}) {
            long bad = 0, bad_hi = 0;
            for (int rep = 0; rep < 50; ++rep) {
                if (mode == 0) k_probe<0><<<blocks, 256>>>(200, d); else if (mode == 1) k_probe<1><<<blocks, 256>>>(200, d); else k_probe<2><<<blocks, 256>>>(200, d);
                (void)hipMemcpy(h.data(), d, (size_t)blocks * per * 4, hipMemcpyDeviceToHost);
                for (int g = 0; g < blocks; ++g) for (size_t i = 0; i < per; ++i) if (memcmp(&h[g * per + i], &ref[i], 4)) { ++bad; if (((i / 32) & 31) >= 16) ++bad_hi; }
            }
            printf("mode %d (%s), %d workgroups (%s per SIMD): %ld wrong values (%ld in columns 16..31) in 50 launches\n", mode,
                   mode == 0 ? "one chain after the other" : mode == 1 ? "interleaved" : "interleaved + VALU between", blocks, blocks <= 256 ? "1 wave" : "2 waves", bad, bad_hi);
        }
    return 0;
}
