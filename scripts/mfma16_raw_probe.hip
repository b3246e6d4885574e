// Probe: wait states needed between the VALU instruction that WRITES a v_mfma_f32_32x32x16_bf16 source operand and the
// MFMA, with one and with two waves per SIMD, while both waves keep the VALU and the transcendental unit busy.
// Per iteration (all inside one asm statement so that the distance is exact):
//     v_exp_f32 x4 (the SIMD's other wave does the same: contention) ; v_cvt_pk_bf16_f32 B.w <- f(iteration) ; s_nop K ; v_mfma
// The accumulator is compared with the value computed with a long wait (K = 15 plus extra nops).
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma16_raw_probe.hip -o /tmp/raw && /tmp/raw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int SAFE>
__global__ __launch_bounds__(256, 2) void k_probe(int iters, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (float)((lane + i) % 7 + 1)); b[i] = (__bf16)0.0f; }
    float t0 = 0.001f * lane, t1 = 0.002f * lane, t2 = 0.5f, t3 = 0.25f;
    for (int it = 0; it < iters; ++it) {
        float x0 = 0.125f * (float)((lane * 5 + it) % 9 + 1), x1 = 0.0625f * (float)((lane + 3 * it) % 11 + 1);
        // b's 4 registers are all rewritten from x0, x1 (two cvt each so that every register is fresh every iteration)
        asm volatile(
            "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_log_f32 %6, %6\n\tv_log_f32 %7, %7\n\t"
            // B = v[40:43], named explicitly (an asm operand cannot address one register of a 128-bit tuple)
            "v_cvt_pk_bf16_f32 v40, %2, %3\n\tv_cvt_pk_bf16_f32 v41, %3, %2\n\tv_cvt_pk_bf16_f32 v42, %2, %2\n\t"
            "v_cvt_pk_bf16_f32 v43, %3, %3\n\t"         // the last writer of the operand, then K + 1 wait states
            "s_nop %8\n\t"
            "v_mfma_f32_32x32x16_bf16 %0, %9, v[40:43], %0\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
            : "+v"(acc), "+v"(b), "+v"(x0), "+v"(x1), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3)
            : "n"(SAFE ? 15 : K), "v"(a) : "v40", "v41", "v42", "v43");
        if (SAFE) asm volatile("s_nop 15" ::);
        t0 = t0 * 0.5f; t1 = t1 * 0.25f; t2 = t2 + 1.5f; t3 = t3 + 2.5f;
    }
    float* o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    for (int r = 0; r < 16; ++r) o[r] = acc[r] + 0.0f * (t0 + t1 + t2 + t3);
}
template <int K>
void run(float* d, std::vector<float>& ref, std::vector<float>& h, int blocks) {
    const size_t n = (size_t)blocks * 256 * 16;
    k_probe<K, 1><<<blocks, 256>>>(300, d);
    (void)hipMemcpy(ref.data(), d, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad_hi = 0, launches_bad = 0;
    for (int rep = 0; rep < 20; ++rep) {
        k_probe<K, 0><<<blocks, 256>>>(300, d);
        (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        long b = 0;
        for (size_t i = 0; i < n; ++i) if (memcmp(&h[i], &ref[i], 4)) { ++b; if (((i / 16) & 31) >= 16) ++bad_hi; }
        bad += b; launches_bad += b != 0;
    }
    printf("  s_nop %2d between the cvt and the MFMA, %4d workgroups: %ld wrong accumulator values (%ld in columns 16..31) in %ld of 20 launches\n",
           K, blocks, bad, bad_hi, launches_bad);
}
int main() {
    float* d; (void)hipMalloc(&d, (size_t)512 * 256 * 16 * 4);
    std::vector<float> ref((size_t)512 * 256 * 16), h((size_t)512 * 256 * 16);
    for (int blocks : {256, 512}) {
        printf("%d workgroups of 4 waves (%s per SIMD):\n", blocks, blocks <= 256 ? "1 wave" : "2 waves");
        run<0>(d, ref, h, blocks); run<1>(d, ref, h, blocks); run<2>(d, ref, h, blocks); run<3>(d, ref, h, blocks); run<4>(d, ref, h, blocks);
        run<6>(d, ref, h, blocks); run<8>(d, ref, h, blocks);
    }
    return 0;
}
