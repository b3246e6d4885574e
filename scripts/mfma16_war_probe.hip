// Probe: how long after issuing v_mfma_f32_32x32x16_bf16 may its B (and A) operand registers be overwritten?
// Each iteration issues ONE MFMA from an asm statement followed by `s_nop N`; the compiler-visible statement after it
// overwrites the operand registers with a NaN pattern (hipcc does not know an MFMA sits in the asm string, so it pads
// nothing: the distance is N + the few cycles of its own instructions).  If the MFMA has not finished READING the operand
// when the overwrite lands, NaNs appear in the accumulator.  Run at 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma16_war_probe.hip -o /tmp/war && /tmp/war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int N, int WHICH>   // WHICH: 0 = clobber B, 1 = clobber A
__global__ __launch_bounds__(256, 2) void k_probe(int iters, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    bf16x8 a, b, fa, fb;
    f32x16 f0, f1;
    for (int r = 0; r < 16; ++r) { f0[r] = 0.0f; f1[r] = 0.0f; }
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)0.001f; fb[i] = (__bf16)0.002f; }
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (float)((lane + i) % 7 + 1)); }
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < 8; ++i) { b[i] = (__bf16)(0.02f * (float)((lane * 3 + i + it) % 5 + 1)); a[i] = (__bf16)(0.01f * (float)((lane + i + it) % 7 + 1)); }
        // a and b are read-write operands: the compiler keeps them in the same registers after the statement, so the OR
        // below is emitted in place (checked in the ISA: v_or_b32 on the MFMA's own source registers)
#ifdef FILL   // keep the matrix pipe saturated (both waves of a SIMD stream MFMAs): the probed MFMA may have to queue
        asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_bf16 %3, %5, %6, %3\n\tv_mfma_f32_32x32x16_bf16 %4, %5, %6, %4\n\t"
                     "v_mfma_f32_32x32x16_bf16 %3, %5, %6, %3\n\tv_mfma_f32_32x32x16_bf16 %4, %5, %6, %4\n\t"
                     "v_mfma_f32_32x32x16_bf16 %3, %5, %6, %3\n\tv_mfma_f32_32x32x16_bf16 %4, %5, %6, %4\n\t"
                     "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop %7"
                     : "+v"(acc), "+v"(a), "+v"(b), "+v"(f0), "+v"(f1) : "v"(fa), "v"(fb), "n"(N));
        asm volatile("" : "+v"(f1));
#else
        asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop %3" : "+v"(acc), "+v"(a), "+v"(b) : "n"(N));
#endif
        u32x4 g = __builtin_bit_cast(u32x4, WHICH == 0 ? b : a) | 0x7fc07fc0u;
        if (WHICH == 0) b = __builtin_bit_cast(bf16x8, g); else a = __builtin_bit_cast(bf16x8, g);
        asm volatile("" : "+v"(a), "+v"(b));     // the overwrite is not dead, and stays where it is
#ifndef FILL
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc));   // let the MFMA drain before the next round
#else
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
#endif
    }
    float* o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    for (int r = 0; r < 16; ++r) o[r] = acc[r] + 0.0f * (f0[r] + f1[r]);
}
template <int N, int WHICH>
void run(float* d, std::vector<float>& h, int blocks) {
    k_probe<N, WHICH><<<blocks, 256>>>(200, d);
    hipMemcpy(h.data(), d, (size_t)blocks * 256 * 16 * 4, hipMemcpyDeviceToHost);
    long nan = 0, nan_hi = 0;
    for (size_t i = 0; i < (size_t)blocks * 256 * 16; ++i) if (h[i] != h[i]) { ++nan; if (((i / 16) & 31) >= 16) ++nan_hi; }
    printf("  clobber %s after s_nop %2d, %4d workgroups: %ld NaN accumulator values (%ld of them in columns 16..31)\n", WHICH ? "A" : "B", N, blocks, nan, nan_hi);
}
template <int WHICH>
void sweep(float* d, std::vector<float>& h, int blocks) {
    run<0, WHICH>(d, h, blocks); run<1, WHICH>(d, h, blocks); run<2, WHICH>(d, h, blocks); run<3, WHICH>(d, h, blocks); run<4, WHICH>(d, h, blocks);
    run<6, WHICH>(d, h, blocks); run<8, WHICH>(d, h, blocks); run<10, WHICH>(d, h, blocks); run<12, WHICH>(d, h, blocks); run<15, WHICH>(d, h, blocks);
}
int main() {
    float* d; hipMalloc(&d, (size_t)1024 * 256 * 16 * 4);
    std::vector<float> h((size_t)1024 * 256 * 16);
    for (int blocks : {256, 512}) {
        printf("%d workgroups of 4 waves (%s per SIMD):\n", blocks, blocks <= 256 ? "1 wave" : "2 waves");
        sweep<0>(d, h, blocks); sweep<1>(d, h, blocks);
    }
    return 0;
}
