// Probe: wait states needed between v_mfma_f32_32x32x16_bf16 and a VALU instruction READING its result, with one and two
// waves per SIMD, the matrix pipe kept busy by both waves (FILL filler MFMAs on other accumulators issued just before).
// hipcc pads this pair itself when it sees both instructions (s_nop 9..10 in the ISA of scripts/repro_mfma16_hazard.hip);
// here the pair sits in one asm statement so that the distance is exactly K + 1 wait states.
//   hipcc --offload-arch=gfx950 -O3 [-DFILL=n] scripts/mfma16_result_probe.hip -o /tmp/res && /tmp/res
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#ifndef FILL
#define FILL 4
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define XSTR(s) STR(s)
#define STR(s) #s

template <int K>
__global__ __launch_bounds__(256, 2) void k_probe(int iters, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    f32x16 f0, f1;
    bf16x8 a, b;
    for (int r = 0; r < 16; ++r) { f0[r] = 0.0f; f1[r] = 0.0f; }
    float sum = 0.0f;
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (float)((lane + i + it) % 7 + 1)); b[i] = (__bf16)(0.02f * (float)((lane * 3 + i + it) % 5 + 1)); }
        const float init = 0.001f * (float)(it % 13);
        float s0, s1, s2, s3, e0, e1, e2, e3;
        // the probed accumulator is v[64:79], named explicitly (an asm operand cannot address one register of a tuple)
        asm volatile(
            ".irp r,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79\n\tv_mov_b32 v\\r, %10\n\t.endr\n\t"
            "s_nop 7\n\t"
            ".rept " XSTR(FILL) "\n\tv_mfma_f32_32x32x16_bf16 %0, %11, %12, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %11, %12, %1\n\t.endr\n\t"
            "v_mfma_f32_32x32x16_bf16 v[64:79], %11, %12, v[64:79]\n\t"
            "s_nop %13\n\t"
            "v_mov_b32 %2, v79\n\tv_mov_b32 %3, v64\n\tv_mov_b32 %4, v71\n\tv_mov_b32 %5, v72\n\t"   // early reads
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %6, v79\n\tv_mov_b32 %7, v64\n\tv_mov_b32 %8, v71\n\tv_mov_b32 %9, v72"          // the finished values
            : "+v"(f0), "+v"(f1), "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
            : "v"(init), "v"(a), "v"(b), "n"(K)
            : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
        sum += (s0 != e0 ? 1.0f : 0.0f) + (s1 != e1 ? 1.0f : 0.0f) + (s2 != e2 ? 1.0f : 0.0f) + (s3 != e3 ? 1.0f : 0.0f);
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum + 0.0f * (f0[0] + f1[0]);
}
template <int K>
void run(float* d, std::vector<float>& h, int blocks) {
    k_probe<K><<<blocks, 256>>>(200, d);
    (void)hipMemcpy(h.data(), d, (size_t)blocks * 256 * 4, hipMemcpyDeviceToHost);
    double bad = 0, bad_hi = 0;
    for (size_t i = 0; i < (size_t)blocks * 256; ++i) { bad += h[i]; if ((i & 31) >= 16) bad_hi += h[i]; }
    printf("  s_nop %2d after the MFMA: %.0f stale result registers read (%.0f of them by lanes with column >= 16)\n", K, bad, bad_hi);
}
int main() {
    float* d; (void)hipMalloc(&d, (size_t)512 * 256 * 4);
    std::vector<float> h((size_t)512 * 256);
    for (int blocks : {256, 512}) {
        printf("%d workgroups of 4 waves (%s per SIMD), %d + %d filler MFMAs before the probed one:\n", blocks, blocks <= 256 ? "1 wave" : "2 waves", FILL, FILL);
        run<0>(d, h, blocks); run<2>(d, h, blocks); run<4>(d, h, blocks); run<6>(d, h, blocks); run<8>(d, h, blocks); run<9>(d, h, blocks);
        run<10>(d, h, blocks); run<11>(d, h, blocks); run<12>(d, h, blocks); run<13>(d, h, blocks); run<15>(d, h, blocks);
    }
    return 0;
}
