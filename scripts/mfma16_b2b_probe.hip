// Probe (gfx950): back-to-back DEPENDENT v_mfma_f32_32x32x16_bf16 on ONE accumulator (SrcC = the previous vDst, zero wait
// states: the case hipcc treats as hardware-forwarded) while a second wavefront on the same SIMD issues MFMAs of its own.
// All operands are 1.0, so every MFMA adds exactly 16 to every accumulator element and the expected result is exact.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma16_b2b_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// CHAIN dependent MFMAs back to back, then GAP independent VALU instructions, repeated; two accumulators used in turn like
// mfma_half of k_geo_rows_h (six on one, six on the other)
template <int WAVES, int GAPNOPS>
__global__ __launch_bounds__(256, WAVES) void k(int iters, int* bad, int* badcols) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)1.0f; b[i] = (__bf16)1.0f; }
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    float filler = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
            if (GAPNOPS == 1) asm volatile("s_nop 1");
            if (GAPNOPS == 2) asm volatile("s_nop 7\n\ts_nop 3");
        }
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(filler));
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
            if (GAPNOPS == 1) asm volatile("s_nop 1");
            if (GAPNOPS == 2) asm volatile("s_nop 7\n\ts_nop 3");
        }
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(filler));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const float expect = 16.0f * 6.0f * iters;
    int nb = 0;
    for (int i = 0; i < 16; ++i) nb += (acc0[i] != expect) + (acc1[i] != expect);
    if (nb) { atomicAdd(bad, nb); atomicAdd(badcols + ((threadIdx.x & 31) >> 4), 1); }
    if (filler == 12345.678f) bad[0] = -1;
}

template <int WAVES, int GAPNOPS>
void run(const char* name, int blocks) {
    int *bad, *cols; hipMalloc(&bad, 4); hipMalloc(&cols, 8);
    long total = 0, tb = 0; int c[2] = {0, 0};
    for (int rep = 0; rep < 20; ++rep) {
        hipMemset(bad, 0, 4); hipMemset(cols, 0, 8);
        hipLaunchKernelGGL((k<WAVES, GAPNOPS>), dim3(blocks), dim3(256), 0, 0, 4000, bad, cols);
        hipDeviceSynchronize();
        int hb, hc[2]; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hc, cols, 8, hipMemcpyDeviceToHost);
        tb += hb; c[0] += hc[0]; c[1] += hc[1]; total += (long)blocks * 4 * 4000 * 12;
    }
    printf("%-64s %ld MFMAs: %ld wrong accumulator elements (lanes with columns 0-15: %d, columns 16-31: %d)\n", name, total, tb, c[0], c[1]);
    hipFree(bad); hipFree(cols);
}

int main() {
    run<1, 0>("one wave per SIMD, dependent MFMAs back to back", 256);
    run<2, 0>("two waves per SIMD, dependent MFMAs back to back", 512);
    run<2, 1>("two waves per SIMD, s_nop 1 between dependent MFMAs", 512);
    run<2, 2>("two waves per SIMD, 12 wait states between dependent MFMAs", 512);
    return 0;
}
