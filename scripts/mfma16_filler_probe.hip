// Probe (one wavefront per SIMD, gfx950): how many single-issue instructions hide under one v_mfma_f32_32x32x16_bf16,
// by filler kind.  Stream = 4 independent accumulators (AGPRs) in rotation, F fillers after every MFMA, all inline asm
// with fixed registers so that nothing is reordered or padded by the compiler.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma16_filler_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>

#define MFMA(acc) "v_mfma_f32_32x32x16_bf16 " acc ", v[8:11], v[12:15], " acc "\n\t"
template <int KIND> struct Fill;
template <> struct Fill<0> { static constexpr const char* name = "v_fma_f32 (independent)"; };
template <> struct Fill<1> { static constexpr const char* name = "v_exp_f32"; };
template <> struct Fill<2> { static constexpr const char* name = "v_cvt_pk_bf16_f32"; };
template <> struct Fill<3> { static constexpr const char* name = "v_accvgpr_read_b32 (idle AGPR)"; };
template <> struct Fill<4> { static constexpr const char* name = "v_and_b32"; };
template <> struct Fill<5> { static constexpr const char* name = "v_fma_f32 (one dependent chain)"; };

#define F0(i) "v_fma_f32 v" #i ", v" #i ", v6, v7\n\t"
#define F1(i) "v_exp_f32 v" #i ", v" #i "\n\t"
#define F2(i) "v_cvt_pk_bf16_f32 v" #i ", v6, v7\n\t"
#define F3(i) "v_accvgpr_read_b32 v" #i ", a100\n\t"
#define F4(i) "v_and_b32 v" #i ", 0xffff0000, v6\n\t"
#define F5(i) "v_fma_f32 v16, v16, v6, v7\n\t"

#define FILL_1(X) X(16)
#define FILL_2(X) X(16) X(17)
#define FILL_3(X) X(16) X(17) X(18)
#define FILL_4(X) X(16) X(17) X(18) X(19)
#define FILL_5(X) X(16) X(17) X(18) X(19) X(20)
#define FILL_6(X) X(16) X(17) X(18) X(19) X(20) X(21)
#define FILL_8(X) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23)
#define FILL_0(X) ""

#define BODY(FILL) MFMA("a[0:15]") FILL MFMA("a[16:31]") FILL MFMA("a[32:47]") FILL MFMA("a[48:63]") FILL

#define KERNEL(NAME, FILL)                                                                                                  \
    __global__ __launch_bounds__(256, 1) void NAME(float* out, long long* cycles, int slot) {                               \
        long long t0, t1;                                                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));                                                     \
        for (int it = 0; it < 256; ++it)                                                                                    \
            asm volatile(BODY(FILL) BODY(FILL) BODY(FILL) BODY(FILL) ::: "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17",  \
                         "v18", "v19", "v20", "v21", "v22", "v23", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", \
                         "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31",     \
                         "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49",     \
                         "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a100");                        \
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));                             \
        if (threadIdx.x == 0 && blockIdx.x == 0) cycles[slot] = t1 - t0;                                                    \
        if (out) out[threadIdx.x] = 0.f;                                                                                    \
    }

#define ROW(K, X)                                                                                      \
    KERNEL(k##K##_0, FILL_0(X)) KERNEL(k##K##_2, FILL_2(X)) KERNEL(k##K##_4, FILL_4(X)) KERNEL(k##K##_5, FILL_5(X)) \
    KERNEL(k##K##_6, FILL_6(X)) KERNEL(k##K##_8, FILL_8(X))
ROW(0, F0) ROW(1, F1) ROW(2, F2) ROW(3, F3) ROW(4, F4) ROW(5, F5)

int main() {
    long long* cyc; hipMalloc(&cyc, 64 * 8); hipMemset(cyc, 0, 64 * 8);
    typedef void (*kern)(float*, long long*, int);
    kern ks[6][6] = {{k0_0, k0_2, k0_4, k0_5, k0_6, k0_8}, {k1_0, k1_2, k1_4, k1_5, k1_6, k1_8}, {k2_0, k2_2, k2_4, k2_5, k2_6, k2_8},
                     {k3_0, k3_2, k3_4, k3_5, k3_6, k3_8}, {k4_0, k4_2, k4_4, k4_5, k4_6, k4_8}, {k5_0, k5_2, k5_4, k5_5, k5_6, k5_8}};
    const char* names[6] = {Fill<0>::name, Fill<1>::name, Fill<2>::name, Fill<3>::name, Fill<4>::name, Fill<5>::name};
    const int fs[6] = {0, 2, 4, 5, 6, 8};
    printf("s_memtime ticks per MFMA (16 MFMAs x 256 iterations per measurement), one wave per SIMD, all 256 CUs busy\n%-36s", "fillers per MFMA:");
    for (int f : fs) printf("%8d", f);
    printf("\n");
    for (int k = 0; k < 6; ++k) {
        printf("%-36s", names[k]);
        for (int i = 0; i < 6; ++i) {
            for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(ks[k][i], dim3(256), dim3(256), 0, 0, (float*)nullptr, cyc, k * 6 + i); hipDeviceSynchronize(); }
            long long h; hipMemcpy(&h, cyc + k * 6 + i, 8, hipMemcpyDeviceToHost);
            printf("%8.1f", (double)h / (256.0 * 16.0));
        }
        printf("\n");
    }
    return 0;
}
