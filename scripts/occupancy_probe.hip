// Probe (gfx950): the rows kernel's instruction mix — one v_mfma_f32_32x32x16_f16 followed by one slice of the activation + two-piece
// split of an operand pair (the three slices of kpn_h2_slice in rotation: 16 instructions, 4 of them transcendental, per 3 MFMAs)
// plus F independent v_fma_f32 (the rest of the kernel's VALU work) — with ONE wave per SIMD (256 threads per CU) and with TWO
// (512 threads): does a second wave turn the first one's dependency stalls into issue slots?  No memory traffic, fixed registers,
// four accumulators in rotation.  Printed: s_memtime ticks per MFMA per wave, and per SIMD (= per wave / waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 scripts/occupancy_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>

#define MFMA(acc) "v_mfma_f32_32x32x16_f16 " acc ", v[8:11], v[12:15], " acc "\n\t"
// slice 0: read two accumulator elements, 2^-|u|, max(u, 0)   slice 1: 1 + e, log2, max   slice 2: add, hi pieces, residuals, lo pieces
#define Q0 "v_accvgpr_read_b32 v16, a100\n\tv_accvgpr_read_b32 v17, a101\n\tv_exp_f32 v18, -|v16|\n\tv_exp_f32 v19, -|v17|\n\tv_max_f32 v16, 0, v16\n\t"
#define Q1 "v_add_f32 v18, 1.0, v18\n\tv_add_f32 v19, 1.0, v19\n\tv_log_f32 v18, v18\n\tv_log_f32 v19, v19\n\tv_max_f32 v17, 0, v17\n\t"
#define Q2 "v_add_f32 v16, v16, v18\n\tv_add_f32 v17, v17, v19\n\tv_cvt_pk_f16_f32 v20, v16, v17\n\tv_fma_mix_f32 v16, v20, -1.0, v16 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 v17, v20, -1.0, v17 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_cvt_pk_f16_f32 v21, v16, v17\n\t"
#define X0 ""
#define X1 "v_fma_f32 v24, v24, v6, v7\n\t"
#define X2 X1 "v_fma_f32 v25, v25, v6, v7\n\t"
#define X3 X2 "v_fma_f32 v26, v26, v6, v7\n\t"
#define X4 X3 "v_fma_f32 v27, v27, v6, v7\n\t"
#define BODYM MFMA("a[0:15]") MFMA("a[16:31]") MFMA("a[32:47]") MFMA("a[48:63]") MFMA("a[0:15]") MFMA("a[16:31]") MFMA("a[32:47]") MFMA("a[48:63]") MFMA("a[0:15]") MFMA("a[16:31]") MFMA("a[32:47]") MFMA("a[48:63]")
#define BODY(X) MFMA("a[0:15]") Q0 X MFMA("a[16:31]") Q1 X MFMA("a[32:47]") Q2 X MFMA("a[48:63]") Q0 X MFMA("a[0:15]") Q1 X MFMA("a[16:31]") Q2 X \
                MFMA("a[32:47]") Q0 X MFMA("a[48:63]") Q1 X MFMA("a[0:15]") Q2 X MFMA("a[16:31]") Q0 X MFMA("a[32:47]") Q1 X MFMA("a[48:63]") Q2 X
#define CLOB "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v24", "v25", "v26", "v27", \
             "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", \
             "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", \
             "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a100", "a101"
#define KERNEL(NAME, X)                                                                                        \
    __global__ __launch_bounds__(512) void NAME(float* out, long long* cycles, int slot) {                     \
        long long t0, t1;                                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));                                        \
        for (int it = 0; it < 256; ++it) asm volatile(BODY(X) BODY(X) ::: CLOB);                               \
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));                \
        if (threadIdx.x == 0 && blockIdx.x == 0) cycles[slot] = t1 - t0;                                       \
        if (out) out[threadIdx.x] = 0.f;                                                                       \
    }
KERNEL(k0, X0) KERNEL(k1, X1) KERNEL(k2, X2) KERNEL(k3, X3) KERNEL(k4, X4)
__global__ __launch_bounds__(512) void kmf(float* out, long long* cycles, int slot) {      // MFMAs only: the matrix pipe's own rate in ticks
    long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < 256; ++it) asm volatile(BODYM BODYM ::: CLOB);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[slot] = t1 - t0;
    if (out) out[threadIdx.x] = 0.f;
}

int main() {
    long long* cyc; hipMalloc(&cyc, 64 * 8); hipMemset(cyc, 0, 64 * 8);
    typedef void (*kern)(float*, long long*, int);
    kern ks[5] = {k0, k1, k2, k3, k4};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512}) {     // calibration: MFMAs only, ticks and wall time
        hipLaunchKernelGGL(kmf, dim3(256), dim3(threads), 0, 0, (float*)nullptr, cyc, 40); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kmf, dim3(256), dim3(threads), 0, 0, (float*)nullptr, cyc, 40); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); long long h; hipMemcpy(&h, cyc + 40, 8, hipMemcpyDeviceToHost);
        printf("MFMAs only, %d wave(s)/SIMD: %.1f ticks per MFMA per wave, kernel %.1f us -> %.2f ticks per ns; %.0f TFLOP/s\n", threads / 256, (double)h / (256.0 * 24.0), ms * 1e3,
               (double)h / (ms * 1e6), 256.0 * (threads / 64) * 256.0 * 24.0 * 32768.0 / (ms * 1e-3) / 1e12);
    }
    printf("ticks per MFMA: slices of the rows kernel (4 VALU + 1.33 transcendental per MFMA) + F independent v_fma_f32 per MFMA; 256 CUs busy\n");
    printf("%-28s %10s %10s %10s %10s %10s\n", "F =", "0", "1", "2", "3", "4");
    for (int threads : {256, 512}) {
        double per[5], wall[5];
        for (int i = 0; i < 5; ++i) {
            hipLaunchKernelGGL(ks[i], dim3(256), dim3(threads), 0, 0, (float*)nullptr, cyc, i); hipDeviceSynchronize();
            hipEventRecord(e0); hipLaunchKernelGGL(ks[i], dim3(256), dim3(threads), 0, 0, (float*)nullptr, cyc, i); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h; hipMemcpy(&h, cyc + i, 8, hipMemcpyDeviceToHost);
            per[i] = (double)h / (256.0 * 24.0);
            wall[i] = ms * 1e6 / (256.0 * 24.0);
        }
        printf("%d wave(s)/SIMD, per wave     ", threads / 256); for (double p : per) printf(" %10.1f", p); printf("\n");
        printf("%d wave(s)/SIMD, per SIMD     ", threads / 256); for (double p : per) printf(" %10.1f", p / (threads / 256)); printf("\n");
        printf("%d wave(s)/SIMD, ns per MFMA and SIMD (events, incl. launch)", threads / 256); for (double p : wall) printf(" %8.2f", p / (threads / 256)); printf("\n");
    }
    return 0;
}
