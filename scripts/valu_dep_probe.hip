// Probe: issue cost of DEPENDENT vs INDEPENDENT VALU instructions with one wavefront per SIMD on gfx950 (no other wave to
// fill the gaps), plain and transcendental, and the same beside a stream of v_mfma_f32_32x32x16_bf16.
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_dep_probe.hip -o /tmp/valu_dep_probe && /tmp/valu_dep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
__global__ __launch_bounds__(256, 1) void k(int mode, float* out, long long* cycles) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = a + 1.0f, c = a + 2.0f, d = a + 3.0f;
    const float m = 1.0001f, q = 0.5f;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    bf16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(0.001f * (threadIdx.x + i)); bv[i] = (__bf16)(0.002f * i); }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    long long t0 = clock64();
    for (int it = 0; it < 64; ++it) {
        if (mode == 0) {
#pragma unroll
            for (int i = 0; i < REP; ++i) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(q)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(q));
                                            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(q)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(q)); }
        } else if (mode == 1) {
#pragma unroll
            for (int i = 0; i < REP; ++i) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(q)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(b) : "v"(m), "v"(q));
                                            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c) : "v"(m), "v"(q)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(m), "v"(q)); }
        } else if (mode == 2) {   // dependent exp/log chain
#pragma unroll
            for (int i = 0; i < REP; ++i) { asm volatile("v_exp_f32 %0, %0" : "+v"(a)); asm volatile("v_log_f32 %0, %0" : "+v"(a));
                                            asm volatile("v_exp_f32 %0, %0" : "+v"(a)); asm volatile("v_log_f32 %0, %0" : "+v"(a)); }
        } else if (mode == 3) {   // four independent exp/log chains
#pragma unroll
            for (int i = 0; i < REP; ++i) { asm volatile("v_exp_f32 %0, %0" : "+v"(a)); asm volatile("v_exp_f32 %0, %0" : "+v"(b));
                                            asm volatile("v_log_f32 %0, %0" : "+v"(c)); asm volatile("v_log_f32 %0, %0" : "+v"(d)); }
        } else if (mode == 4) {   // the softplus chain as compiled: mul, exp, add, log, mul — dependent
#pragma unroll
            for (int i = 0; i < REP; ++i) { asm volatile("v_mul_f32 %0, %0, %1\n\tv_exp_f32 %0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_log_f32 %0, %0\n\tv_mul_f32 %0, %0, %1" : "+v"(a) : "v"(q)); }
        } else if (mode == 5) {   // two such chains interleaved instruction by instruction
#pragma unroll
            for (int i = 0; i < REP / 2; ++i) { asm volatile("v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %1, 1.0, %1\n\t"
                                                             "v_log_f32 %0, %0\n\tv_log_f32 %1, %1\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(q)); }
        } else if (mode >= 6) {   // four independent accumulators, F plain VALU fillers after every MFMA (F = mode - 6)
#define FILL asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(b) : "v"(m), "v"(q)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c) : "v"(m), "v"(q));
#pragma unroll
            for (int i = 0; i < REP / 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[j], 0, 0, 0);
                    if (mode >= 8) { FILL } if (mode >= 10) { FILL } if (mode >= 12) { FILL } if (mode >= 14) { FILL }
                }
            }
        }
    }
    long long t1 = clock64();
    a += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[mode] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 128);
    hipMemset(cyc, 0, 128);
    const char* names[16] = {"dependent v_fma chain", "4 independent v_fma chains", "dependent exp/log chain", "4 independent exp/log chains",
                            "softplus chain (mul exp add log mul), dependent", "two softplus chains interleaved", "MFMA 32x32x16 bf16, 0 fillers", "", "MFMA + 2 VALU", "", "MFMA + 4 VALU", "", "MFMA + 6 VALU", "", "MFMA + 8 VALU", ""};
    const int per_iter[16] = {4 * REP, 4 * REP, 4 * REP, 4 * REP, 5 * REP, 5 * REP, REP, 0, REP, 0, REP, 0, REP, 0, REP, 0};
    for (int mode = 0; mode < 15; ++mode) {
        if (!per_iter[mode]) continue;
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, mode, out, cyc);
        hipDeviceSynchronize();
    }
    long long h[16]; hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
    for (int mode = 0; mode < 15; ++mode) if (per_iter[mode]) printf("%-52s %6.2f clock64 ticks per %s\n", names[mode], (double)h[mode] / (64.0 * per_iter[mode]), mode >= 6 ? "MFMA (+ its fillers)" : "instruction");
    return 0;
}
