// Microbenchmark: v_mfma_f32_32x32x2_f32 issue rate vs the number of INDEPENDENT accumulator chains per wave
// (NACC = 1: every MFMA depends on the previous one through SrcC) and waves per SIMD.  Decides whether the NOB = 1
// layers of k_fuse_color (one accumulator block) are latency-limited.  hipcc --offload-arch=gfx950 -O3 mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k_chain(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16 / NACC; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-7f;
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024}) {
        const int iters = 20000 * (1024 / blocks);
        k_chain<NACC><<<blocks, 256>>>(d, 100, 1.f, 1e-3f);
        hipDeviceSynchronize();
        hipEventRecord(e0); k_chain<NACC><<<blocks, 256>>>(d, iters, 1.f, 1e-3f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 2;
        printf("NACC %d, waves/SIMD %.0f: %.1f TFLOP/s\n", NACC, blocks / 256.0, flop / ms / 1e9);
    }
}
int main() {
    float* d; hipMalloc(&d, 2048 * 256 * 4);
    run<1>(d); run<2>(d); run<4>(d);
    return 0;
}
