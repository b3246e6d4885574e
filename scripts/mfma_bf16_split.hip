// Microbenchmark for the "split bf16" option of k_geo_rows: fp32 activations are split on the
// fly into bf16 pieces (hi + lo, or hi + mid + lo) and multiplied with pre-split weights on
// v_mfma_f32_32x32x16_bf16; NPROD MFMAs stand for one fp32 product term set:
//   NPIECE 2, NPROD 3: hi*hi + hi*lo + lo*hi          (~2^-16 relative)
//   NPIECE 2, NPROD 4: + lo*lo                         (~2^-17, representation-limited)
//   NPIECE 3, NPROD 6: hh + hm + mh + hl + lh + mm     (~2^-24, fp32-class)
// Reports fp32-equivalent TFLOP/s (2*32*32*16 flop per product SET) including the split's VALU work, next to the
// 157.3 TFLOP/s fp32-MFMA peak.  hipcc --offload-arch=gfx950 -O3 mfma_bf16_split.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NPIECE>
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 (&p)[NPIECE]) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = x[i];
#pragma unroll
    for (int k = 0; k < NPIECE; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const __bf16 h = (__bf16)r[i];
            p[k][i] = h;
            r[i] -= (float)h;
        }
}

template <int NPIECE, int NPROD, int NOB>
__global__ __launch_bounds__(256, 2) void k_split(float* out, int iters, float a0) {
    f32x16 acc[NOB];
    for (int i = 0; i < NOB; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 w[NPIECE];
    for (int k = 0; k < NPIECE; ++k) for (int i = 0; i < 8; ++i) w[k][i] = (__bf16)(a0 * (1 + i + k));
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = a0 + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
        bf16x8 p[NPIECE];
        split8<NPIECE>(x, p);
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            // piece pairs in decreasing magnitude
            acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], p[0], acc[ob], 0, 0, 0);
            acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], p[1], acc[ob], 0, 0, 0);
            acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], p[0], acc[ob], 0, 0, 0);
            if constexpr (NPROD >= 4) acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], p[1], acc[ob], 0, 0, 0);
            if constexpr (NPROD >= 6) {
                acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], p[NPIECE - 1], acc[ob], 0, 0, 0);
                acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[NPIECE - 1], p[0], acc[ob], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = x[i] * 1.0001f + acc[0][i] * 1e-30f;  // new activations every step
    }
    float s = 0;
    for (int i = 0; i < NOB; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NPIECE, int NPROD, int NOB>
void run(float* d, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 512, iters = 40000;
    k_split<NPIECE, NPROD, NOB><<<blocks, 256>>>(d, 100, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0); k_split<NPIECE, NPROD, NOB><<<blocks, 256>>>(d, iters, 1e-3f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double sets = (double)blocks * 4 * iters * NOB;
    printf("%-28s NOB %d: %.1f fp32-equivalent TFLOP/s (%.0f bf16 TFLOP/s issued)\n", what, NOB, sets * 2.0 * 32 * 32 * 16 / ms / 1e9,
           sets * NPROD * 2.0 * 32 * 32 * 16 / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 512 * 256 * 4);
    run<2, 3, 4>(d, "2 pieces, 3 products"); run<2, 4, 4>(d, "2 pieces, 4 products"); run<3, 6, 4>(d, "3 pieces, 6 products");
    run<2, 3, 1>(d, "2 pieces, 3 products"); run<3, 6, 1>(d, "3 pieces, 6 products");
    return 0;
}
