// Probe (gfx950): may the SOURCE registers of a v_mfma_f32_32x32x16_f16 be rewritten right behind its issue?
// k_fuse_color_h (two waves per SIMD, compiler-scheduled) came out wrong and non-deterministic in one product order and right in
// another (kpn_device.h, kpn_hlayer); in both hipcc reloads a weight piece's registers from LDS right behind the last MFMA that
// reads them, and with ONE wave per SIMD an MFMA is known to capture its operands at issue (scripts/mfma16_war_probe.hip).  Here the
// same sequences run with one and with two waves per SIMD, the MFMA that reads the registers being the LAST of a chain of dependent
// MFMAs (it cannot start before its predecessors have finished, and the other wave's MFMAs sit in the same pipe):
//     chain of N dependent MFMAs reading A = v[8:11] (all ones), then   ds_read_b128 v[8:11] <- zeros     (variant L)
//                                                                  or   v_mov_b32 v8..v11, 0              (variant V)
// Every accumulator element must come out as 16 N; an MFMA that read the rewritten A leaves less.
// MEASURED on the MI355X (profiles/r04_final_mfma_src_reload_probe.txt): 0 wrong lanes in every variant, with 1, 2 and 4 waves per
// SIMD, chains of 1 to 6 — an MFMA's sources ARE captured at issue; this is not what goes wrong in that build.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_src_reload_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>

#define MF "v_mfma_f32_32x32x16_f16 v[32:47], v[8:11], v[12:15], v[32:47]\n\t"
#define ZERO_ACC "v_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\tv_mov_b32 v36, 0\n\tv_mov_b32 v37, 0\n\tv_mov_b32 v38, 0\n\tv_mov_b32 v39, 0\n\t" \
                 "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\tv_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\t"
#define ONES(r) "v_mov_b32 v" #r ", 0x3c003c00\n\t"
#define SET_AB ONES(8) ONES(9) ONES(10) ONES(11) ONES(12) ONES(13) ONES(14) ONES(15) "s_nop 4\n\t"
#define RELOAD_L "ds_read_b128 v[8:11], %1\n\ts_waitcnt lgkmcnt(0)\n\t"
#define RELOAD_V "v_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\t"
#define DRAIN "s_sleep 40\n\ts_nop 15\n\tv_mov_b32 %0, v32\n\t"   // ~2,500 cycles: every wave's chain has left the pipe
#define CLOB "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"

#define KERNEL(NAME, CHAIN, RELOAD, N)                                                                          \
    __global__ void NAME(unsigned* bad, float* sample, int iters) {                                            \
        __shared__ float4 zeros[6144];   /* 96 KB: one workgroup per CU, so blockDim / 256 = waves per SIMD */        \
        zeros[threadIdx.x & 63] = make_float4(0.f, 0.f, 0.f, 0.f);                                             \
        if (iters < 0) zeros[threadIdx.x + 64] = make_float4(1.f, 1.f, 1.f, 1.f);                              \
        __syncthreads();                                                                                       \
        const unsigned addr = (unsigned)(size_t)(&zeros[threadIdx.x & 63]) & 0x3ffffu;                          \
        unsigned wrong = 0;                                                                                    \
        float last = 0.f;                                                                                      \
        for (int it = 0; it < iters; ++it) {                                                                   \
            float r;                                                                                           \
            asm volatile(ZERO_ACC SET_AB CHAIN RELOAD DRAIN : "=v"(r) : "v"(addr) : CLOB);                    \
            if (r != 16.0f * N) ++wrong;                                                                       \
            last = r;                                                                                          \
        }                                                                                                      \
        if (wrong) atomicAdd(bad, wrong);                                                                      \
        if (blockIdx.x == 0 && threadIdx.x == 0) *sample = last;                                               \
    }
KERNEL(k_l1, MF, RELOAD_L, 1) KERNEL(k_l2, MF MF, RELOAD_L, 2) KERNEL(k_l3, MF MF MF, RELOAD_L, 3) KERNEL(k_l6, MF MF MF MF MF MF, RELOAD_L, 6)
KERNEL(k_v1, MF, RELOAD_V, 1) KERNEL(k_v2, MF MF, RELOAD_V, 2) KERNEL(k_v3, MF MF MF, RELOAD_V, 3) KERNEL(k_v6, MF MF MF MF MF MF, RELOAD_V, 6)

int main() {
    unsigned* bad; float* sample;
    hipMalloc(&bad, 4); hipMalloc(&sample, 4);
    typedef void (*kern)(unsigned*, float*, int);
    struct { kern k; const char* name; int n; } ks[8] = {{k_l1, "LDS reload behind a chain of 1", 1}, {k_l2, "LDS reload behind a chain of 2", 2}, {k_l3, "LDS reload behind a chain of 3", 3},
                                                       {k_l6, "LDS reload behind a chain of 6", 6}, {k_v1, "VALU rewrite behind a chain of 1", 1}, {k_v2, "VALU rewrite behind a chain of 2", 2},
                                                       {k_v3, "VALU rewrite behind a chain of 3", 3}, {k_v6, "VALU rewrite behind a chain of 6", 6}};
    const int iters = 2000;
    printf("lanes with a wrong accumulator (of lanes x %d iterations), 1024 workgroups:\n%-36s %16s %16s %16s\n", iters, "", "1 wave/SIMD (256)", "2 waves/SIMD (512)", "4 waves/SIMD (1024)");
    for (auto& e : ks) {
        printf("%-36s", e.name);
        for (int threads : {256, 512, 1024}) {
            hipMemset(bad, 0, 4);
            hipLaunchKernelGGL(e.k, dim3(1024), dim3(threads), 0, 0, bad, sample, iters);
            hipDeviceSynchronize();
            unsigned h; float s;
            hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(&s, sample, 4, hipMemcpyDeviceToHost);
            printf(" %10u (%4.0f)", h, s);
        }
        printf("   expected %d\n", 16 * e.n);
    }
    return 0;
}
