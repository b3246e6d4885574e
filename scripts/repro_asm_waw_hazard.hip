// Minimal reproducer (gfx950) of what made one build of k_fuse_color_h wrong and non-deterministic:
// a VALU instruction that WRITES a register of the destination tuple of an MFMA still in flight loses against the MFMA's
// write-back (XDL write -> VALU write, WAW).  hipcc pads this pair with wait states when it sees the VALU instruction; it does not
// look inside an inline-asm statement, and its register allocator is free to give the statement's outputs the DEAD registers of
// that tuple (a layer whose output block is only partly used).  In the wrong build (scripts/isa_asm_hazards.py on its assembly):
//     v_mfma_f32_32x32x16_f16 v[48:63], v[132:135], v[96:99], v[48:63]     ; last MFMA of ray_encoder.2's block 1 (3 of 16 registers live)
//     ... 3 instructions ...
//     ;;#ASMSTART  v_cvt_pk_f16_f32 v56, v74, v75 ...                      ; the next layer's operand piece, allocated INTO v[48:63]
// Here, inside ONE asm statement so that the distance is exact:
//     v_mfma D = v[32:47] (every element 16)  ;  s_nop K-1 (K states, K = 0: none)  ;  v_mov_b32 v40, 42.0  ;  long drain  ;  read v40
// v40 must be 42; where the MFMA's write-back landed after the v_mov it is 16.  Printed per K: lanes that read 16, with one, two
// and four waves per SIMD (the other waves' MFMAs sit in the same pipe and delay the write-back further).
//   hipcc --offload-arch=gfx950 -O3 scripts/repro_asm_waw_hazard.hip -o /tmp/waw && /tmp/waw
#include <hip/hip_runtime.h>
#include <cstdio>

#define ZERO4(a, b, c, d) "v_mov_b32 v" #a ", 0\n\tv_mov_b32 v" #b ", 0\n\tv_mov_b32 v" #c ", 0\n\tv_mov_b32 v" #d ", 0\n\t"
#define ZERO_ACC ZERO4(32, 33, 34, 35) ZERO4(36, 37, 38, 39) ZERO4(40, 41, 42, 43) ZERO4(44, 45, 46, 47)
#define ONES(r) "v_mov_b32 v" #r ", 0x3c003c00\n\t"   // two fp16 ones
#define SET_AB ONES(8) ONES(9) ONES(10) ONES(11) ONES(12) ONES(13) ONES(14) ONES(15) "s_nop 7\n\t"
#define MFMA "v_mfma_f32_32x32x16_f16 v[32:47], v[8:11], v[12:15], v[32:47]\n\t"
#define DRAIN "s_sleep 40\n\ts_nop 15\n\tv_mov_b32 %0, v40\n\t"
#define CLOB "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", \
             "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"

template <int K>
__global__ void k_waw(unsigned* lost, int iters) {
    unsigned n = 0;
    for (int it = 0; it < iters; ++it) {
        float r;
        if constexpr (K == 0)
            asm volatile(ZERO_ACC SET_AB MFMA "v_mov_b32 v40, 0x42280000\n\t" DRAIN : "=v"(r) : : CLOB);
        else
            asm volatile(ZERO_ACC SET_AB MFMA "s_nop %1\n\tv_mov_b32 v40, 0x42280000\n\t" DRAIN : "=v"(r) : "n"(K - 1) : CLOB);
        if (r != 42.0f) ++n;    // 16 = the MFMA's write-back overwrote the VALU result
    }
    if (n) atomicAdd(lost, n);
}

template <int K>
void run(unsigned* lost) {
    const int iters = 200;
    printf("%2d wait states:", K);
    for (int threads : {256, 512, 1024}) {   // x 1 workgroup per CU = 1, 2, 4 waves per SIMD
        hipMemset(lost, 0, 4);
        hipLaunchKernelGGL(k_waw<K>, dim3(256), dim3(threads), 0, 0, lost, iters);
        hipDeviceSynchronize();
        unsigned h;
        hipMemcpy(&h, lost, 4, hipMemcpyDeviceToHost);
        printf("  %9u of %9u", h, 256u * threads * iters);
    }
    printf("\n");
}

int main() {
    unsigned* lost;
    hipMalloc(&lost, 4);
    printf("lanes whose VALU write into an in-flight MFMA's destination tuple was lost (1 / 2 / 4 waves per SIMD)\n");
    run<0>(lost); run<1>(lost); run<2>(lost); run<3>(lost); run<4>(lost); run<6>(lost); run<8>(lost); run<10>(lost); run<11>(lost);
    run<12>(lost); run<13>(lost); run<14>(lost); run<16>(lost);
    return 0;
}
