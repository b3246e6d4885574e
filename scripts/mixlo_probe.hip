// Probe (gfx950) for two cheaper forms of the operand production of the two-fp16-piece kernels:
//   A. the lo piece  l = fp16(x - h)  as  v_fma_mixlo_f16 / v_fma_mixhi_f16  (the fused x - h rounded ONCE to fp16, written into
//      the low / high half of the destination) instead of  v_fma_mix_f32 x 2 + v_cvt_pk_f16_f32 : 2 instead of 3 instructions per
//      pair of values.  x - h is exact in fp32, so both forms round the same real number to fp16 — checked bit for bit here,
//      including the fp16 subnormal range, zeros, infinities and NaNs;
//   B. the two "+ 1" and the two final additions of the log2-unit Softplus as v_pk_add_f32 on an aligned register pair;
//   C. the issue cost of the complete per-pair slice patterns beside v_mfma_f32_32x32x16_f16, one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/mixlo_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void k_split_both(const float* x, int n, unsigned* l_old, unsigned* l_new, unsigned* h_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    unsigned ph, lo_old, lo_new;
    float ra, rb;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph) : "v"(a), "v"(b));
    asm volatile("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                 : "=&v"(ra), "=&v"(rb) : "v"(ph), "v"(a), "v"(b));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo_old) : "v"(ra), "v"(rb));
    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                 : "=&v"(lo_new) : "v"(ph), "v"(a), "v"(b));
    l_old[i] = lo_old; l_new[i] = lo_new; h_out[i] = ph;
}
typedef float f2 __attribute__((ext_vector_type(2)));
// log2-unit Softplus of a pair: scalar adds vs packed adds
__global__ void k_softplus_both(const float* x, int n, float* y_old, float* y_new) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    float x0 = x[2 * i], x1 = x[2 * i + 1], e0, e1;
    asm volatile("v_exp_f32 %0, -|%2|\n\tv_exp_f32 %1, -|%3|\n\ts_nop 0\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %1, 1.0, %1\n\t"
                 "v_log_f32 %0, %0\n\tv_log_f32 %1, %1\n\tv_max_f32 %2, 0, %2\n\tv_max_f32 %3, 0, %3\n\tv_add_f32 %2, %2, %0\n\tv_add_f32 %3, %3, %1"
                 : "=&v"(e0), "=&v"(e1), "+v"(x0), "+v"(x1));
    y_old[2 * i] = x0; y_old[2 * i + 1] = x1;
    f2 xv = {x[2 * i], x[2 * i + 1]}, ev;
    asm volatile("v_exp_f32 v200, -|v202|\n\tv_exp_f32 v201, -|v203|\n\ts_nop 0\n\tv_pk_add_f32 v[200:201], v[200:201], 1.0 op_sel_hi:[1,0]\n\t"
                 "v_log_f32 v200, v200\n\tv_log_f32 v201, v201\n\tv_max_f32 v202, 0, v202\n\tv_max_f32 v203, 0, v203\n\ts_nop 0\n\t"
                 "v_pk_add_f32 v[202:203], v[202:203], v[200:201]"
                 : "={v[200:201]}"(ev), "+{v[202:203]}"(xv));
    y_new[2 * i] = xv.x; y_new[2 * i + 1] = xv.y;
}

// ---- issue cost: one activated operand pair per four MFMAs (registers: v16 v17 = x, v18 v19 = e, v20 = hi pieces, v21 = lo pieces) ----
#define MF16(acc) "v_mfma_f32_32x32x16_f16 " acc ", v[8:11], v[12:15], " acc "\n\t"
// the product's slices (geo_rows_pair_kernels.hip kpn_h2_slice, fp16 scheme)
#define P_NOW MF16("a[0:15]") "v_accvgpr_read_b32 v16, a100\n\tv_accvgpr_read_b32 v17, a101\n\tv_exp_f32 v18, -|v16|\n\tv_exp_f32 v19, -|v17|\n\t" \
              MF16("a[16:31]") "v_max_f32 v16, 0, v16\n\tv_add_f32 v18, 1.0, v18\n\tv_add_f32 v19, 1.0, v19\n\tv_log_f32 v18, v18\n\t" \
              MF16("a[32:47]") "v_log_f32 v19, v19\n\tv_max_f32 v17, 0, v17\n\tv_add_f32 v16, v16, v18\n\tv_add_f32 v17, v17, v19\n\tv_cvt_pk_f16_f32 v20, v16, v17\n\t" \
              MF16("a[48:63]") "v_fma_mix_f32 v16, v20, -1.0, v16 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 v17, v20, -1.0, v17 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_cvt_pk_f16_f32 v21, v16, v17\n\t"
// A: lo pieces by v_fma_mixlo / mixhi
#define P_MIX MF16("a[0:15]") "v_accvgpr_read_b32 v16, a100\n\tv_accvgpr_read_b32 v17, a101\n\tv_exp_f32 v18, -|v16|\n\tv_exp_f32 v19, -|v17|\n\t" \
              MF16("a[16:31]") "v_max_f32 v16, 0, v16\n\tv_add_f32 v18, 1.0, v18\n\tv_add_f32 v19, 1.0, v19\n\tv_log_f32 v18, v18\n\t" \
              MF16("a[32:47]") "v_log_f32 v19, v19\n\tv_max_f32 v17, 0, v17\n\tv_add_f32 v16, v16, v18\n\tv_add_f32 v17, v17, v19\n\tv_cvt_pk_f16_f32 v20, v16, v17\n\t" \
              MF16("a[48:63]") "v_fma_mixlo_f16 v21, v20, -1.0, v16 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 v21, v20, -1.0, v17 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
// A + B: packed additions as well (13 instead of 16 instructions per pair)
#define P_PK  MF16("a[0:15]") "v_accvgpr_read_b32 v16, a100\n\tv_accvgpr_read_b32 v17, a101\n\tv_exp_f32 v18, -|v16|\n\tv_exp_f32 v19, -|v17|\n\t" \
              MF16("a[16:31]") "v_max_f32 v16, 0, v16\n\tv_pk_add_f32 v[18:19], v[18:19], 1.0 op_sel_hi:[1,0]\n\tv_max_f32 v17, 0, v17\n\tv_log_f32 v18, v18\n\t" \
              MF16("a[32:47]") "v_log_f32 v19, v19\n\ts_nop 0\n\tv_pk_add_f32 v[16:17], v[16:17], v[18:19]\n\tv_cvt_pk_f16_f32 v20, v16, v17\n\t" \
              MF16("a[48:63]") "v_fma_mixlo_f16 v21, v20, -1.0, v16 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 v21, v20, -1.0, v17 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
// no VALU work at all: the matrix pipe alone
#define P_NONE MF16("a[0:15]") MF16("a[16:31]") MF16("a[32:47]") MF16("a[48:63]")
#define CLOBBERS "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", \
    "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", \
    "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a100", "a101"
#define KERNEL(NAME, TEXT)                                                                               \
    __global__ __launch_bounds__(256, 1) void NAME(float* out, long long* cycles, int slot) {            \
        long long t0, t1;                                                                                \
        asm volatile("v_mov_b32 v6, 1.0\n\tv_mov_b32 v7, 1.0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"v6", "v7"); \
        for (int it = 0; it < 256; ++it) asm volatile(TEXT TEXT TEXT TEXT ::: CLOBBERS);               \
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));          \
        if (threadIdx.x == 0 && blockIdx.x == 0) cycles[slot] = t1 - t0;                                 \
        if (out) out[threadIdx.x] = 0.f;                                                                 \
    }
KERNEL(k_now, P_NOW) KERNEL(k_mix, P_MIX) KERNEL(k_pk, P_PK) KERNEL(k_none, P_NONE)

int main() {
    const int n = 1 << 24;
    std::vector<float> x(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const double e = -40.0 + 57.0 * (rand() / (double)RAND_MAX);   // 2^-40 .. 2^17: beyond fp16's range at both ends
        x[i] = (float)((rand() & 1 ? 1 : -1) * exp2(e) * (1.0 + rand() / (double)RAND_MAX));
    }
    const float specials[] = {0.0f, -0.0f, 65504.0f, 65520.0f, 1e5f, -1e5f, INFINITY, -INFINITY, NAN, 1e-40f, -1e-40f, 5.9604645e-8f, 6.1035156e-5f, 1.0f, -1.0f, 0.1f};
    for (int i = 0; i < 16; ++i) x[i] = specials[i];
    float* dx; unsigned *d0, *d1, *d2;
    hipMalloc(&dx, (size_t)n * 4); hipMalloc(&d0, (size_t)n * 2); hipMalloc(&d1, (size_t)n * 2); hipMalloc(&d2, (size_t)n * 2);
    hipMemcpy(dx, x.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_split_both, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, d0, d1, d2);
    std::vector<unsigned> lo_old(n / 2), lo_new(n / 2), hi(n / 2);
    hipMemcpy(lo_old.data(), d0, (size_t)n * 2, hipMemcpyDeviceToHost); hipMemcpy(lo_new.data(), d1, (size_t)n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hi.data(), d2, (size_t)n * 2, hipMemcpyDeviceToHost);
    long diff = 0, diff_nan_only = 0, subnormal_lo = 0;
    for (int i = 0; i < n / 2; ++i) {
        for (int half = 0; half < 2; ++half) {
            const unsigned a = (lo_old[i] >> (16 * half)) & 0xffffu, b = (lo_new[i] >> (16 * half)) & 0xffffu;
            if ((a & 0x7c00u) == 0 && (a & 0x3ffu) != 0) ++subnormal_lo;
            if (a != b) {
                const bool nan_a = (a & 0x7c00u) == 0x7c00u && (a & 0x3ffu), nan_b = (b & 0x7c00u) == 0x7c00u && (b & 0x3ffu);
                if (nan_a && nan_b) ++diff_nan_only; else { if (diff < 8) printf("  x = %.9g: lo old 0x%04x new 0x%04x (hi 0x%04x)\n", x[2 * i + half], a, b, (hi[i] >> (16 * half)) & 0xffffu); ++diff; }
            }
        }
    }
    printf("A. lo pieces of %d values (2^-40..2^17 + specials), %ld of them fp16 subnormals: %ld differ between cvt_pk(fma_mix) and fma_mixlo/hi (%ld more are NaNs with different payloads)\n",
           n, subnormal_lo, diff, diff_nan_only);
    {   // B. softplus in log2 units, arguments -200 .. 200
        std::vector<float> u(n);
        for (int i = 0; i < n; ++i) u[i] = (float)(400.0 * (rand() / (double)RAND_MAX) - 200.0) * (i % 3 == 0 ? 0.01f : 1.0f);
        u[0] = NAN; u[1] = INFINITY; u[2] = -INFINITY; u[3] = 0.0f;
        float *du, *y0, *y1;
        hipMalloc(&du, (size_t)n * 4); hipMalloc(&y0, (size_t)n * 4); hipMalloc(&y1, (size_t)n * 4);
        hipMemcpy(du, u.data(), (size_t)n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_softplus_both, dim3(n / 2 / 256), dim3(256), 0, 0, du, n, y0, y1);
        std::vector<float> a(n), b(n);
        hipMemcpy(a.data(), y0, (size_t)n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), y1, (size_t)n * 4, hipMemcpyDeviceToHost);
        long d = 0;
        for (int i = 0; i < n; ++i) if (memcmp(&a[i], &b[i], 4) != 0 && !(std::isnan(a[i]) && std::isnan(b[i]))) { if (d < 8) printf("  u = %.9g: scalar %.9g packed %.9g\n", u[i], a[i], b[i]); ++d; }
        printf("B. log2-unit Softplus of %d arguments: %ld results differ between v_add_f32 and v_pk_add_f32 (NaN in -> NaN out: %s / %s)\n", n, d,
               std::isnan(a[0]) ? "yes" : "NO", std::isnan(b[0]) ? "yes" : "NO");
    }
    // C. issue cost
    long long* cyc; hipMalloc(&cyc, 64 * 8); hipMemset(cyc, 0, 64 * 8);
    typedef void (*kern)(float*, long long*, int);
    kern ks[4] = {k_none, k_now, k_mix, k_pk};
    const char* names[4] = {"MFMAs alone", "product slices (16 instructions per pair)", "lo pieces by fma_mixlo/hi (15)", "+ packed additions (13)"};
    printf("C. s_memtime ticks per v_mfma_f32_32x32x16_f16 (16 MFMAs x 256 iterations), one wave per SIMD, 256 CUs busy\n");
    for (int k = 0; k < 4; ++k) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(ks[k], dim3(256), dim3(256), 0, 0, (float*)nullptr, cyc, k); hipDeviceSynchronize(); }
        long long h; hipMemcpy(&h, cyc + k, 8, hipMemcpyDeviceToHost);
        printf("   %-44s %8.1f\n", names[k], (double)h / (256.0 * 16.0));
    }
    return 0;
}
