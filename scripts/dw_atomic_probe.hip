// Probe (gfx950): what would "dW inside the backward kernel" cost on the atomic units?  dW = dY^T X of the geometry MLP's backward
// (k_geo_rows_bwd, field_bwd_kernels.hip) is 76 accumulator blocks of 32 x 32 fp32 (311 KB).  A CU cannot hold them beside the
// kernel's own registers, so an in-kernel dW has to leave the CU once per (tile, view): 76 x 4 KB of fp32
// additions into a buffer every workgroup shares.  This probe issues exactly that traffic — 18,432 (tile, view) items (1024 rays x
// 192 field evaluations x 3 views / 32 rows), each adding 76 blocks of 1024 floats, lane-consecutive addresses, no arithmetic in
// between — and times it:
//   A. device-scope global_atomic_add_f32 into ONE buffer (what a plain atomicAdd is: executed at the memory side, the XCDs' L2s
//      are not coherent with each other);
//   B. workgroup-scope atomics into one buffer PER XCD (8 x 311 KB, chosen by HW_REG_XCC_ID; executed in the XCD's own L2; the
//      eight partial sums would be added by a ninth kernel), checked: the eight copies sum to the right total;
//   C. plain coalesced stores of the same bytes (no atomics): the floor of moving 76 x 4 KB per item at all.
// Against: k_weight_grad reads the 7.1 KB per row of dumps once, 1.00 ms for the same 589,824 rows (profiles/r05_z_train_kernel_stats.md).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/dw_atomic_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int BLOCKS = 76, BLOCK_FLOATS = 1024, DW = BLOCKS * BLOCK_FLOATS;   // 77,824 floats = 311 KB

__device__ __forceinline__ int xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_dw(float* dw, int items, int* tickets, float* sink) {
    float* base = dw;
    if (MODE == 1) base = dw + (size_t)xcc_id() * DW;
    __shared__ int wi_s;
    for (;;) {
        if (threadIdx.x == 0) wi_s = atomicAdd(tickets, 1);
        __syncthreads();
        const int wi = wi_s;
        __syncthreads();
        if (wi >= items) return;
        if (MODE == 2) base = sink + (size_t)(wi % 4096) * DW;          // stores: a 1.2 GB ring, nothing shared
#pragma unroll 4
        for (int i = threadIdx.x; i < DW; i += 256) {
            const float v = 1.0f;
            if (MODE == 0) __hip_atomic_fetch_add(base + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 1) __hip_atomic_fetch_add(base + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else base[i] = v;
        }
    }
}

int main() {
    const int items = 18432;
    float *dw, *sink; int* tickets;
    hipMalloc(&dw, (size_t)8 * DW * 4); hipMalloc(&sink, (size_t)4096 * DW * 4); hipMalloc(&tickets, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"A device-scope atomics, one buffer", "B workgroup-scope atomics, one buffer per XCD", "C plain stores (no sharing)"};
    for (int mode = 0; mode < 3; ++mode)
        for (int wgs : {256, 1024, 2048}) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(dw, 0, (size_t)8 * DW * 4); hipMemset(tickets, 0, 4); hipDeviceSynchronize();
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_dw<0>, dim3(wgs), dim3(256), 0, 0, dw, items, tickets, sink);
                else if (mode == 1) hipLaunchKernelGGL(k_dw<1>, dim3(wgs), dim3(256), 0, 0, dw, items, tickets, sink);
                else hipLaunchKernelGGL(k_dw<2>, dim3(wgs), dim3(256), 0, 0, dw, items, tickets, sink);
                hipEventRecord(e1); hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double bytes = (double)items * DW * 4;
            printf("%-48s %5d workgroups: %8.3f ms  (%.2f TB/s of additions, %.1f ns per item and CU)\n", names[mode], wgs, best, bytes / (best * 1e-3) / 1e12,
                   best * 1e6 / (items / 256.0));
            if (mode < 2) {
                std::vector<float> h((size_t)8 * DW);
                hipMemcpy(h.data(), dw, h.size() * 4, hipMemcpyDeviceToHost);
                double s0 = 0, sAll = 0; int used = 0;
                for (int c = 0; c < 8; ++c) { double s = 0; for (int i = 0; i < DW; ++i) s += h[(size_t)c * DW + i]; if (s > 0) ++used; if (c == 0) s0 = s; sAll += s; }
                printf("    check: sum over copies / expected = %.6f (%d copies used)\n", sAll / ((double)items * DW), used);
            }
        }
    return 0;
}
