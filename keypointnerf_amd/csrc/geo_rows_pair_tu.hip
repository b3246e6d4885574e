// Translation unit of k_geo_rows_h2 (geo_rows_pair_kernels.hip) for the device build: compiled with -fno-slp-vectorize.
// The kernel's VALU work is interleaved with its MFMAs by hand; the SLP vectoriser gathers those scalar operations into
// packed-f32 lumps (v_pk_mul_f32 ...) placed ahead of the MFMAs they were meant to sit between (measured: 7.4 -> 6.8 ms per
// launch without it), and packed f32 VALU is an anti-lever beside MFMAs anyway (MI355X_MICROARCH.md).  The flag changes
// the rounding of other kernels' contracted arithmetic, so it is confined to this file; the rest of the library is
// kpn_api.hip.  The host emulator build includes the kernel into kpn_api.hip directly (one translation unit, no flags).
#include "kpn_field_shared.h"
#include "geo_rows_pair_kernels.hip"

// not part of the C ABI: called by run_field (kpn_api.hip) for rows modes 2 and 3
extern "C" __attribute__((visibility("hidden"))) void kpn_internal_launch_geo_rows_pair(
    int mode, int blocks, void* stream, const kpn_scene_dev* sc, const kpn_points* ps, const float* wp, const int* list, const int* count,
    int* tickets, float* xscr, const kpn_batch* batch) {
    if (batch->pool) {   // pooling over the views inside the kernel (POOL layout of the scratch)
        if (mode == 3) KPN_LAUNCH(k_geo_rows_f2p, dim3(blocks), dim3(256), stream, *sc, *ps, wp, list, count, tickets, xscr, *batch);
        else KPN_LAUNCH(k_geo_rows_h2p, dim3(blocks), dim3(256), stream, *sc, *ps, wp, list, count, tickets, xscr, *batch);
    } else if (mode == 3)
        KPN_LAUNCH(k_geo_rows_f2, dim3(blocks), dim3(256), stream, *sc, *ps, wp, list, count, tickets, xscr, *batch);
    else
        KPN_LAUNCH(k_geo_rows_h2, dim3(blocks), dim3(256), stream, *sc, *ps, wp, list, count, tickets, xscr, *batch);
}
// the gather records of the same batch (the pair-tile kernels do not write them)
extern "C" __attribute__((visibility("hidden"))) void kpn_internal_launch_row_records(
    int blocks, void* stream, const kpn_scene_dev* sc, const kpn_points* ps, const float* wp, const int* list, const int* count, float* xscr,
    const kpn_batch* batch) {
    KPN_LAUNCH(k_row_records, dim3(blocks), dim3(256), stream, *sc, *ps, wp, list, count, xscr, *batch);
}
// ... of the batch's live points only (density-first render passes)
extern "C" __attribute__((visibility("hidden"))) void kpn_internal_launch_row_records_live(
    int blocks, void* stream, const kpn_scene_dev* sc, const kpn_points* ps, const float* wp, const int* list, const int* count,
    const int* tickets, const int* live, float* xscr, const kpn_batch* batch) {
    KPN_LAUNCH(k_row_records_live, dim3(blocks), dim3(256), stream, *sc, *ps, wp, list, count, tickets, live, xscr, *batch);
}

#ifdef KPN_PRECISION_PROBE
extern "C" __attribute__((visibility("hidden"))) int kpn_internal_probe_set_mask_pair(unsigned long long m) {
    return hipMemcpyToSymbol(HIP_SYMBOL(kpn_probe_mask_dev), &m, sizeof(m)) != hipSuccess;
}
#endif
#ifdef KPN_H2_TIMING
// debug builds only: read (and clear) the per-phase cycle sums of k_geo_rows_h2
extern "C" int kpn_h2_timing(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(kpn_h2_cycles), 64) != hipSuccess) return 1;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(kpn_h2_cycles), z, 64) != hipSuccess;
}
#endif
