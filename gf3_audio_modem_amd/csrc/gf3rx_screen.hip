// libgf3rx -- the heavy kernels of the screened stream sync (gf3rx_screen.h): the fp32 screening pass and the fp64
// re-evaluation of the listed cells.  gf3rx_sync.hip strings them together with the list kernels.
#include "gf3rx_host.h"
#include "gf3rx_screen.h"

// The screening pass: the band-limited ring kernel when the plan allows it (and `general` is not asked for), else
// the general kernel.  R, the ring kernel's blocks per workgroup: every workgroup transforms Q - 1 windows without
// finishing a block, so R is as large as leaves a whole number of rounds of 2 workgroups per CU (swept on the
// config-3 stream when it was 83 582 blocks and every block ran its inverse transform: R = 164 -> 510 workgroups
// 1.71 ms, R = 82 1.72, R = 32 2.0, and 2.2 at R = 110 = 1.5 rounds).
hipError_t launch_screen(const gf3_ctx* c, ScreenArgs a, bool general, hipStream_t st) {
    const auto& sp = c->scr;
    hipError_t e = hipSuccess;
    if (sp.ring && !general) {
        const int64_t slots = 2 * (int64_t)c->n_cu;                                       // workgroups resident at once
        const int64_t rounds = (a.nblk + slots * 170 - 1) / (slots * 170);
        int64_t R = sp.R_forced > 0 ? sp.R_forced : (a.nblk + slots * rounds - 1) / (slots * rounds);
        if (R < 1) R = 1;                                                                 // (short streams: one block per workgroup, Q windows each, all at once)
        a.Hb = sp.d_Hb; a.ecoef = sp.d_ecoef; a.R = (int)R;
        const size_t lds = (size_t)2 * GF3_SCR_NC * sizeof(cf) + (64 + 64 + 8 + 2 + 2 * GF3_SCR_RQ + 2) * sizeof(float);
        const int64_t grid = (((a.nblk + R - 1) / R + 7) / 8) * 8;                        // padded to the 8 XCDs (xcd_order)
        DISPATCH_DT(a.dt, e = launch((scr_ring_kernel<DTC>), grid, GF3_SCR_T, lds, st, a));
    } else {
        const size_t lds = (size_t)2 * GF3_SCR_NC * sizeof(cf) + (128 + 11 * GF3_SCR_B + 4) * sizeof(float);
        const int64_t grid = (((a.nblk + GF3_SCR_B - 1) / GF3_SCR_B + 7) / 8) * 8;
        DISPATCH_DT(a.dt, e = launch((scr_ols_kernel<DTC>), grid, GF3_SCR_T, lds, st, a));
    }
    return e;
}


// fp64 re-evaluation of the listed cells (a wave per cell at a time; the list length lives on the device, so the grid is
// sized by the list's capacity)
hipError_t launch_refine(const gf3_ctx* c, const RefineArgs& a, int64_t cap_cells, hipStream_t st) {
    const int64_t slots = 2 * (int64_t)c->n_cu;                         // (the LDS staging allows two workgroups per CU)
    const int64_t wgs = (cap_cells + 3) / 4;                             // (a wave per cell at a time)
    const unsigned grid = (unsigned)(wgs < slots ? wgs : slots);
#if GF3_REFINE_MFMA
    const int64_t wg4 = 4 * slots;                                       // (no LDS staging: more resident waves, a wave per cell)
    DISPATCH_DT(a.dt, hipLaunchKernelGGL((scr_refine_mfma_kernel<DTC>), dim3((unsigned)(wg4 < wgs ? wg4 : wgs)), dim3(SCR_REF_THREADS), 0, st, a));
    (void)grid;
#else
    DISPATCH_DT(a.dt, hipLaunchKernelGGL((scr_refine_kernel<DTC>), dim3(grid), dim3(SCR_REF_THREADS), 0, st, a));
#endif
    return hipGetLastError();
}
