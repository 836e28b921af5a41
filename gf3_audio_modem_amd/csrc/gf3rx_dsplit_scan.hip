// libgf3rx -- data stage of the two-phase demodulation (gf3rx_demod_split.hip): demod_kernel<.., MODE_SCAN, STAGE_DATA>.
#include "gf3rx_demod.h"

hipError_t launch_dsplit_scan(const gf3_ctx* c, const DemodArgs& a, int64_t grid, hipStream_t st) {
    hipError_t e = hipSuccess;
    DISPATCH_NC(c->NC, a.dt, e = launch((demod_kernel<NCC, DTC, false, MODE_SCAN, STAGE_DATA>), grid, NCC / 8, demod_lds_bytes(c, GF3_ABL >= 2), st, a));
    return e;
}
