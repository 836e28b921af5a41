// libgf3rx -- data stage of the two-phase demodulation (gf3rx_demod_split.hip): demod_kernel<.., MODE_QPSK, STAGE_DATA>.
#include "gf3rx_demod.h"

hipError_t launch_dsplit_qpsk(const gf3_ctx* c, const DemodArgs& a, int64_t grid, hipStream_t st) {
    hipError_t e = hipSuccess;
    DISPATCH_NC(c->NC, a.dt, e = launch((demod_kernel<NCC, DTC, false, MODE_QPSK, STAGE_DATA>), grid, NCC / 8, demod_lds_bytes(c, true), st, a));
    return e;
}
