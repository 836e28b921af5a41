// libgf3rx -- demod_kernel<.., MODE_SCAN>: bits only, any constellation.
#include "gf3rx_demod.h"

hipError_t launch_demod_scan(const gf3_ctx* c, const DemodArgs& a, int64_t F, hipStream_t st) {
    hipError_t e = hipSuccess;
    DISPATCH_NC(c->NC, a.dt, e = launch((demod_kernel<NCC, DTC, false, MODE_SCAN>), F, NCC / 8, demod_lds_bytes(c, GF3_ABL >= 2), st, a));
    return e;
}
