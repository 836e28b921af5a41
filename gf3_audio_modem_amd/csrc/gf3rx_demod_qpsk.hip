// libgf3rx -- demod_kernel<.., MODE_QPSK>: bits only, the reference QPSK table (sign decisions; ping-pong FFT buffers).
#include "gf3rx_demod.h"

hipError_t launch_demod_qpsk(const gf3_ctx* c, const DemodArgs& a, int64_t F, hipStream_t st) {
    hipError_t e = hipSuccess;
    DISPATCH_NC(c->NC, a.dt, e = launch((demod_kernel<NCC, DTC, false, MODE_QPSK>), F, NCC / 8, demod_lds_bytes(c, true), st, a));
    return e;
}
