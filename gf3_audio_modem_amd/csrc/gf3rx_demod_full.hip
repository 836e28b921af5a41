// libgf3rx -- demod_kernel<.., MODE_FULL>: bits + the dumps (eq, Hest); also the spectra-input form behind gf3_equalise.
#include "gf3rx_demod.h"

hipError_t launch_demod_full(const gf3_ctx* c, const DemodArgs& a, int64_t F, hipStream_t st) {
    hipError_t e = hipSuccess;
    DISPATCH_NC(c->NC, a.dt, e = launch((demod_kernel<NCC, DTC, false, MODE_FULL>), F, NCC / 8, demod_lds_bytes(c, false), st, a));
    return e;
}

// receiver.equalise as a stand-alone stage: frequency-domain inputs (complex128), everything else as above
hipError_t launch_demod_spectra(const gf3_ctx* c, const DemodArgs& a, int64_t F, hipStream_t st) {
    switch (c->NC) {
#ifndef GF3_DEV_BUILD
        case 512:  return launch((demod_kernel<512, DT_F64, true, MODE_FULL>), F, 64, demod_lds_bytes(c), st, a);
        case 1024: return launch((demod_kernel<1024, DT_F64, true, MODE_FULL>), F, 128, demod_lds_bytes(c), st, a);
        case 4096: return launch((demod_kernel<4096, DT_F64, true, MODE_FULL>), F, 512, demod_lds_bytes(c), st, a);
#endif
        default:   return launch((demod_kernel<2048, DT_F64, true, MODE_FULL>), F, 256, demod_lds_bytes(c), st, a);
    }
}
