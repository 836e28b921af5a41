// libgf3rx -- the screened frames-mode sync (gf3rx_fscreen.h): fp32 search windows with a proven bound.
#include "gf3rx_host.h"
#include "gf3rx_fscreen.h"

hipError_t launch_fscreen(const gf3_ctx* c, const FScreenArgs& a, int64_t F, hipStream_t st) {
    (void)c;
    if (F <= 0) return hipSuccess;
    DISPATCH_DT(a.dt, hipLaunchKernelGGL((corr_screen_kernel<DTC>), dim3((unsigned)F), dim3(64), 0, st, a));
    return hipGetLastError();
}
