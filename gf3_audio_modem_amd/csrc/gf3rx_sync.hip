// libgf3rx -- chirp_method on whole streams (OFDM.py:356-372): the screened path and the all-fp64 path of
// gf3_sync_stream(_ex), the piecewise form for host ingest (gf3_sync_chunk / gf3_sync_decide), and the peak-picking
// kernels they share.  The heavy kernels are launched through gf3rx_screen.hip and gf3rx_corr.hip.
#include "gf3rx_host.h"
#include "gf3rx_screen_list.h"

// ... and the diagnostics of the calling thread's last gf3_sync_stream / gf3_sync_stream_ex (gf3_sync_stream_info)
static thread_local int64_t g_last_info[4] = {0, 0, 0, 0};
// Where a call reads its few result words back to: 256 bytes of PINNED host memory per calling thread (allocated on the
// thread's first call and released when the thread ends -- a pool that is recreated per job does not accumulate pinned
// pages; a copy into pageable memory goes through the runtime's staging path, which costs a call tens of microseconds).
// nullptr if the allocation fails -- the caller then copies into a local variable as before.
struct ReadbackPage {
    void* p = nullptr;
    bool tried = false;
    ~ReadbackPage() { if (p) { (void)hipHostFree(p); p = nullptr; } }   // (a failure at process teardown is of no consequence)
};
static void* readback_buffer() {
    static thread_local ReadbackPage page;
    if (!page.tried) {
        page.tried = true;
        if (hipHostMalloc(&page.p, 256, hipHostMallocPortable) != hipSuccess) { page.p = nullptr; (void)hipGetLastError(); }
    }
    return page.p;
}

// ============================================================================
// stream-mode peak picking on the full correlation P (OFDM.py:359-370)
// ============================================================================
#define PK_THREADS 256
#define PK_ITEMS 8

__global__ void pk_max_final(const double* partial, int n, double* out) {
    __shared__ double scratch[16];
    double mx = -INFINITY;
    bool nan = false;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const double v = partial[i]; mx = fmax(mx, v); nan = nan || !(v == v); }
    mx = block_max(mx, scratch);
    const int anynan = __syncthreads_or(nan ? 1 : 0);
    if (threadIdx.x == 0) out[0] = anynan ? NAN : mx;
}
// pass 0: count per block; pass 1: write ascending indices at the block's offset (blocks that counted none return
// at once, and candidates are a handful per chirp, so the second pass costs next to nothing).
// Candidate at i  <=>  (p1-p0)(p2-p1) <= 0 and p1 > thresh with p = P/max (OFDM.py:359-361: the division is done
// first there, so it is done here too -- one correctly rounded division per lag, shared by its three uses).
// Lags that cannot reach the threshold skip the divisions: P1 < thresh*max*(1-1e-6) implies fl(P1/max) < thresh.
__global__ void pk_candidates(const double* __restrict__ P, int64_t nz, const double* mxp, double thresh,
                              int64_t* counts, const int64_t* offsets, int64_t* cand) {
    __shared__ int wsum[PK_THREADS / 64];
    if (offsets && counts[blockIdx.x] == 0) return;
    const double mx = mxp[0];
    const bool filt = mx > 0.0 && thresh > 0.0 && mx < INFINITY && thresh < INFINITY;
    const double lim = filt ? thresh * mx * (1.0 - 1e-6) : -INFINITY;
    const int64_t base = ((int64_t)blockIdx.x * PK_THREADS + threadIdx.x) * PK_ITEMS;
    int c = 0;
    unsigned flags = 0;
    if (base < nz) {
        double q[PK_ITEMS + 2];                       // P[base .. base+PK_ITEMS+1] (nz = len - 2: always inside P when i < nz)
        const int cnt = (nz - base < PK_ITEMS) ? (int)(nz - base) : PK_ITEMS;
        bool any = false;
#pragma unroll
        for (int k = 0; k < PK_ITEMS + 2; ++k) q[k] = (k < cnt + 2) ? P[base + k] : 0.0;
#pragma unroll
        for (int k = 0; k < PK_ITEMS; ++k) any = any || (k < cnt && !(q[k + 1] < lim));
        if (any) {
#pragma unroll
            for (int k = 0; k < PK_ITEMS + 2; ++k) q[k] = q[k] / mx;
#pragma unroll
            for (int k = 0; k < PK_ITEMS; ++k)
                if (k < cnt && ((q[k + 1] - q[k]) * (q[k + 2] - q[k + 1]) <= 0.0) && (q[k + 1] > thresh)) { flags |= 1u << k; ++c; }
        }
    }
    // block-wide exclusive scan of c
    int x = c;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int woff = 0, total = 0;
    for (int w = 0; w < PK_THREADS / 64; ++w) { if (w < wave) woff += wsum[w]; total += wsum[w]; }
    if (!offsets) { if (threadIdx.x == 0) counts[blockIdx.x] = total; return; }
    int64_t o = offsets[blockIdx.x] + woff + (x - c);
    for (int k = 0; k < PK_ITEMS; ++k) if (flags & (1u << k)) cand[o++] = base + k;
}
#define SCAN_PER 16
// n_dev (optional, device): scan only the first min(n, *n_dev) counts -- lists whose length lives on the device
__global__ void pk_scan(const int64_t* counts, int64_t n, int64_t* offsets, int64_t* total, const long long* n_dev = nullptr,
                        const long long* void_if_odd = nullptr) {
    // one workgroup, exclusive scan; each thread owns SCAN_PER consecutive counts per step
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry;
    if (n_dev && (int64_t)n_dev[0] < n) n = n_dev[0] < 0 ? 0 : (int64_t)n_dev[0];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int64_t base = 0; base < n; base += (int64_t)blockDim.x * SCAN_PER) {
        const int64_t i0 = base + (int64_t)threadIdx.x * SCAN_PER;
        int64_t loc[SCAN_PER];
        int64_t c = 0;
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k) { loc[k] = (i0 + k < n) ? counts[i0 + k] : 0; c += loc[k]; }
        int64_t x = c;
        for (int d = 1; d < 64; d <<= 1) { const int64_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int64_t woff = 0, tot = 0;
        for (int w = 0; w < nw; ++w) { if (w < wave) woff += wsum[w]; tot += wsum[w]; }
        int64_t o = carry + woff + (x - c);
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k) { if (i0 + k < n) offsets[i0 + k] = o; o += loc[k]; }
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
    // (void_if_odd: the screened sync's status word -- a call that is about to fall back must not leave a list length
    //  that was summed over counts nobody wrote)
    if (threadIdx.x == 0) total[0] = (void_if_odd && (void_if_odd[0] & 1)) ? 0 : carry;
}
// sequential suppression (OFDM.py:364-370) over the sorted candidate list: an accepted candidate i suppresses
// everything up to i+Lc, so the next survivor is succ(k) = the first candidate >= i+Lc+1, and the accepted set is
// the orbit of the first candidate under succ.  One workgroup walks the list in chunks staged in LDS:
//   1. every thread finds succ of its candidates by binary search (parallel);
//   2. five doubling rounds give succ^2, succ^4 ... succ^32;
//   3. one wave emits 64 accepted peaks per step: lane l composes succ^l from the bits of l (six dependent LDS
//      reads for all lanes at once, instead of one dependent read per accepted peak).
#define NMS_CHUNK 2048
#define NMS_THREADS 1024
GF3_DEV int nms_lower_bound(const int64_t* v, int n, int64_t want) {     // first k in [0, n] with v[k] >= want
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (v[mid] >= want) hi = mid; else lo = mid + 1; }
    return lo;
}
__global__ __launch_bounds__(NMS_THREADS) void pk_nms(const int64_t* cand, const int64_t* totalp, int64_t Lc, int64_t nz,
                                                      int64_t* peaks, int64_t cap, int64_t* npeaks) {
    __shared__ int64_t sv[NMS_CHUNK];
    __shared__ unsigned short J[6][NMS_CHUNK + 1];                       // J[t][k] = succ^(2^t)(k); index m = "past the chunk"
    __shared__ int64_t st[3];                                            // np, status, want (wave 0 -> all)
    const int64_t total = totalp[0];
    if (threadIdx.x == 0) { st[0] = 0; st[1] = 0; st[2] = INT64_MIN; }
    __syncthreads();
    for (int64_t c0 = 0; c0 < total; c0 += NMS_CHUNK) {
        const int m = (int)((total - c0 < NMS_CHUNK) ? (total - c0) : NMS_CHUNK);
        for (int k = threadIdx.x; k < m; k += NMS_THREADS) sv[k] = cand[c0 + k];
        __syncthreads();
        if (st[1] == 1) break;                                           // wiped: nothing can be accepted any more
        for (int k = threadIdx.x; k <= m; k += NMS_THREADS)
            J[0][k] = (unsigned short)(k < m ? nms_lower_bound(sv, m, sv[k] + Lc + 1) : m);
        __syncthreads();
        {   // The usual stream: every candidate from the entry point on is followed, Lc + 1 later at the earliest, by
            // the NEXT candidate, so the orbit is the whole rest of the chunk and all of it is written at once
            // (the walk below emits 64 peaks per six dependent LDS reads: 14 us per chunk of 2048).
            const int k0 = nms_lower_bound(sv, m, st[2]);
            bool chain = true, wipe = false;
            for (int k = k0 + threadIdx.x; k < m; k += NMS_THREADS) { chain = chain && (J[0][k] == k + 1); wipe = wipe || (sv[k] + Lc >= nz); }
            const int64_t np = st[0], status = st[1];
            if (__syncthreads_and(chain ? 1 : 0)) {
                const int any_wipe = __syncthreads_or(wipe ? 1 : 0);
                if (any_wipe) {
                    if (threadIdx.x == 0) { st[0] = 0; st[1] = 1; }      // the except-branch wipes everything
                } else {
                    for (int k = k0 + threadIdx.x; k < m; k += NMS_THREADS) if (np + (k - k0) < cap) peaks[np + (k - k0)] = sv[k];
                    if (threadIdx.x == 0 && k0 < m) {
                        st[0] = np + (m - k0);
                        st[1] = (np + (m - k0) > cap) ? 2 : status;
                        st[2] = sv[m - 1] + Lc + 1;
                    }
                }
                __syncthreads();
                continue;
            }
        }
        for (int t = 1; t < 6; ++t) {
            for (int k = threadIdx.x; k <= m; k += NMS_THREADS) J[t][k] = J[t - 1][J[t - 1][k]];
            __syncthreads();
        }
        if (threadIdx.x < 64) {                                          // wave 0
            const int lane = threadIdx.x;
            int64_t np = st[0], status = st[1], want = st[2];
            int k0 = nms_lower_bound(sv, m, want);                       // wave-uniform
            while (k0 < m) {
                int k = k0;                                              // lane l: succ^l(k0)
#pragma unroll
                for (int t = 0; t < 6; ++t) if ((lane >> t) & 1) k = J[t][k];
                const bool live = k < m;
                const int64_t i = live ? sv[k] : 0;
                const unsigned long long wipe = __ballot(live && (i + Lc >= nz));
                if (wipe) { np = 0; status = 1; break; }                 // the except-branch wipes everything
                if (live) { if (np + lane < cap) peaks[np + lane] = i; }
                const unsigned long long lv = __ballot(live);
                const int cnt = __popcll(lv);                            // live lanes are a prefix: succ is increasing
                if (np + cnt > cap) status = 2;
                np += cnt;
                const int klast = __shfl(k, cnt - 1, 64);
                want = __shfl(i, cnt - 1, 64) + Lc + 1;
                k0 = J[0][klast];
            }
            if (lane == 0) { st[0] = np; st[1] = status; st[2] = want; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { npeaks[0] = st[0]; npeaks[1] = st[1]; }
}

// workspace layout for gf3_sync_stream
struct StreamWs { int64_t plen, nz, nb_max, nb_c, nblk, nwin; size_t o_P, o_part, o_cnt, o_off, o_cand, o_misc, o_spec, total;
                  // screened path (gf3rx_screen.h); P32 overlays o_P, the per-workgroup counts / offsets overlay o_cnt / o_off
                  int64_t s_nblk, s_ncell, s_nwg, s_cap;
                  size_t o_sblk, o_smisc, o_segm, o_cell, o_cval, o_mask, o_ccnt, o_coff; };
static StreamWs stream_ws(const gf3_ctx* c, int64_t n) {
    StreamWs w;
    w.plen = n + c->Lc - 1; w.nz = w.plen - 2;
    w.nb_max = ((w.plen + c->stream_plan.Lp - 1) / c->stream_plan.Lp + OLS_B - 1) / OLS_B;      // one partial maximum per ols workgroup
    w.nb_c = (w.nz + PK_THREADS * PK_ITEMS - 1) / (PK_THREADS * PK_ITEMS);
    if (w.nb_c < 1) w.nb_c = 1;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    w.o_P = take((size_t)w.plen * 8);
    w.o_part = take((size_t)w.nb_max * 8);
    w.o_cnt = take((size_t)w.nb_c * 8);
    w.o_off = take((size_t)w.nb_c * 8);
    w.o_misc = take(64);                                 // (ahead of the lists: nothing that grows can reach it)
    // Every lag can be a candidate: the rule admits minima and flat runs, so a constant or DC-biased stream
    // (u8 silence at 128) puts rounding-noise extrema above the threshold on nearly all lags of the overlap.
    w.o_cand = take((size_t)(w.nz + 2) * 8);
    w.nblk = (w.plen + c->stream_plan.Lp - 1) / c->stream_plan.Lp;
    w.nwin = w.nblk + c->stream_plan.Q - 1;
    w.o_spec = take((size_t)w.nwin * (c->stream_plan.NC + 1) * sizeof(cplx));
    w.s_nblk = w.s_ncell = w.s_nwg = w.s_cap = 0;
    if (c->scr.ok) {
        w.s_nblk = (w.plen + c->scr.H - 1) / c->scr.H;
        w.s_ncell = (w.nz + GF3_SCR_CELL - 1) / GF3_SCR_CELL;
        if (w.s_ncell < 1) w.s_ncell = 1;
        w.s_nwg = (w.s_ncell + 64 * SCR_LIST_SEGS - 1) / (64 * SCR_LIST_SEGS);
        // work list: a sixteenth of all cells (a clean stream lists two or three cells per chirp, one chirp per > 5 Lc
        // samples = 25 cells at the very least), never fewer than 4096
        w.s_cap = w.s_ncell / 16 > 4096 ? w.s_ncell / 16 : 4096;
        w.o_sblk = take((size_t)w.s_nblk * 8);                 // blk_max | blk_err (float each)
        w.o_smisc = take(sizeof(ScrMisc) + 16);                // (+ the screen's running lower bound of the maximum)
        w.o_segm = take((size_t)w.s_nwg * SCR_LIST_SEGS * 8);  // hit mask per segment of 64 cells
        w.o_cell = take((size_t)w.s_cap * 8);
        w.o_cval = take((size_t)w.s_cap * 16 * 8);
        w.o_mask = take((size_t)w.s_cap * 4);
        w.o_ccnt = take((size_t)w.s_cap * 8);
        w.o_coff = take((size_t)w.s_cap * 8);
        if ((size_t)w.s_nwg > (size_t)w.nb_c) {                // (cannot happen: 3584 lags per list workgroup vs 2048 per candidate block)
            w.s_nblk = 0;
        }
    }
    w.total = o;
    return w;
}
extern "C" int64_t gf3_sync_stream_workspace_bytes(const gf3_ctx* c, int64_t n) {
    if (!c || n < 1) return 0;
    return (int64_t)stream_ws(c, n).total;
}

// Screened path of gf3_sync_stream (gf3rx_screen.h).  Enqueues everything on `st`; the caller reads back
// {peaks, suppression status} at np and the ScrMisc block.
static int sync_stream_screened(const gf3_ctx* c, const void* d_r, int64_t n, const StreamWs& w, char* base, int64_t* d_peaks,
                                int64_t cap, int mode, hipStream_t st) {
    const auto& sp = c->scr;
    float* P32 = (float*)(base + w.o_P);
    float* blk_max = (float*)(base + w.o_sblk);
    float* blk_err = blk_max + w.s_nblk;
    ScrMisc* misc = (ScrMisc*)(base + w.o_smisc);
    int64_t* cnt = (int64_t*)(base + w.o_cnt);
    int64_t* offs = (int64_t*)(base + w.o_off);
    int64_t* total = (int64_t*)&misc->total;
    int64_t* np = (int64_t*)misc->np;
    int64_t* cand = (int64_t*)(base + w.o_cand);
    unsigned long long* segm = (unsigned long long*)(base + w.o_segm);
    int64_t* cell = (int64_t*)(base + w.o_cell);
    double* cval = (double*)(base + w.o_cval);
    unsigned* mask = (unsigned*)(base + w.o_mask);
    int64_t* ccnt = (int64_t*)(base + w.o_ccnt);
    int64_t* coff = (int64_t*)(base + w.o_coff);
    const int dt = c->cfg.in_dtype;
    // (the scalars of the call, and behind them the screen's running lower bound of the maximum: float bits, 0 = none yet)
    HIPCHK(c, hipMemsetAsync(misc, 0, sizeof(ScrMisc) + 16, st));
    {   // 1. every lag in fp32, with a bound per block
        ScreenArgs a{d_r, n, dt, sp.d_tw, sp.d_twn, sp.d_Hs, sp.d_H0N, sp.d_Hinf, sp.Q, sp.H, c->Lc, w.s_nblk, w.plen,
                     P32, blk_max, blk_err, (int*)(base + w.o_smisc + sizeof(ScrMisc)), (float)c->cfg.thresh, nullptr, nullptr, 0,
                     (unsigned long long*)&misc->status};
        HIPCHK(c, launch_screen(c, a, mode == 3, st));
    }
    // 2. the cells whose lags the bounds cannot exclude, in ascending order (flag + count, scan, scatter)
    {
        int64_t g = (w.s_nblk + SCR_MLO_THREADS * 4 - 1) / (SCR_MLO_THREADS * 4);
        g = g < 1 ? 1 : (g > 128 ? 128 : g);
        hipLaunchKernelGGL(scr_mlo_kernel, dim3((unsigned)g), dim3(SCR_MLO_THREADS), 0, st, (const float*)blk_max, (const float*)blk_err, w.s_nblk, misc,
                           (double)c->cfg.thresh);
    }
    hipLaunchKernelGGL(scr_flag_kernel, dim3((unsigned)w.s_nwg), dim3(SCR_LIST_THREADS), 0, st, (const float*)P32, (const float*)blk_max,
                       (const float*)blk_err, sp.H, w.plen, w.s_ncell, (const ScrMisc*)misc, segm, cnt);
    hipLaunchKernelGGL(pk_scan, dim3(1), dim3(1024), 0, st, (const int64_t*)cnt, w.s_nwg, offs, total, (const long long*)nullptr, (const long long*)nullptr);
    hipLaunchKernelGGL(scr_scatter_kernel, dim3((unsigned)w.s_nwg), dim3(64), 0, st, (const unsigned long long*)segm, (const int64_t*)cnt,
                       (const int64_t*)offs, (const int64_t*)total, misc, cell, w.s_cap);
    {   // 3. their lags in fp64, once (the maximum is kept as the cells complete); the reference's rule on those values
        RefineArgs a{d_r, n, dt, c->d_chirp, c->Lc, cell, misc, w.plen, cval, c->d_chirp_t, c->stamps};
        HIPCHK(c, launch_refine(c, a, w.s_cap, st));
    }
    hipLaunchKernelGGL(scr_decide_kernel, dim3((unsigned)((w.s_cap + 255) / 256)), dim3(256), 0, st, (const int64_t*)cell, (const double*)cval, misc,
                       w.nz, (double)c->cfg.thresh, mask, ccnt);
    // 4. candidates in ascending order, suppression walk
    hipLaunchKernelGGL(pk_scan, dim3(1), dim3(1024), 0, st, (const int64_t*)ccnt, w.s_cap, coff, total, (const long long*)&misc->ncell,
                       (const long long*)&misc->status);
    hipLaunchKernelGGL(scr_expand_kernel, dim3((unsigned)((w.s_cap + 255) / 256)), dim3(256), 0, st, (const int64_t*)cell, (const unsigned*)mask,
                       (const int64_t*)coff, (const ScrMisc*)misc, cand, w.nz + 2);
    hipLaunchKernelGGL(pk_nms, dim3(1), dim3(NMS_THREADS), 0, st, (const int64_t*)cand, (const int64_t*)total,
                       (int64_t)c->Lc, w.nz, d_peaks, cap, np);
    HIPCHK(c, hipGetLastError());
    return GF3_OK;
}

// Below this many samples the dozen small launches of the screened path cost more than the fp64 transforms they save
// (3 M-sample recording: 0.24 ms screened, 0.15 ms all-fp64; 321 M samples: 3.8 vs 7.2 ms; the lines cross near 7 M).
#define GF3_SCR_MIN_SAMPLES ((int64_t)1 << 23)
extern "C" int gf3_sync_stream_mode(gf3_ctx* c, int32_t mode) {
    if (!c || mode < 0 || mode > 3)
        return fail(c, GF3_EINVAL, "gf3_sync_stream_mode: mode must be 0 (by length), 1 (fp64 only), 2 (always screen) or 3 (always screen, general kernel)");
    c->default_stream_mode.store(mode, std::memory_order_relaxed);
    return GF3_OK;
}
extern "C" int gf3_sync_stream_info(const gf3_ctx* c, int64_t* h_out4) {
    if (!c || !h_out4) return fail(c, GF3_EINVAL, "null argument");
    memcpy(h_out4, g_last_info, sizeof(g_last_info));
    return GF3_OK;
}
// tests: the screening pass alone.  d_p32 [n + Lc - 1] float, d_blk [2 * nblk] float (block maxima, then block error
// bounds), *h_hop = lags per block.
extern "C" int gf3_debug_stream_screen(gf3_ctx* c, const void* d_r, int64_t n, float* d_p32, float* d_blk, int32_t* h_hop, void* stream) {
    DeviceGuard dg(c);
    if (!c || !d_r || !d_p32 || !d_blk || !h_hop || n < 3) return fail(c, GF3_EINVAL, "gf3_debug_stream_screen: bad argument");
    if (!c->scr.ok) return fail(c, GF3_EINVAL, "gf3_debug_stream_screen: no screening plan for this geometry");
    const auto& sp = c->scr;
    const int64_t plen = n + c->Lc - 1, nblk = (plen + sp.H - 1) / sp.H;
    *h_hop = sp.H;
    ScreenArgs a{d_r, n, c->cfg.in_dtype, sp.d_tw, sp.d_twn, sp.d_Hs, sp.d_H0N, sp.d_Hinf, sp.Q, sp.H, c->Lc, nblk, plen,
                 d_p32, d_blk, d_blk + nblk, nullptr, 0.0f, nullptr, nullptr, 0, nullptr};      // (no skipping: the tests look at every lag)
    HIPCHK(c, launch_screen(c, a, c->default_stream_mode.load(std::memory_order_relaxed) == 3, (hipStream_t)stream));
    return GF3_OK;
}

extern "C" int gf3_sync_stream_ex(const gf3_ctx* c, const void* d_r, int64_t n, int64_t* d_peaks, int64_t cap,
                                  int64_t* n_peaks, void* d_work, double* d_corr, int32_t mode, int64_t* h_info4, void* stream) {
    DeviceGuard dg(c);
    if (!c || !d_r || !d_peaks || !n_peaks || !d_work || n < 3 || cap < 1 || mode < 0 || mode > 3)
        return fail(c, GF3_EINVAL, "gf3_sync_stream: bad argument (mode must be 0 by length, 1 fp64 only, 2 always screen, 3 always screen with the general kernel)");
    hipStream_t st = (hipStream_t)stream;
    const StreamWs w = stream_ws(c, n);
    char* base = (char*)d_work;
    double* P = d_corr ? d_corr : (double*)(base + w.o_P);
    double* part = (double*)(base + w.o_part);
    int64_t* cnt = (int64_t*)(base + w.o_cnt);
    int64_t* offs = (int64_t*)(base + w.o_off);
    int64_t* cand = (int64_t*)(base + w.o_cand);
    double* mx = (double*)(base + w.o_misc);
    int64_t* total = (int64_t*)(base + w.o_misc + 8);
    int64_t* np = (int64_t*)(base + w.o_misc + 16);        // [count, status]
    const CorrPlan& pl = c->stream_plan;
    // diagnostics of this call: the caller's array when given, and always the calling thread's own copy
    // (gf3_sync_stream_info); nothing of a call is kept in the context
    int64_t info_local[4];
    int64_t* info = h_info4 ? h_info4 : info_local;
    struct Publish { int64_t* i; ~Publish() { memcpy(g_last_info, i, sizeof(g_last_info)); } } publish{info};
    info[0] = 2; info[1] = info[2] = info[3] = 0;
    if (!d_corr && c->scr.ok && w.s_nblk > 0 && (mode >= 2 || (mode == 0 && n >= GF3_SCR_MIN_SAMPLES))) {
        int rc = sync_stream_screened(c, d_r, n, w, base, d_peaks, cap, mode, st);
        if (rc != GF3_OK) return rc;
        static_assert(sizeof(ScrMisc) <= 256, "readback_buffer() holds 256 bytes");
        ScrMisc hm_local;
        void* pin = readback_buffer();
        ScrMisc& hm = pin ? *(ScrMisc*)pin : hm_local;
        HIPCHK(c, hipMemcpyAsync(&hm, base + w.o_smisc, sizeof(ScrMisc), hipMemcpyDeviceToHost, st));      // (the one read-back of the call)
        HIPCHK(c, hipStreamSynchronize(st));
        const int64_t h[2] = {hm.np[0], hm.np[1]}, ncand = hm.total;
        info[1] = hm.ncell; info[2] = hm.nhit; info[3] = ncand;
        if (!(hm.status & 1)) {
            info[0] = 0;
            *n_peaks = h[0];
            if (h[1] == 2) return fail(c, GF3_ERANGE, "gf3_sync_stream: %lld peaks exceed capacity %lld", (long long)h[0], (long long)cap);
            return GF3_OK;
        }
        info[0] = 1;                                      // the screen was not selective: all-fp64 path below
    }
    {
        OlsArgs a{};
        a.t = pl.t; a.in = d_r; a.n_in = n; a.dt = c->cfg.in_dtype;
        a.Hq = pl.d_Hq; a.Q = pl.Q; a.H = pl.Lp; a.Lc = c->Lc;
        a.spec = (cplx*)(base + w.o_spec); a.nwin = w.nwin; a.plen = w.plen; a.corr = P; a.part = part;
        HIPCHK(c, run_spec_ols(pl, a, w.nwin, w.nblk, st));
    }
    hipLaunchKernelGGL(pk_max_final, dim3(1), dim3(256), 0, st, (const double*)part, (int)w.nb_max, mx);
    hipLaunchKernelGGL(pk_candidates, dim3((unsigned)w.nb_c), dim3(PK_THREADS), 0, st, (const double*)P, w.nz,
                       (const double*)mx, c->cfg.thresh, cnt, (const int64_t*)nullptr, (int64_t*)nullptr);
    hipLaunchKernelGGL(pk_scan, dim3(1), dim3(1024), 0, st, (const int64_t*)cnt, w.nb_c, offs, total, (const long long*)nullptr, (const long long*)nullptr);
    hipLaunchKernelGGL(pk_candidates, dim3((unsigned)w.nb_c), dim3(PK_THREADS), 0, st, (const double*)P, w.nz,
                       (const double*)mx, c->cfg.thresh, cnt, (const int64_t*)offs, cand);
    hipLaunchKernelGGL(pk_nms, dim3(1), dim3(NMS_THREADS), 0, st, (const int64_t*)cand, (const int64_t*)total,
                       (int64_t)c->Lc, w.nz, d_peaks, cap, np);
    HIPCHK(c, hipGetLastError());
    int64_t h_local[2] = {0, 0};
    int64_t* h = readback_buffer() ? (int64_t*)readback_buffer() : h_local;
    HIPCHK(c, hipMemcpyAsync(h, np, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    *n_peaks = h[0];
    if (h[1] == 2) return fail(c, GF3_ERANGE, "gf3_sync_stream: %lld peaks exceed capacity %lld", (long long)h[0], (long long)cap);
    return GF3_OK;
}

// the legacy entry point: the context's default mode (gf3_sync_stream_mode), diagnostics through gf3_sync_stream_info
extern "C" int gf3_sync_stream(gf3_ctx* c, const void* d_r, int64_t n, int64_t* d_peaks, int64_t cap,
                               int64_t* n_peaks, void* d_work, double* d_corr, void* stream) {
    if (!c) return fail(c, GF3_EINVAL, "gf3_sync_stream: bad argument");
    return gf3_sync_stream_ex(c, d_r, n, d_peaks, cap, n_peaks, d_work, d_corr, c->default_stream_mode.load(std::memory_order_relaxed), nullptr, stream);
}

// ============================================================================
// Chunked stream sync: chirp_method (OFDM.py:356-372) on a stream that arrives piece by piece (host ingest, streams
// longer than HBM) with the EXACT global rule.  The threshold of the reference is relative to the maximum of the WHOLE
// stream (:359), which is only known at the end; so every piece keeps, next to the running maximum, the few lags that
// could still pass whatever the final maximum turns out to be -- P[g] >= thresh * (maximum so far) * (1 - 1e-6); the
// final maximum can only be larger -- together with the three raw fp64 values P[g-1], P[g], P[g+1] the rule looks at.
// gf3_sync_decide then applies the rule literally (division by the maximum first, extremum test, threshold) to those
// raw values, with the final maximum or, provisionally, with the maximum so far, and walks the suppression.
// ============================================================================
#define CK_THREADS 256
__global__ __launch_bounds__(CK_THREADS) void ck_max_kernel(const double* __restrict__ P, int64_t lo, int64_t hi, double* part) {
    __shared__ double scratch[16];
    double mx = -INFINITY;
    bool nan = false;
    for (int64_t i = lo + (int64_t)blockIdx.x * CK_THREADS + threadIdx.x; i < hi; i += (int64_t)gridDim.x * CK_THREADS) {
        const double v = P[i];
        mx = fmax(mx, v);
        nan = nan || !(v == v);
    }
    mx = block_max(mx, scratch);
    const int anynan = __syncthreads_or(nan ? 1 : 0);                   // np.amax propagates NaN (OFDM.py:359)
    if (threadIdx.x == 0) part[blockIdx.x] = anynan ? NAN : mx;
}
// run_max[0] = amax(run_max[0], part[0..n)) with NumPy's NaN rule; run_max[1] = amax(part[0..n)): this piece's own maximum
__global__ void ck_fold_max(const double* part, int n, double* run_max) {
    __shared__ double scratch[16];
    double mx = -INFINITY;
    bool nan = false;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const double v = part[i]; mx = fmax(mx, v); nan = nan || !(v == v); }
    mx = block_max(mx, scratch);
    const int anynan = __syncthreads_or(nan ? 1 : 0);
    if (threadIdx.x == 0) {
        const double run = run_max[0];
        run_max[0] = (anynan || !(run == run)) ? NAN : fmax(run, mx);
        run_max[1] = anynan ? NAN : mx;
    }
}
// pass 0 (offsets == nullptr): count per block; pass 1: write zeros-index g - 1 + lag_offset and the raw triple of every
// listed lag g in [lo, hi), ascending
__global__ __launch_bounds__(PK_THREADS) void ck_list_kernel(const double* __restrict__ P, int64_t lo, int64_t hi, const double* mxp, double thresh,
                                                             int64_t lag_offset, int64_t* counts, const int64_t* offsets, int64_t* idx, double* val3) {
    __shared__ int wsum[PK_THREADS / 64];
    if (offsets && counts[blockIdx.x] == 0) return;
    const double mx = mxp[0];
    const bool filt = mx > 0.0 && thresh > 0.0 && mx < INFINITY && thresh < INFINITY;
    const double lim = filt ? thresh * mx * (1.0 - 1e-6) : -INFINITY;   // (no positive finite maximum yet: every lag stays listed)
    const int64_t base = lo + ((int64_t)blockIdx.x * PK_THREADS + threadIdx.x) * PK_ITEMS;
    int c = 0;
    unsigned flags = 0;
#pragma unroll
    for (int k = 0; k < PK_ITEMS; ++k)
        if (base + k < hi && !(P[base + k] < lim)) { flags |= 1u << k; ++c; }
    int x = c;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int woff = 0, total = 0;
    for (int w = 0; w < PK_THREADS / 64; ++w) { if (w < wave) woff += wsum[w]; total += wsum[w]; }
    if (!offsets) { if (threadIdx.x == 0) counts[blockIdx.x] = total; return; }
    int64_t o = offsets[blockIdx.x] + woff + (x - c);
    for (int k = 0; k < PK_ITEMS; ++k)
        if (flags & (1u << k)) {
            const int64_t g = base + k;
            idx[o] = g - 1 + lag_offset;
            val3[3 * o] = P[g - 1]; val3[3 * o + 1] = P[g]; val3[3 * o + 2] = P[g + 1];
            ++o;
        }
}
// the reference's rule on the listed raw values (OFDM.py:359-361): p = P / max first, then
// (p1 - p0)(p2 - p1) <= 0 and p1 > thresh; survivors compacted in order.  One workgroup.
__global__ __launch_bounds__(1024) void ck_decide_kernel(const int64_t* idx, const double* val3, int64_t n, const double* mxp, double thresh,
                                                         int64_t* cand, int64_t* total) {
    __shared__ int wsum[16];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const double mx = mxp[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t b0 = 0; b0 < n; b0 += 1024) {
        const int64_t i = b0 + threadIdx.x;
        int f = 0;
        if (i < n) {
            const double p0 = val3[3 * i] / mx, p1 = val3[3 * i + 1] / mx, p2 = val3[3 * i + 2] / mx;
            f = (((p1 - p0) * (p2 - p1) <= 0.0) && (p1 > thresh)) ? 1 : 0;
        }
        int x = f;
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) woff += wsum[w]; tot += wsum[w]; }
        if (f) cand[carry + woff + (x - 1)] = idx[i];
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = carry;
}

extern "C" int64_t gf3_sync_chunk_workspace_bytes(const gf3_ctx* c, int64_t n) { return gf3_sync_stream_workspace_bytes(c, n); }

extern "C" int gf3_sync_chunk(const gf3_ctx* c, const void* d_buf, int64_t n, int64_t lag_lo, int64_t lag_hi, int64_t lag_offset,
                              double* d_run_max, int64_t* d_idx, double* d_val3, int64_t cap, int64_t* n_listed,
                              double* h_piece_max, void* d_work, void* stream) {
    DeviceGuard dg(c);
    // (cap == 0 with no list buffers is a legitimate call: the caller's list is full, the piece then reports how many
    //  lags it WOULD keep through GF3_ERANGE, or nothing when it keeps none)
    if (!c || !d_buf || !d_run_max || !n_listed || !d_work || n < 3 || cap < 0 || (cap > 0 && (!d_idx || !d_val3)))
        return fail(c, GF3_EINVAL, "gf3_sync_chunk: bad argument");
    const StreamWs w = stream_ws(c, n);
    if (lag_lo < 1 || lag_hi > w.plen - 1 || lag_lo > lag_hi)
        return fail(c, GF3_EINVAL, "gf3_sync_chunk: lags [%lld, %lld) outside [1, %lld)", (long long)lag_lo, (long long)lag_hi, (long long)(w.plen - 1));
    *n_listed = 0;
    if (h_piece_max) *h_piece_max = -INFINITY;
    if (lag_lo == lag_hi) return GF3_OK;
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)d_work;
    double* P = (double*)(base + w.o_P);
    double* part = (double*)(base + w.o_part);
    int64_t* cnt = (int64_t*)(base + w.o_cnt);
    int64_t* offs = (int64_t*)(base + w.o_off);
    int64_t* total = (int64_t*)(base + w.o_misc + 8);
    const CorrPlan& pl = c->stream_plan;
    {   // P of the whole buffer, all fp64 (the overlap-save of gf3_sync_stream's fp64 path)
        OlsArgs a{};
        a.t = pl.t; a.in = d_buf; a.n_in = n; a.dt = c->cfg.in_dtype;
        a.Hq = pl.d_Hq; a.Q = pl.Q; a.H = pl.Lp; a.Lc = c->Lc;
        a.spec = (cplx*)(base + w.o_spec); a.nwin = w.nwin; a.plen = w.plen; a.corr = P; a.part = part;
        HIPCHK(c, run_spec_ols(pl, a, w.nwin, w.nblk, st));
    }
    // the maximum of the lags this piece owns joins the running maximum (the ols workgroups' own maxima cover lags at
    // the buffer's edges whose sums are cut off: they are not values of the stream's P)
    int64_t gmax = (lag_hi - lag_lo + CK_THREADS * 8 - 1) / (CK_THREADS * 8);
    gmax = gmax < 1 ? 1 : (gmax > w.nb_max ? w.nb_max : (gmax > 2048 ? 2048 : gmax));
    hipLaunchKernelGGL(ck_max_kernel, dim3((unsigned)gmax), dim3(CK_THREADS), 0, st, (const double*)P, lag_lo, lag_hi, part);
    hipLaunchKernelGGL(ck_fold_max, dim3(1), dim3(256), 0, st, (const double*)part, (int)gmax, d_run_max);
    const int64_t nb = (lag_hi - lag_lo + PK_THREADS * PK_ITEMS - 1) / (PK_THREADS * PK_ITEMS);     // <= nb_c of the workspace
    hipLaunchKernelGGL(ck_list_kernel, dim3((unsigned)nb), dim3(PK_THREADS), 0, st, (const double*)P, lag_lo, lag_hi, (const double*)d_run_max,
                       c->cfg.thresh, lag_offset, cnt, (const int64_t*)nullptr, (int64_t*)nullptr, (double*)nullptr);
    hipLaunchKernelGGL(pk_scan, dim3(1), dim3(1024), 0, st, (const int64_t*)cnt, nb, offs, total, (const long long*)nullptr, (const long long*)nullptr);
    HIPCHK(c, hipGetLastError());
    int64_t want_local = 0;
    double pmax_local = -INFINITY;
    void* pin = readback_buffer();
    int64_t& want = pin ? *(int64_t*)pin : want_local;
    double& pmax = pin ? *(double*)((char*)pin + 8) : pmax_local;
    HIPCHK(c, hipMemcpyAsync(&want, total, 8, hipMemcpyDeviceToHost, st));
    if (h_piece_max) HIPCHK(c, hipMemcpyAsync(&pmax, d_run_max + 1, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    *n_listed = want;
    if (h_piece_max) *h_piece_max = pmax;
    if (want > cap) return fail(c, GF3_ERANGE, "gf3_sync_chunk: %lld lags to keep exceed capacity %lld", (long long)want, (long long)cap);
    if (want > 0) {
        hipLaunchKernelGGL(ck_list_kernel, dim3((unsigned)nb), dim3(PK_THREADS), 0, st, (const double*)P, lag_lo, lag_hi, (const double*)d_run_max,
                           c->cfg.thresh, lag_offset, cnt, (const int64_t*)offs, d_idx, d_val3);
        HIPCHK(c, hipGetLastError());
    }
    return GF3_OK;
}

extern "C" int64_t gf3_sync_decide_workspace_bytes(const gf3_ctx* c, int64_t n_listed) {
    if (!c || n_listed < 0) return 0;
    return (int64_t)((size_t)(n_listed + 2) * 8 + 64);
}

extern "C" int gf3_sync_decide(const gf3_ctx* c, const int64_t* d_idx, const double* d_val3, int64_t n_listed, const double* d_max,
                               int64_t nz_total, int64_t* d_peaks, int64_t cap, int64_t* n_peaks, void* d_work, void* stream) {
    DeviceGuard dg(c);
    if (!c || !d_max || !d_peaks || !n_peaks || !d_work || n_listed < 0 || cap < 1 || (n_listed > 0 && (!d_idx || !d_val3)))
        return fail(c, GF3_EINVAL, "gf3_sync_decide: bad argument");
    hipStream_t st = (hipStream_t)stream;
    int64_t* cand = (int64_t*)d_work;
    int64_t* total = cand + n_listed + 1;
    int64_t* np = total + 1;                               // [count, status]  (64 bytes of slack behind the list)
    hipLaunchKernelGGL(ck_decide_kernel, dim3(1), dim3(1024), 0, st, d_idx, d_val3, n_listed, d_max, c->cfg.thresh, cand, total);
    hipLaunchKernelGGL(pk_nms, dim3(1), dim3(NMS_THREADS), 0, st, (const int64_t*)cand, (const int64_t*)total,
                       (int64_t)c->Lc, nz_total, d_peaks, cap, np);
    HIPCHK(c, hipGetLastError());
    int64_t h_local[2] = {0, 0};
    int64_t* h = readback_buffer() ? (int64_t*)readback_buffer() : h_local;
    HIPCHK(c, hipMemcpyAsync(h, np, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    *n_peaks = h[0];
    if (h[1] == 2) return fail(c, GF3_ERANGE, "gf3_sync_decide: %lld peaks exceed capacity %lld", (long long)h[0], (long long)cap);
    return GF3_OK;
}
