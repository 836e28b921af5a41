// libgf3rx -- the chirp matched filter in fp64: one search window per workgroup (frames), and the uniformly
// partitioned overlap-save over whole streams (the fallback and `d_corr` path of gf3_sync_stream, and gf3_sync_chunk).
#include "gf3rx_host.h"

// ============================================================================
// chirp matched filter by partitioned FFT correlation, one search window per workgroup
// (convolve(r, chirp[::-1]) + peak rule, OFDM.py:357-361; whole streams: spec_kernel + ols_kernel)
//   corr[s] = sum_k r[s+k] c[k],  s = s0 .. s0+W-1   (== P[s+Lc-1])
//   c split into Q partitions of Lp taps; each partition's contribution is a
//   circular correlation of size N = 2NC, valid for lags < N-Lp+1.
// ============================================================================
#ifndef GF3_CORR_WPS
#define GF3_CORR_WPS 2
#endif
#ifndef GF3_CORR_PP
#define GF3_CORR_PP true
#endif
// One search window: everything corr_kernel does for window b.
template <int NC, int DT>
GF3_DEV void corr_window(const CorrArgs& a, const int64_t b, double2* smem) {
    constexpr int T = NC / 8;
    constexpr bool PP = GF3_CORR_PP && FftGeom<NC>::PINGPONG;
    cplx* lds = smem;
    double* scratch = (double*)(smem + (PP ? FftGeom<NC>::LDS_ELEMS : FftGeom<NC>::LDS_ELEMS_INPLACE));
    const int tid = threadIdx.x;
    const int64_t s0 = b * a.stride + a.win_lo;      // absolute sample index of lag 0 of this window
    const int W = a.W;                               // lags to resolve

    FftTw<NC> ft;
    ft.init(tid, a.t.tw);
    cplx wb = a.t.twn[tid];
    cplx acc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) acc[s] = cmk(0.0, 0.0);
    double accDC = 0.0, accNy = 0.0;
    const int need = a.Lp + a.Wmax - 1;           // samples of a segment that reach valid lags
    // a segment that lies wholly inside the buffer can use unguarded pair loads
    RawPair<DT> nxt[8];
    auto fetch = [&](int q) {
        const int64_t seg = s0 + (int64_t)q * a.Lp;
        const bool inside = seg >= 0 && seg + 2 * NC <= a.n_in;
        typedef typename RawT<DT>::E E;
        if (inside && need >= 14 * T) {
            // common case: the segment lies inside the buffer and only the last of the eight
            // strided loads can reach past the samples that matter (j >= need)
            const E* base = (const E*)a.in + seg;                        // wave-uniform
            const unsigned t2 = 2u * (unsigned)tid;
#pragma unroll
            for (int r = 0; r < 7; ++r) nxt[r].load_u(base, t2 + 2u * (unsigned)(r * T));
            const int j = 2 * (tid + 7 * T);
            if (j + 1 < need) nxt[7].load_u(base, (unsigned)j);
            else { nxt[7].zero(); if (j < need) nxt[7].v.a = ((const E*)a.in)[seg + j]; }
            return;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int j = 2 * (tid + r * T);
            nxt[r].zero();
            if (j < need && seg + j >= 0 && seg + j < a.n_in) nxt[r].v.a = ((const E*)a.in)[seg + j];
            if (j + 1 < need && seg + j + 1 >= 0 && seg + j + 1 < a.n_in) nxt[r].v.b = ((const E*)a.in)[seg + j + 1];
        }
    };
    fetch(0);
    cplx v[8], z0;
    for (int q = 0; q < a.Q; ++q) {
        const cplx* Hq = a.Hq + (int64_t)q * (NC + 1);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = nxt[r].get();
        if (q + 1 < a.Q) fetch(q + 1);
        const int tq = tid;
        // this partition's chirp spectrum, requested before the transform: the barriers' memory clobber keeps the
        // compiler from moving these loads up itself, and their L2 latency would sit between transform and MAC
        cplx hq[8];
        double h0 = 0.0, hN = 0.0;
        constexpr bool HOIST = (NC <= 1024 && DT != DT_F64);      // (f64 samples: the prefetch needs the registers)
        if constexpr (!HOIST) {
#pragma unroll
            for (int s = 0; s < 8; ++s) hq[s] = Hq[Spec<NC>::bin(tq, s)];
            if (tid == 0) { h0 = Hq[0].x; hN = Hq[NC].x; }
            ft.refresh();
        }
        // Frames plan with narrow samples (HOIST): nothing is made opaque, so every twiddle power and the split's pair twiddles are
        // loop invariants kept in registers across the partition loop (~45 registers, ~30 fp64 operations per
        // partition saved); the registers come from loading the chirp spectrum after the transform instead of before.
        rfft_regs<NC, PP, true>(v, lds, ft, wb, tq, z0, q & 1);       // slots hold 2 X: undone by `inv` below
        if constexpr (HOIST) {
#pragma unroll
            for (int s = 0; s < 8; ++s) hq[s] = Hq[Spec<NC>::bin(tq, s)];
            if (tid == 0) { h0 = Hq[0].x; hN = Hq[NC].x; }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[s] = cfma(v[s], cconj(hq[s]), acc[s]);        // acc += v conj(h): four fma
        if (tid == 0) {
            accDC += (z0.x + z0.y) * h0;
            accNy += (z0.x - z0.y) * hN;
        }
    }
    // ---- inverse real FFT of the accumulated Hermitian spectrum Y
    lds_barrier();
    {
        const int tq = launder(tid);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = Spec<NC>::bin(tq, 2 * r);
            const cplx A = acc[2 * r];
            const cplx B = cconj(acc[2 * r + 1]);
            const cplx E = cscale(cadd(A, B), 0.5);
            const cplx Op = cmul_conj(cscale(csub(A, B), 0.5), Spec<NC>::pair_tw(tq, r, wb));   // * exp(+2 pi i k/N)
            const cplx Zk = cadd(E, mul_posi(Op));
            const cplx Zm = cadd(cconj(E), mul_posi(cconj(Op)));
            lds[k] = cconj(Zk);
            if (Spec<NC>::live(tq, 2 * r + 1)) lds[NC - k] = cconj(Zm);
        }
    }
    if (tid == 0) {
        const double E = accDC + accNy, Op = accDC - accNy;                 // 2 x (E, Op): same scale as the slots
        lds[0] = cmk(E, -Op);                                               // conj(E + i Op)
    }
    lds_barrier();
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = lds[tid + r * T];
    lds_barrier();                                   // everyone holds its inputs: both buffers are free
    cplx* yb = fft_core<NC, PP>(v, lds, ft, launder(tid));
    // z = conj(FFT(conj Z))/NC ; y[2n] = Re z, y[2n+1] = Im z  -> in place as doubles
    const double inv = 0.5 / (double)NC;             // 1/NC of the inverse transform and the 2 of the forward ones (exact)
    for (int i = tid; i < NC; i += T) { const cplx z = yb[i]; yb[i] = cmk(z.x * inv, -z.y * inv); }
    lds_barrier();
    const double* y = (const double*)yb;

    // ---- peak rule on the window (OFDM.py:359-361): normalise by the max, first
    // local extremum above thresh
    double mx = -INFINITY;
    for (int j = tid; j < W; j += T) mx = fmax(mx, y[j]);
    mx = block_max(mx, scratch);
    int first = 0x7fffffff;
    // Lags that cannot reach the threshold skip the three divisions (y < thresh*max*(1-1e-6) implies
    // fl(y/max) < thresh; same prefilter as pk_candidates): only the few lags around the peak pay for them.
    const bool filt = mx > 0.0 && a.thresh > 0.0 && mx < INFINITY && a.thresh < INFINITY;
    const double lim = filt ? a.thresh * mx * (1.0 - 1e-6) : -INFINITY;
    for (int j = 1 + tid; j < W - 1; j += T) {
        const double y0 = y[j];
        if (!(y0 < lim)) {
            const double pm1 = y[j - 1] / mx, p0 = y0 / mx, pp1 = y[j + 1] / mx;
            if (((p0 - pm1) * (pp1 - p0) <= 0.0) && (p0 > a.thresh)) first = min(first, j);
        }
    }
    first = block_min_i(first, (int*)(scratch + 16));
    if (tid == 0) {
        const bool found = first != 0x7fffffff;
        a.starts[b] = found ? (s0 + first + a.Lc) : -1;
        if (a.peak) a.peak[b] = found ? y[first] : 0.0;
    }
}
// LISTED: the windows are those of a device-side list (the screened sync's unresolved windows, gf3rx_fscreen.h), whose
// length lives on the device: the grid is the list's capacity and workgroups past its length return at once (65 536 empty
// workgroups are ~25 us; a persistent grid walking the list was tried -- the loop's invariants spill 4-36 registers of a
// kernel that has none to spare).  The plain instantiation is the kernel it was.
template <int NC, int DT, bool LISTED = false>
__global__ __launch_bounds__(NC / 8, (NC <= 2048 ? GF3_CORR_WPS : 2)) void corr_kernel(CorrArgs a) {
    extern __shared__ double2 smem[];
    if constexpr (!LISTED) corr_window<NC, DT>(a, blockIdx.x, smem);
    else {
        if ((int)blockIdx.x >= *a.count) return;
        corr_window<NC, DT>(a, a.list[blockIdx.x], smem);
    }
}

template <int NC, int DT>
__global__ __launch_bounds__(NC / 8, 2) void spec_kernel(OlsArgs a) {
    extern __shared__ double2 smem[];
    constexpr int T = NC / 8;
    typedef typename RawT<DT>::E E;
    const int tid = threadIdx.x;
    const int64_t j = xcd_order(blockIdx.x, gridDim.x);     // neighbouring windows overlap by N - H samples: same XCD, same L2
    if (j >= a.nitems) return;
    const int64_t seg = j * (int64_t)a.H - (a.Lc - 1);
    FftTw<NC> ft;
    ft.init(tid, a.t.tw);
    const cplx wb = a.t.twn[tid];
    cplx v[8];
    const bool inside = seg >= 0 && seg + 2 * NC <= a.n_in;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int64_t i = seg + 2 * (int64_t)(tid + r * T);
        RawPair<DT> raw;
        if (inside) raw.load(a.in, i);
        else {
            raw.zero();
            if (i >= 0 && i < a.n_in) raw.v.a = ((const E*)a.in)[i];
            if (i + 1 >= 0 && i + 1 < a.n_in) raw.v.b = ((const E*)a.in)[i + 1];
        }
        v[r] = raw.get();
    }
    cplx z0;
    rfft_regs<NC>(v, smem, ft, wb, tid, z0, 0);
    cplx* out = a.spec + j * (int64_t)(NC + 1);
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2)
        if (Spec<NC>::live(tid, s2)) out[Spec<NC>::bin(tid, s2)] = v[s2];
    if (tid == 0) {
        out[0] = cmk(z0.x + z0.y, 0.0);
        out[NC] = cmk(z0.x - z0.y, 0.0);
    }
}

template <int NC>
__global__ __launch_bounds__(NC / 8, 2) void ols_kernel(OlsArgs a) {
    // OLS_B adjacent output blocks per workgroup: blocks b .. b+B-1 need windows b .. b+Q+B-2 and share most of
    // them, so every window spectrum is fetched once for up to B MACs (the kernel is bound by those reads).
    extern __shared__ double2 smem[];
    constexpr int T = NC / 8, B = OLS_B;
    cplx* lds = smem;
    const int tid = threadIdx.x;
    const int64_t item = xcd_order(blockIdx.x, gridDim.x);  // neighbouring groups share Q - 1 of their windows: same XCD, same L2
    if (item >= a.nitems) return;
    const int64_t b = B * item;
    FftTw<NC> ft;
    ft.init(tid, a.t.tw);
    cplx wb = a.t.twn[tid];
    cplx acc[B][8];
    double dc[B], ny[B];
#pragma unroll
    for (int g = 0; g < B; ++g) {
        dc[g] = ny[g] = 0.0;
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[g][s] = cmk(0.0, 0.0);
    }
    for (int q = 0; q < a.Q + B - 1; ++q) {          // window b+q feeds block b+g with H_{q-g}
        if (!(b + q < a.nwin)) break;
        const cplx* X = a.spec + (b + q) * (int64_t)(NC + 1);
        cplx x[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) x[s] = X[Spec<NC>::bin(tid, s)];
        double x0 = 0.0, xn = 0.0;
        if (tid == 0) { x0 = X[0].x; xn = X[NC].x; }
#pragma unroll
        for (int g = 0; g < B; ++g) {
            const int h = q - g;
            if (h >= 0 && h < a.Q) {
                const cplx* H = a.Hq + (int64_t)h * (NC + 1);
#pragma unroll
                for (int s = 0; s < 8; ++s) acc[g][s] = cfma(x[s], cconj(H[Spec<NC>::bin(tid, s)]), acc[g][s]);
                if (tid == 0) { dc[g] += x0 * H[0].x; ny[g] += xn * H[NC].x; }
            }
        }
    }
    const double inv = 1.0 / (double)NC;
    double mx = -INFINITY;                            // max of the lags this workgroup writes (OFDM.py:359 needs max(P))
    bool nan = false;
#pragma unroll
    for (int g = 0; g < B; ++g) {
        const int64_t m0 = (b + g) * (int64_t)a.H;
        if (m0 < a.plen) {
            lds_barrier();                                // previous output fully read out of LDS
            // inverse real FFT of the Hermitian spectrum (same construction as corr_kernel)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = Spec<NC>::bin(tid, 2 * r);
                const cplx A = acc[g][2 * r];
                const cplx Bm = cconj(acc[g][2 * r + 1]);
                const cplx E = cscale(cadd(A, Bm), 0.5);
                const cplx Op = cmul_conj(cscale(csub(A, Bm), 0.5), Spec<NC>::pair_tw(tid, r, wb));
                const cplx Zk = cadd(E, mul_posi(Op));
                const cplx Zm = cadd(cconj(E), mul_posi(cconj(Op)));
                lds[k] = cconj(Zk);
                if (Spec<NC>::live(tid, 2 * r + 1)) lds[NC - k] = cconj(Zm);
            }
            if (tid == 0) lds[0] = cmk(0.5 * (dc[g] + ny[g]), -0.5 * (dc[g] - ny[g]));
            lds_barrier();
            cplx v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = lds[tid + r * T];
            lds_barrier();
            ft.refresh();
            cplx* yb = fft_core<NC>(v, lds, ft, tid);
            const int64_t left = a.plen - m0;
            const int W = left < a.H ? (int)left : a.H;
            for (int i = tid; 2 * i < W; i += T) {        // y[2n] = Re z / NC, y[2n+1] = -Im z / NC
                const cplx z = yb[i];
                const double y0 = z.x * inv, y1 = -z.y * inv;
                a.corr[m0 + 2 * i] = y0;
                mx = fmax(mx, y0);
                nan = nan || !(y0 == y0);
                if (2 * i + 1 < W) { a.corr[m0 + 2 * i + 1] = y1; mx = fmax(mx, y1); nan = nan || !(y1 == y1); }
            }
        }
    }
    mx = block_max(mx, (double*)lds);                 // (starts with a barrier: every wave is done reading yb)
    const int anynan = __syncthreads_or(nan ? 1 : 0);  // np.amax propagates NaN (OFDM.py:359): so does this maximum
    if (tid == 0) a.part[item] = anynan ? NAN : mx;
}

hipError_t run_corr(const gf3_ctx* c, const CorrPlan& pl, const CorrArgs& a, int64_t grid, hipStream_t st, bool listed) {
    const int NCp = pl.NC;
    const size_t lds = (GF3_CORR_PP ? fft_lds_bytes(NCp) : (size_t)(NCp + NCp / 8) * sizeof(cplx)) + 32 * sizeof(double);
    hipError_t e = hipSuccess;
#ifdef GF3_DEV_BUILD
    if (NCp == 1024 && !listed) {
        if (a.dt == DT_F64) return launch((corr_kernel<1024, DT_F64>), grid, 128, lds, st, a);
        return launch((corr_kernel<1024, DT_F32>), grid, 128, lds, st, a);
    }
#endif
    if (listed) { DISPATCH_NC(NCp, a.dt, e = launch((corr_kernel<NCC, DTC, true>), grid, NCC / 8, lds, st, a)); }
    else { DISPATCH_NC(NCp, a.dt, e = launch((corr_kernel<NCC, DTC>), grid, NCC / 8, lds, st, a)); }
    return e;
}

// spec_kernel over every window, then ols_kernel over groups of OLS_B output blocks (both grids padded to the 8 XCDs)
hipError_t run_spec_ols(const CorrPlan& pl, OlsArgs a, int64_t nwin, int64_t nblk, hipStream_t st) {
    const size_t lds = fft_lds_bytes(pl.NC);
    hipError_t e = hipSuccess;
    auto pad8 = [](int64_t x) { return (x + 7) / 8 * 8; };              // grids padded to the 8 XCDs (xcd_order)
    a.nitems = nwin;
    DISPATCH_NC(pl.NC, a.dt, e = launch((spec_kernel<NCC, DTC>), pad8(nwin), NCC / 8, lds, st, a));
    if (e != hipSuccess) return e;
    a.nitems = (nblk + OLS_B - 1) / OLS_B;
    switch (pl.NC) {
#ifndef GF3_DEV_BUILD
        case 512:  e = launch(ols_kernel<512>, pad8(a.nitems), 64, lds, st, a); break;
        case 1024: e = launch(ols_kernel<1024>, pad8(a.nitems), 128, lds, st, a); break;
        case 4096: e = launch(ols_kernel<4096>, pad8(a.nitems), 512, lds, st, a); break;
#endif
        default:   e = launch(ols_kernel<2048>, pad8(a.nitems), 256, lds, st, a); break;
    }
    return e;
}
