// libgf3rx -- MI355X (gfx950) OFDM receive-path engine: context, plans and the C ABI (include/gf3rx.h) of everything
// but the stream sync (gf3rx_sync.hip); the small stand-alone kernels (demappers, zero forcing, Schmidl-Cox) live here too.
// See DESIGN.md for the layout and gf3rx_host.h for the map of translation units.
#include "gf3rx_demod.h"
#include "gf3rx_fscreen.h"

// Message of the calling thread's last failure.  One buffer per host thread, none in the context: concurrent calls
// on one context (different streams, different threads) cannot overwrite each other's text, and a failing call
// writes nothing into the context it was given.
static thread_local char g_err[512] = "";
int fail(const gf3_ctx*, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// Scratch device allocations of the set-up helpers: released on every return path.
struct DevTmp {
    std::vector<void*> p;
    ~DevTmp() { for (void* q : p) if (q) (void)hipFree(q); }
    template <typename T> hipError_t alloc(T** out, size_t n) {
        hipError_t e = hipMalloc((void**)out, n * sizeof(T));
        if (e == hipSuccess) p.push_back((void*)*out);
        return e;
    }
    template <typename T> hipError_t put(T** out, const T* h, size_t n) {
        hipError_t e = alloc(out, n);
        return e != hipSuccess ? e : hipMemcpy(*out, h, n * sizeof(T), hipMemcpyHostToDevice);
    }
};


template <typename T> static hipError_t upload(T** dptr, const T* h, size_t n) {
    hipError_t e = hipMalloc((void**)dptr, n * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*dptr, h, n * sizeof(T), hipMemcpyHostToDevice);
}


// ============================================================================
// standalone demappers
// ============================================================================
struct DemapArgs {
    const cplx* sym; int64_t n; int M, mu;
    const double* cre; const double* cim; const int* clab;
    uint8_t* bits; float* llr; double inv_nv; uint8_t* idx;
    SepTab sep;
};
__global__ void demap_hard_kernel(DemapArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        const cplx e = a.sym[i];
        const int best = scan_table(e, a.cre, a.cim, a.M);          // literal: this entry point is `demap` itself
        const int lab = a.clab[best];
        for (int b = 0; b < a.mu; ++b) a.bits[i * a.mu + b] = (lab >> (a.mu - 1 - b)) & 1;
        if (a.idx) a.idx[i] = (uint8_t)best;
    }
}
// max-log LLR per bit: (min over points with bit=1 of d^2 - min over points with bit=0 of d^2) / noise_var
__global__ void soft_demap_kernel(DemapArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        const cplx e = a.sym[i];
        double m0[8], m1[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) m0[b] = m1[b] = INFINITY;
        if (a.sep.nI > 0) {
            // separable table: a bit owned by one axis sees the other axis' term cancel in the difference
            for (int k = 0; k < a.sep.nI; ++k) {
                const double d = (e.x - a.sep.lvI[k]) * (e.x - a.sep.lvI[k]);
                const int lab = a.sep.labI[k];
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (b < a.mu && ((a.sep.maskI >> (a.mu - 1 - b)) & 1)) {
                        if ((lab >> (a.mu - 1 - b)) & 1) m1[b] = fmin(m1[b], d); else m0[b] = fmin(m0[b], d);
                    }
            }
            for (int k = 0; k < a.sep.nQ; ++k) {
                const double d = (e.y - a.sep.lvQ[k]) * (e.y - a.sep.lvQ[k]);
                const int lab = a.sep.labQ[k];
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (b < a.mu && !((a.sep.maskI >> (a.mu - 1 - b)) & 1)) {
                        if ((lab >> (a.mu - 1 - b)) & 1) m1[b] = fmin(m1[b], d); else m0[b] = fmin(m0[b], d);
                    }
            }
        } else {
            for (int c = 0; c < a.M; ++c) {
                const double dx = e.x - a.cre[c], dy = e.y - a.cim[c];
                const double d = dx * dx + dy * dy;
                const int lab = a.clab[c];
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (b < a.mu) { if ((lab >> (a.mu - 1 - b)) & 1) m1[b] = fmin(m1[b], d); else m0[b] = fmin(m0[b], d); }
            }
        }
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (b < a.mu) a.llr[i * a.mu + b] = (float)((m1[b] - m0[b]) * a.inv_nv);
    }
}

// Separable tables (grid constellations with per-axis bit labels): a bit owned by one axis sees the other axis'
// term cancel in the difference, so its LLR needs that axis' <= 8 squared distances only.  Everything that steers
// the reduction (which axis owns bit b, which levels carry a 1 there, how many levels exist) is wave-uniform and
// lives in scalar registers; the loops are fully unrolled over MU bits x 8 levels, each step one scalar bit test
// around one v_min_f64.
template <int MU>
__global__ __launch_bounds__(256) void soft_demap_sep_kernel(DemapArgs a) {
    int ones[MU];                                    // bit b: mask of the owning axis' levels whose label has a 1 there
    bool onI[MU];
#pragma unroll
    for (int b = 0; b < MU; ++b) {
        onI[b] = (a.sep.maskI >> (MU - 1 - b)) & 1;
        ones[b] = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) ones[b] |= (((onI[b] ? a.sep.labI[k] : a.sep.labQ[k]) >> (MU - 1 - b)) & 1) << k;
    }
    const int nI = a.sep.nI, nQ = a.sep.nQ;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        const cplx e = a.sym[i];
        double dI[8], dQ[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double tI = e.x - a.sep.lvI[k], tQ = e.y - a.sep.lvQ[k];
            dI[k] = tI * tI; dQ[k] = tQ * tQ;
        }
        float out[MU];
#pragma unroll
        for (int b = 0; b < MU; ++b) {
            double m0 = INFINITY, m1 = INFINITY;
            // (opaque per symbol: otherwise the 8 MU level tests are hoisted out of the symbol loop as 8 MU SGPR
            //  pairs, which spill to VGPR lanes and come back through v_readlane on every use)
            asm volatile("" : "+s"(ones[b]));
            if (onI[b]) {
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k < nI) { if ((ones[b] >> k) & 1) m1 = fmin(m1, dI[k]); else m0 = fmin(m0, dI[k]); }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k < nQ) { if ((ones[b] >> k) & 1) m1 = fmin(m1, dQ[k]); else m0 = fmin(m0, dQ[k]); }
            }
            out[b] = (float)((m1 - m0) * a.inv_nv);
        }
        if constexpr (MU % 4 == 0) {                  // 16-byte aligned rows
#pragma unroll
            for (int b = 0; b < MU; b += 4) *(float4*)(a.llr + i * MU + b) = make_float4(out[b], out[b + 1], out[b + 2], out[b + 3]);
        } else if constexpr (MU % 2 == 0) {           // 8-byte aligned rows
#pragma unroll
            for (int b = 0; b < MU; b += 2) *(float2*)(a.llr + i * MU + b) = make_float2(out[b], out[b + 1]);
        } else {
#pragma unroll
            for (int b = 0; b < MU; ++b) a.llr[i * MU + b] = out[b];
        }
    }
}

// The same for the tables every square Gray QAM generator produces (and the reference's QPSK): 2^HI x 2^HQ grid, the
// first HI label bits are the binary index of the I level in `lvI`, the last HQ bits that of the Q level.  Which
// levels carry a 1 in which bit is then known at compile time, so the whole reduction is straight-line v_min_f64 --
// no scalar bit tests, no branches (the generic kernel above spends more time steering than computing: 48 scalar
// branches per symbol against 48 minima).
template <int HI, int HQ>
__global__ __launch_bounds__(256) void soft_demap_bin_kernel(DemapArgs a) {
    constexpr int MU = HI + HQ, NI = 1 << HI, NQ = 1 << HQ;
    double lvI[NI], lvQ[NQ];
#pragma unroll
    for (int k = 0; k < NI; ++k) lvI[k] = a.sep.lvI[k];
#pragma unroll
    for (int k = 0; k < NQ; ++k) lvQ[k] = a.sep.lvQ[k];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        const cplx e = a.sym[i];
        double dI[NI], dQ[NQ];
#pragma unroll
        for (int k = 0; k < NI; ++k) { const double t = e.x - lvI[k]; dI[k] = t * t; }
#pragma unroll
        for (int k = 0; k < NQ; ++k) { const double t = e.y - lvQ[k]; dQ[k] = t * t; }
        float out[MU];
#pragma unroll
        for (int b = 0; b < HI; ++b) {                 // label bit b = bit (HI - 1 - b) of the I index
            double m0 = INFINITY, m1 = INFINITY;
#pragma unroll
            for (int k = 0; k < NI; ++k) { if ((k >> (HI - 1 - b)) & 1) m1 = fmin(m1, dI[k]); else m0 = fmin(m0, dI[k]); }
            out[b] = (float)((m1 - m0) * a.inv_nv);
        }
#pragma unroll
        for (int b = 0; b < HQ; ++b) {
            double m0 = INFINITY, m1 = INFINITY;
#pragma unroll
            for (int k = 0; k < NQ; ++k) { if ((k >> (HQ - 1 - b)) & 1) m1 = fmin(m1, dQ[k]); else m0 = fmin(m0, dQ[k]); }
            out[HI + b] = (float)((m1 - m0) * a.inv_nv);
        }
        if constexpr (MU % 4 == 0) {
#pragma unroll
            for (int b = 0; b < MU; b += 4) *(float4*)(a.llr + i * MU + b) = make_float4(out[b], out[b + 1], out[b + 2], out[b + 3]);
        } else {
#pragma unroll
            for (int b = 0; b < MU; b += 2) *(float2*)(a.llr + i * MU + b) = make_float2(out[b], out[b + 1]);
        }
    }
}
// is the separable table of that binary-indexed kind?
static bool sep_is_binary(const SepTab& sp, int mu, int& hI, int& hQ) {
    hI = hQ = 0;
    while ((1 << hI) < sp.nI) ++hI;
    while ((1 << hQ) < sp.nQ) ++hQ;
    if (sp.nI < 2 || sp.nQ < 2 || (1 << hI) != sp.nI || (1 << hQ) != sp.nQ || hI + hQ != mu || hI != hQ) return false;
    if (sp.maskI != (((1 << hI) - 1) << hQ)) return false;
    for (int k = 0; k < sp.nI; ++k) if (sp.labI[k] != (k << hQ)) return false;
    for (int k = 0; k < sp.nQ; ++k) if (sp.labQ[k] != k) return false;
    return true;
}

// spectra of the zero-padded chirp partitions, computed with the engine's own FFT
static int build_plan(gf3_ctx* c, CorrPlan* pl, int NCp, FftTables t, int Lp_max) {
    const int N = 2 * NCp;
    pl->NC = NCp; pl->t = t;
    int Q = (c->Lc + Lp_max - 1) / Lp_max;
    int Lp = (c->Lc + Q - 1) / Q;
    pl->Q = Q; pl->Lp = Lp; pl->W = N - Lp + 1;
    std::vector<double> h((size_t)Q * N, 0.0);
    for (int q = 0; q < Q; ++q)
        for (int k = 0; k < Lp && q * Lp + k < c->Lc; ++k) h[(size_t)q * N + k] = c->chirp[(size_t)q * Lp + k];
    std::vector<int64_t> off(Q);
    for (int q = 0; q < Q; ++q) off[q] = (int64_t)q * N;
    double* d_h = nullptr; int64_t* d_off = nullptr;
    DevTmp tmp;                                        // frees d_h, d_off on every path out of here
    HIPCHK(c, tmp.put(&d_h, h.data(), h.size()));
    HIPCHK(c, tmp.put(&d_off, off.data(), off.size()));
    HIPCHK(c, hipMalloc((void**)&pl->d_Hq, (size_t)Q * (NCp + 1) * sizeof(cplx)));     // owned by the plan (gf3_ctx_destroy)
    HIPCHK(c, run_rfft_nc(NCp, t, d_h, (int64_t)h.size(), DT_F64, d_off, Q, pl->d_Hq, 0));
    HIPCHK(c, hipStreamSynchronize(0));
    return GF3_OK;
}

static int build_known_time(gf3_ctx* c);

// Screening plan (gf3rx_screen.h): fp32 spectra of the chirp partitions for 8192-sample windows, in the slot order
// the kernel reads them, with max |H_q| per partition for the error bound.  The spectra are computed here on the
// host in fp64 (iterative radix-2, a few hundred kflop) and rounded once.
static void host_fft(std::vector<double>& re, std::vector<double>& im) {           // in place, length a power of two
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const long double ang = -6.283185307179586476925286766559005768L * (long double)k / (long double)len;
                const double wr = (double)cosl(ang), wi = (double)sinl(ang);
                const size_t a = i + k, b = i + k + len / 2;
                const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
    }
}
static int build_screen_plan(gf3_ctx* c) {
    constexpr int NC = GF3_SCR_NC, N = 2 * GF3_SCR_NC, T = GF3_SCR_T;
    auto& sp = c->scr;
    sp.ok = false;
    const int Q = (c->Lc + NC - 1) / NC;
    // hop = partition length: the full 4096 whenever the chirp needs more than one partition (the last one is short) --
    // every sample is then transformed exactly twice and the blocks are as few as they can be (config 3: 78 342
    // instead of 83 565 with six equal partitions of 3 840)
    int H = Q > 1 ? NC : c->Lc;
    H += H & 1;                                          // even: the kernel stores lag pairs
    if (Q > 16 || H > NC || H < 1024) return GF3_OK;      // (outside the plan's range: fp64 path only; scr_cells_kernel's block mask
                                                        //  assumes at most 64 blocks under one workgroup's 57 346 lags)
    sp.Q = Q; sp.H = H;
    std::vector<float> Hs((size_t)Q * 8 * T * 4), H0N((size_t)Q * 2), Hinf(Q);
    // band-limited kernel: the kept bins in its slot order, and per partition the error per unit |x|_2 -- rounding
    // (GF3_SCR_GAMMA max|H_q|) plus the 2-norm of what the dropped bins |k| >= 256 KS hold (gf3rx_screen.h)
    constexpr int KS = GF3_SCR_KS;
    std::vector<float> Hb((size_t)Q * (KS / 2) * T * 4), ecoef(2 * (size_t)Q);
    double hout_sum = 0.0, hall_sum = 0.0;
    for (int q = 0; q < Q; ++q) {
        std::vector<double> re(N, 0.0), im(N, 0.0);
        for (int k = 0; k < H && q * H + k < c->Lc; ++k) re[k] = c->chirp[(size_t)q * H + k];
        host_fft(re, im);
        double mx = 0.0;
        for (int k = 0; k <= NC; ++k) mx = fmax(mx, hypot(re[k], im[k]));
        Hinf[q] = (float)(mx * (1.0 + 1e-6));
        H0N[2 * q] = (float)re[0]; H0N[2 * q + 1] = (float)re[NC];
        for (int r = 0; r < 8; ++r)
            for (int t = 0; t < T; ++t) {
                const int k = (t == 0 && r == 0) ? NC / 2 : t + 256 * r;
                float* o = &Hs[(((size_t)q * 8 + r) * T + t) * 4];
                o[0] = (float)re[k]; o[1] = (float)im[k]; o[2] = (float)re[NC - k]; o[3] = (float)im[NC - k];
            }
        for (int p = 0; p < KS / 2; ++p)
            for (int t = 0; t < T; ++t) {
                const int k = t + 512 * p;
                float* o = &Hb[(((size_t)q * (KS / 2) + p) * T + t) * 4];
                o[0] = (float)re[k]; o[1] = (float)im[k]; o[2] = (float)re[k + 256]; o[3] = (float)im[k + 256];
            }
        double out2 = re[NC] * re[NC] + im[NC] * im[NC], all2 = 0.0;       // two-sided sums over the N bins of the real window
        for (int k = 256 * KS; k < NC; ++k) out2 += 2.0 * (re[k] * re[k] + im[k] * im[k]);
        for (int k = 0; k < N; ++k) all2 += re[k] * re[k] + im[k] * im[k];
        const double hout = sqrt(out2 / N) * (1.0 + 1e-9);
        ecoef[q] = (float)((double)GF3_SCR_GAMMA * ((double)Hinf[q] + hout) * (1.0 + 1e-6));     // per unit |x|_2
        ecoef[Q + q] = (float)(hout * (1.0 + 1e-6));                                             // per unit |x_out|_2
        hout_sum += hout; hall_sum += sqrt(all2 / N);
    }
    // (selective only when the chirp lives below the cut: the reference's 0-8 kHz sweep at 48 kHz drops ~1.3 %)
    sp.ring = Q <= GF3_SCR_RQ && hout_sum <= 0.05 * hall_sum;
#ifdef GF3_DEV_BUILD
    if (const char* e = getenv("GF3_SCR_R")) sp.R_forced = atoi(e);       // (tuning aid of developer builds only: output blocks per workgroup)
#endif
    std::vector<float> tw(2 * NC), twn(2 * (NC / 2 + 1));
    const long double PI2 = 6.283185307179586476925286766559005768L;
    for (int m = 0; m < NC; ++m) { const long double a = -PI2 * m / NC; tw[2 * m] = (float)cosl(a); tw[2 * m + 1] = (float)sinl(a); }
    for (int k = 0; k <= NC / 2; ++k) { const long double a = -PI2 * k / N; twn[2 * k] = (float)cosl(a); twn[2 * k + 1] = (float)sinl(a); }
    HIPCHK(c, upload((float**)&sp.d_tw, tw.data(), tw.size()));
    HIPCHK(c, upload((float**)&sp.d_twn, twn.data(), twn.size()));
    HIPCHK(c, upload((float**)&sp.d_Hs, Hs.data(), Hs.size()));
    HIPCHK(c, upload(&sp.d_H0N, H0N.data(), H0N.size()));
    HIPCHK(c, upload(&sp.d_Hinf, Hinf.data(), Hinf.size()));
    HIPCHK(c, upload((float**)&sp.d_Hb, Hb.data(), Hb.size()));
    HIPCHK(c, upload(&sp.d_ecoef, ecoef.data(), ecoef.size()));
    sp.ok = true;
    return GF3_OK;
}

// Screening plan of the frames-mode sync (gf3rx_fscreen.h): fp32 spectra of the chirp partitions for 2048-sample transforms
// in the kernel's slot order, max |H_q| per partition for the bound.  Computed on the host in fp64, rounded once.
static int build_fscreen_plan(gf3_ctx* c, int wmax) {
    constexpr int NC = GF3_FS_NC, N = 2 * GF3_FS_NC;
    auto& fp = c->fscr;
    fp.ok = false;
    if (wmax < 3 || wmax > N / 2) return GF3_OK;          // (wider windows: the all-fp64 kernel only)
    const int Lp = N - wmax + 1, Q = (c->Lc + Lp - 1) / Lp;
    if (Q > 256) return GF3_OK;
    fp.Q = Q; fp.Lp = Lp; fp.wmax = wmax;
    std::vector<float> Hs((size_t)Q * 8 * 64 * 4), H0N((size_t)Q * 2), Hinf(Q);
    for (int q = 0; q < Q; ++q) {
        std::vector<double> re(N, 0.0), im(N, 0.0);
        for (int k = 0; k < Lp && q * Lp + k < c->Lc; ++k) re[k] = c->chirp[(size_t)q * Lp + k];
        host_fft(re, im);
        double mx = 0.0;
        for (int k = 0; k <= NC; ++k) mx = fmax(mx, hypot(re[k], im[k]));
        Hinf[q] = (float)(mx * (1.0 + 1e-6));
        H0N[2 * q] = (float)re[0]; H0N[2 * q + 1] = (float)re[NC];
        for (int r = 0; r < 8; ++r)
            for (int t = 0; t < 64; ++t) {
                const int k = (t == 0 && r == 0) ? NC / 2 : t + 64 * r;
                float* o = &Hs[(((size_t)q * 8 + r) * 64 + t) * 4];
                o[0] = (float)re[k]; o[1] = (float)im[k]; o[2] = (float)re[NC - k]; o[3] = (float)im[NC - k];
            }
    }
    std::vector<float> tw(2 * NC), twn(2 * 64);
    const long double PI2 = 6.283185307179586476925286766559005768L;
    for (int m = 0; m < NC; ++m) { const long double a = -PI2 * m / NC; tw[2 * m] = (float)cosl(a); tw[2 * m + 1] = (float)sinl(a); }
    for (int k = 0; k < 64; ++k) { const long double a = -PI2 * k / N; twn[2 * k] = (float)cosl(a); twn[2 * k + 1] = (float)sinl(a); }
    HIPCHK(c, upload((float**)&fp.d_tw, tw.data(), tw.size()));
    HIPCHK(c, upload((float**)&fp.d_twn, twn.data(), twn.size()));
    HIPCHK(c, upload((float**)&fp.d_Hs, Hs.data(), Hs.size()));
    HIPCHK(c, upload(&fp.d_H0N, H0N.data(), H0N.size()));
    HIPCHK(c, upload(&fp.d_Hinf, Hinf.data(), Hinf.size()));
    fp.ok = true;
    return GF3_OK;
}

extern "C" const char* gf3_last_error(const gf3_ctx*) { return g_err; }
extern "C" int gf3_clear_runtime_error(void) { return (int)hipGetLastError(); }

extern "C" int gf3_ctx_create(const gf3_config* cfg, gf3_ctx** out) {
    if (!cfg || !out) return fail(nullptr, GF3_EINVAL, "null argument");
    *out = nullptr;
    const int N = cfg->N;
    if (N != 1024 && N != 2048 && N != 4096 && N != 8192)
        return fail(nullptr, GF3_EINVAL, "N=%d unsupported (1024, 2048, 4096, 8192)", N);
    if (cfg->CP < 0 || cfg->P < 1 || cfg->D < 1) return fail(nullptr, GF3_EINVAL, "need CP>=0, P>=1, D>=1");
    if (cfg->M < 2 || cfg->M > 64 || cfg->mu < 1 || cfg->mu > 8 || (1 << cfg->mu) < cfg->M)
        return fail(nullptr, GF3_EINVAL, "Invalid Modulation Type (M=%d, mu=%d)", cfg->M, cfg->mu);
    if (!cfg->const_re || !cfg->const_im || !cfg->const_bits || !cfg->known_re || !cfg->known_im || !cfg->data_bins)
        return fail(nullptr, GF3_EINVAL, "null table pointer");
    if (cfg->in_dtype < 0 || cfg->in_dtype > 3) return fail(nullptr, GF3_EINVAL, "bad in_dtype");
#ifdef GF3_DEV_BUILD   /* developer iteration builds instantiate N = 4096 with f32 / f64 samples only: say so instead of launching the wrong kernel */
    if (N != 4096 || cfg->in_dtype > GF3_F32)
        return fail(nullptr, GF3_EINVAL, "developer build (-DGF3_DEV_BUILD): only N=4096 with f64 / f32 samples is instantiated (asked for N=%d, in_dtype=%d)", N, cfg->in_dtype);
#endif
    gf3_ctx* c = new gf3_ctx();
    c->cfg = *cfg;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return fail(nullptr, GF3_EHIP, "hipGetDevice failed: no usable GPU"); }
    if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || c->n_cu < 1) c->n_cu = 256;
    c->NC = N / 2; c->K = N / 2 - 1; c->S = N + cfg->CP;
    c->Lc = cfg->Lc > 0 ? cfg->Lc : 5 * c->S;
    const int K = c->K;
    if (cfg->C < 1 || cfg->C > K) { delete c; return fail(nullptr, GF3_EINVAL, "C out of range"); }
    c->row_bytes = (int)(((int64_t)cfg->D * cfg->C * cfg->mu + 7) / 8);
    // polyfit range: python slice [fit_lo:fit_hi] of a length-K row (OFDM.py:462)
    c->fit_lo = cfg->fit_lo < K ? cfg->fit_lo : K;
    c->fit_hi = cfg->fit_hi < K ? cfg->fit_hi : K;
    const int L = c->fit_hi - c->fit_lo;
    if (L < 2) { delete c; return fail(nullptr, GF3_EINVAL, "phase-slope fit range [%d:%d] holds %d carriers (K=%d)", cfg->fit_lo, cfg->fit_hi, L, K); }
    c->xbar = 0.5 * (L - 1);
    { double sxx = 0; for (int i = 0; i < L; ++i) { const double d = i - c->xbar; sxx += d * d; } c->inv_sxx = 1.0 / sxx; }

    const int NC = c->NC;
    std::vector<cplx> tw(NC), twn(NC / 2 + 1), known(K), known_pts(K);
    const long double PI2 = 6.283185307179586476925286766559005768L;
    for (int m = 0; m < NC; ++m) { const long double a = -PI2 * m / NC; tw[m] = make_double2((double)cosl(a), (double)sinl(a)); }
    for (int k = 0; k <= NC / 2; ++k) { const long double a = -PI2 * k / N; twn[k] = make_double2((double)cosl(a), (double)sinl(a)); }
    for (int k = 0; k < K; ++k) {                      // 1/known = conj(known)/|known|^2
        const long double re = cfg->known_re[k], im = cfg->known_im[k], d = re * re + im * im;
        known[k] = make_double2((double)(re / d), (double)(-im / d));
        known_pts[k] = make_double2(cfg->known_re[k], cfg->known_im[k]);
    }
    std::vector<int> pos(K, -1), clab(cfg->M);
    for (int i = 0; i < cfg->C; ++i) {
        const int b = cfg->data_bins[i];
        if (b < 1 || b > K || pos[b - 1] != -1) { delete c; return fail(nullptr, GF3_EINVAL, "data_bins[%d]=%d invalid or repeated", i, b); }
        pos[b - 1] = i;
    }
    {
        bool contig = true;
        for (int i = 1; i < cfg->C; ++i) contig = contig && cfg->data_bins[i] == cfg->data_bins[0] + i;
        c->contig_lo = contig ? cfg->data_bins[0] : 0;
    }
    for (int m = 0; m < cfg->M; ++m) {
        int lab = 0;
        for (int b = 0; b < cfg->mu; ++b) lab = (lab << 1) | (cfg->const_bits[m * cfg->mu + b] & 1);
        clab[m] = lab;
    }
    // separable grid? distinct re / im levels, full grid, every label bit a function of one axis only
    {
        SepTab& sp = c->sep;
        sp.nI = sp.nQ = 0; sp.maskI = 0;
        std::vector<double> li, lq;
        auto find = [](std::vector<double>& v, double x) { for (size_t i = 0; i < v.size(); ++i) if (v[i] == x) return (int)i; v.push_back(x); return (int)v.size() - 1; };
        std::vector<int> ai(cfg->M), aq(cfg->M);
        for (int m = 0; m < cfg->M; ++m) { ai[m] = find(li, cfg->const_re[m]); aq[m] = find(lq, cfg->const_im[m]); }
        bool ok = li.size() <= 8 && lq.size() <= 8 && (int)(li.size() * lq.size()) == cfg->M;
        std::vector<int> seen(64, 0);
        for (int m = 0; ok && m < cfg->M; ++m) { int& sflag = seen[ai[m] * 8 + aq[m]]; if (sflag) ok = false; sflag = 1; }
        std::vector<int> lI(8, -1), lQ(8, -1);
        int maskI = 0, maskQ = 0;
        for (int b = 0; ok && b < cfg->mu; ++b) {
            const int bit = 1 << (cfg->mu - 1 - b);
            bool byI = true, byQ = true;
            std::vector<int> vi(8, -1), vq(8, -1);
            for (int m = 0; m < cfg->M; ++m) {
                const int v = (clab[m] & bit) ? 1 : 0;
                if (vi[ai[m]] < 0) vi[ai[m]] = v; else if (vi[ai[m]] != v) byI = false;
                if (vq[aq[m]] < 0) vq[aq[m]] = v; else if (vq[aq[m]] != v) byQ = false;
            }
            if (byI) maskI |= bit; else if (byQ) maskQ |= bit; else ok = false;
        }
        if (ok) {
            sp.nI = (int)li.size(); sp.nQ = (int)lq.size(); sp.maskI = maskI;
            for (int i = 0; i < 8; ++i) { sp.lvI[i] = sp.lvQ[i] = 0; sp.labI[i] = sp.labQ[i] = 0; }
            for (int m = 0; m < cfg->M; ++m) {
                sp.lvI[ai[m]] = cfg->const_re[m]; sp.labI[ai[m]] = clab[m] & maskI;
                sp.lvQ[aq[m]] = cfg->const_im[m]; sp.labQ[aq[m]] = clab[m] & maskQ;
            }
        }
    }
    // equally spaced levels on both axes?  (sorted ascending; spacing equal to 1e-12 relative)
    {
        const SepTab& sp = c->sep;
        UniGrid& ug = c->ug;
        ug = UniGrid{};
        auto axis = [](const double* lv, const int* lab, int n, double& lo, double& inv, unsigned long long& pack) -> bool {
            if (n < 2 || n > 8) return false;
            int order[8];
            for (int i = 0; i < n; ++i) order[i] = i;
            for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) if (lv[order[j]] < lv[order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
            const double step = (lv[order[n - 1]] - lv[order[0]]) / (n - 1);
            if (!(step > 0.0)) return false;
            pack = 0;
            for (int i = 0; i < n; ++i) {
                if (fabs(lv[order[i]] - (lv[order[0]] + i * step)) > 1e-12 * step) return false;
                if (lab[order[i]] & ~0xff) return false;
                pack |= (unsigned long long)(lab[order[i]] & 0xff) << (8 * i);
            }
            lo = lv[order[0]]; inv = 1.0 / step;
            return true;
        };
        if (sp.nI > 0 && axis(sp.lvI, sp.labI, sp.nI, ug.loI, ug.invI, ug.packI) && axis(sp.lvQ, sp.labQ, sp.nQ, ug.loQ, ug.invQ, ug.packQ)) {
            ug.nI = sp.nI; ug.nQ = sp.nQ;
        } else ug.nI = ug.nQ = 0;
    }
    // the reference's QPSK table (OFDM.py:72-77): (+,+)00 (+,-)10 (-,-)11 (-,+)01 with |re|=|im|
    if (cfg->M == 4 && cfg->mu == 2) {
        const double q = cfg->const_re[0];
        const double sr[4] = {1, 1, -1, -1}, si[4] = {1, -1, -1, 1};
        const int labs[4] = {0, 2, 3, 1};
        bool okq = q > 0.1 && q < 10.0;
        for (int m = 0; m < 4; ++m)
            okq = okq && cfg->const_re[m] == sr[m] * q && cfg->const_im[m] == si[m] * q && clab[m] == labs[m];
        c->qpsk_q = okq ? q : 0.0;
    }
    // chirp replica (sync_chirp, OFDM.py:106-109): linspace incl. endpoint, scipy linear chirp, /5
    c->chirp.resize(c->Lc);
    {
        const double t1 = (double)c->Lc / cfg->fs;
        const double step = t1 / (double)(c->Lc - 1);
        const double beta = (cfg->f1 - cfg->f0) / t1;
        for (int i = 0; i < c->Lc; ++i) {
            const double t = (i == c->Lc - 1) ? t1 : (double)i * step;
            const double ph = 2 * M_PI * (cfg->f0 * t + 0.5 * beta * t * t);
            c->chirp[i] = cos(ph) / 5;
        }
    }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { int rc_ = fail(nullptr, GF3_EHIP, "%s: %s", #x, hipGetErrorString(e_)); gf3_ctx_destroy(c); return rc_; } } while (0)
    CK(upload(&c->d_tw, tw.data(), tw.size()));
    CK(upload(&c->d_twn, twn.data(), twn.size()));
    CK(upload(&c->d_known, known.data(), known.size()));
    CK(upload(&c->d_pos, pos.data(), pos.size()));
    CK(upload(&c->d_clab, clab.data(), clab.size()));
    {
        std::vector<int> inv(1 << cfg->mu, 0);
        for (int m = cfg->M - 1; m >= 0; --m) inv[clab[m]] = m;
        CK(upload(&c->d_idx_of_label, inv.data(), inv.size()));
        CK(upload(&c->d_chirp, c->chirp.data(), c->chirp.size()));
        const int nst = (c->Lc + SCR_REF_WT - 1) / SCR_REF_WT;
        std::vector<double> tiled((size_t)nst * SCR_REF_WT, 0.0);
        for (int st = 0; st < nst; ++st)
            for (int q = 0; q < 8; ++q)
                for (int lane = 0; lane < 64; ++lane)
                    for (int h = 0; h < 2; ++h) {
                        const int k = SCR_REF_WT * st + 16 * lane + 2 * q + h;
                        if (k < c->Lc) tiled[(((size_t)st * 8 + q) * 64 + lane) * 2 + h] = c->chirp[k];
                    }
        CK(upload(&c->d_chirp_t, tiled.data(), tiled.size()));
        c->known_pts = known_pts;
    }
    CK(upload(&c->d_cre, cfg->const_re, (size_t)cfg->M));
    CK(upload(&c->d_cim, cfg->const_im, (size_t)cfg->M));
#undef CK
    // the tables are now device-resident; do not keep the caller's host pointers
    c->cfg.const_re = c->cfg.const_im = c->cfg.known_re = c->cfg.known_im = nullptr;
    c->cfg.const_bits = nullptr; c->cfg.data_bins = nullptr;
    int wmax = cfg->max_window > 0 ? cfg->max_window : 512;
    if (wmax > N / 2) wmax = N / 2;
    // frames-mode plan: (Q+1) transforms of size Nf per packet; pick Nf in {N, N/2} by cost ~ (Q+1) Nf log2 Nf
    int NCf = NC;
    if (NC >= 1024 && wmax <= NC / 2) {
        auto cost = [&](int nc) { const int nf = 2 * nc, lp = nf - wmax + 1; const int q = (c->Lc + lp - 1) / lp;
                                  return (double)(q + 1) * nf * log2((double)nf); };
        if (cost(NC / 2) < cost(NC)) NCf = NC / 2;
    }
    auto tables_for = [&](int NCp, FftTables& t) -> bool {         // twiddle set for a plan's FFT size
        if (NCp == NC) { t = FftTables{c->d_tw, c->d_twn}; return true; }
        for (int i = 0; i < 2; ++i) if (c->nc_x[i] == NCp) { t = FftTables{c->d_tw_x[i], c->d_twn_x[i]}; return true; }
        const int i = c->nc_x[0] ? 1 : 0;
        std::vector<cplx> twh(NCp), twnh(NCp / 2 + 1);
        for (int m = 0; m < NCp; ++m) { const long double a2 = -PI2 * m / NCp; twh[m] = make_double2((double)cosl(a2), (double)sinl(a2)); }
        for (int k = 0; k <= NCp / 2; ++k) { const long double a2 = -PI2 * k / (2 * NCp); twnh[k] = make_double2((double)cosl(a2), (double)sinl(a2)); }
        if (upload(&c->d_tw_x[i], twh.data(), twh.size()) != hipSuccess) return false;
        if (upload(&c->d_twn_x[i], twnh.data(), twnh.size()) != hipSuccess) return false;
        c->nc_x[i] = NCp;
        t = FftTables{c->d_tw_x[i], c->d_twn_x[i]};
        return true;
    };
    // stream-mode plan (spectral delay line, hop = partition length): FFT size 2N where the kernels exist --
    // half as many partitions, half the spectrum bytes per lag
    int NCs = NC;
#ifndef GF3_DEV_BUILD
    if (2 * NC <= 4096) NCs = 2 * NC;
#endif
    FftTables tf, ts;
    if (!tables_for(NCf, tf) || !tables_for(NCs, ts)) { gf3_ctx_destroy(c); return fail(nullptr, GF3_EHIP, "table upload failed"); }
    int rc = build_plan(c, &c->frames_plan, NCf, tf, 2 * NCf - wmax + 1);
    if (rc == GF3_OK) rc = build_plan(c, &c->stream_plan, NCs, ts, NCs);
    if (rc == GF3_OK) rc = build_known_time(c);
    if (rc == GF3_OK) rc = build_screen_plan(c);
    if (rc == GF3_OK) rc = build_fscreen_plan(c, wmax);
    if (rc != GF3_OK) { gf3_ctx_destroy(c); return rc; }
    *out = c;
    return GF3_OK;
}

extern "C" void gf3_ctx_destroy(gf3_ctx* c) {
    if (!c) return;
    DeviceGuard dg(c);
    void* ptrs[] = {c->d_tw_x[0], c->d_twn_x[0], c->d_tw_x[1], c->d_twn_x[1], c->d_tw, c->d_twn, c->d_known, c->d_pos, c->d_clab, c->d_cre, c->d_cim,
                    c->frames_plan.d_Hq, c->stream_plan.d_Hq, c->d_idx_of_label, c->d_chirp, c->d_chirp_t, c->d_known_time,
                    c->scr.d_tw, c->scr.d_twn, c->scr.d_Hs, c->scr.d_H0N, c->scr.d_Hinf, c->scr.d_Hb, c->scr.d_ecoef,
                    c->fscr.d_tw, c->fscr.d_twn, c->fscr.d_Hs, c->fscr.d_H0N, c->fscr.d_Hinf};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete c;
}

// diagnostic builds only: device buffer [F][8] of uint64 that receives per-phase s_memtime stamps
extern "C" void gf3_debug_set_stamps(gf3_ctx* c, void* d_buf) { if (c) c->stamps = (unsigned long long*)d_buf; }

extern "C" int32_t gf3_bytes_per_frame(const gf3_ctx* c) { return c ? c->row_bytes : 0; }
extern "C" int32_t gf3_sync_max_window(const gf3_ctx* c) { return c ? c->frames_plan.W : 0; }

extern "C" int gf3_chirp_replica(const gf3_ctx* c, double* h_out) {
    if (!c || !h_out) return fail(c, GF3_EINVAL, "null argument");
    memcpy(h_out, c->chirp.data(), c->chirp.size() * sizeof(double));
    return GF3_OK;
}

extern "C" int gf3_rfft_batch(gf3_ctx* c, const void* d_in, int64_t n_in, const int64_t* d_offsets, int64_t n_sym,
                              void* d_out, void* stream) {
    DeviceGuard dg(c);
    if (c && n_sym == 0) return GF3_OK;
    if (!c || !d_in || !d_offsets || !d_out || n_sym < 0) return fail(c, GF3_EINVAL, "gf3_rfft_batch: bad argument");
    HIPCHK(c, run_rfft(c, d_in, n_in, c->cfg.in_dtype, d_offsets, n_sym, (cplx*)d_out, (hipStream_t)stream));
    return GF3_OK;
}

extern "C" int gf3_demod_frames_ex(gf3_ctx* c, const void* d_in, int64_t n_in, const int64_t* d_off, int64_t F,
                                   uint8_t* d_bits, void* d_eq, void* d_Hs, void* d_He, double* d_slope, void* d_Hest,
                                   int32_t* d_status, void* d_work, int32_t mode, void* stream) {
    DeviceGuard dg(c);
    if (c && F == 0) return GF3_OK;
    if (!c || !d_in || !d_off || !d_bits || F < 0 || mode < 0 || mode > 2) return fail(c, GF3_EINVAL, "gf3_demod_frames: bad argument");
    const gf3_config& g = c->cfg;
    DemodArgs a{{c->d_tw, c->d_twn}, d_in, n_in, d_off, g.in_dtype,
                g.CP, c->S, g.P, g.D, c->K, g.C, g.mu, g.M,
                c->d_known, c->d_pos, c->contig_lo, demod_ring(c), c->d_cre, c->d_cim, c->d_clab,
                c->fit_lo, c->fit_hi, c->xbar, c->inv_sxx,
                d_bits, c->row_bytes, (cplx*)d_eq, (cplx*)d_Hs, (cplx*)d_He, d_slope, (cplx*)d_Hest, d_status,
                nullptr, nullptr, nullptr, nullptr, c->qpsk_q, c->ug, c->stamps, 0, 0, nullptr};
    // long packets, few at a time: pilot sums, estimate and data symbols as three launches (gf3rx_demod_split.hip)
    if (d_work && demod_wants_split(c, F, mode)) return demod_split(c, a, F, d_work, (hipStream_t)stream);
    hipError_t e = hipSuccess;
    if (d_eq || d_Hest) e = launch_demod_full(c, a, F, (hipStream_t)stream);
    else if (c->qpsk_q > 0.0) e = launch_demod_qpsk(c, a, F, (hipStream_t)stream);
    else e = launch_demod_scan(c, a, F, (hipStream_t)stream);
    HIPCHK(c, e);
    return GF3_OK;
}
extern "C" int gf3_demod_frames(gf3_ctx* c, const void* d_in, int64_t n_in, const int64_t* d_off, int64_t F,
                                uint8_t* d_bits, void* d_eq, void* d_Hs, void* d_He, double* d_slope, void* d_Hest,
                                int32_t* d_status, void* stream) {
    return gf3_demod_frames_ex(c, d_in, n_in, d_off, F, d_bits, d_eq, d_Hs, d_He, d_slope, d_Hest, d_status, nullptr, 1, stream);
}

extern "C" int gf3_equalise(gf3_ctx* c, const void* d_data, const void* d_start, const void* d_end, int64_t F,
                            void* d_eq_all, void* d_Hs, void* d_He, double* d_slope, void* d_Hest,
                            uint8_t* d_bits, void* stream) {
    DeviceGuard dg(c);
    if (c && F == 0) return GF3_OK;
    if (!c || !d_data || !d_start || !d_end || !d_bits || F < 0) return fail(c, GF3_EINVAL, "gf3_equalise: bad argument");
    const gf3_config& g = c->cfg;
    DemodArgs a{{c->d_tw, c->d_twn}, nullptr, 0, nullptr, g.in_dtype,
                g.CP, c->S, g.P, g.D, c->K, g.C, g.mu, g.M,
                c->d_known, c->d_pos, c->contig_lo, demod_ring(c), c->d_cre, c->d_cim, c->d_clab,
                c->fit_lo, c->fit_hi, c->xbar, c->inv_sxx,
                d_bits, c->row_bytes, nullptr, (cplx*)d_Hs, (cplx*)d_He, d_slope, (cplx*)d_Hest, nullptr,
                (const cplx*)d_data, (const cplx*)d_start, (const cplx*)d_end, (cplx*)d_eq_all, c->qpsk_q, c->ug, nullptr, 0, 0, nullptr};
    const hipError_t e = launch_demod_spectra(c, a, F, (hipStream_t)stream);
    HIPCHK(c, e);
    return GF3_OK;
}


// Known pilot symbol in the time domain (with prefix, before the x2 gain), built once at context
// creation with the transmit kernel itself: a one-symbol packet whose "filler" is the known-symbol
// vector and which has no data carriers.
static int build_known_time(gf3_ctx* c) {
    const gf3_config& g = c->cfg;
    TxArgs a{};
    a.t = {c->d_tw, c->d_twn};
    a.CP = g.CP; a.S = c->S; a.K = c->K; a.mu = g.mu; a.M = g.M; a.Lc = c->Lc;
    a.cre = c->d_cre; a.cim = c->d_cim; a.idx_of_label = c->d_idx_of_label; a.chirp = c->d_chirp;
    cplx* d_kn = nullptr; double* d_row = nullptr; int* d_nopos = nullptr; uint8_t* d_nobits = nullptr;
    std::vector<int> nopos(c->K, -1);
    const int64_t rowlen = c->Lc + c->S;
    DevTmp tmp;                                        // frees the four scratch buffers on every path out of here
    HIPCHK(c, tmp.put(&d_kn, c->known_pts.data(), c->known_pts.size()));
    HIPCHK(c, tmp.put(&d_nopos, nopos.data(), nopos.size()));
    HIPCHK(c, tmp.alloc(&d_row, (size_t)rowlen));
    HIPCHK(c, tmp.alloc(&d_nobits, (size_t)16));
    HIPCHK(c, hipMalloc((void**)&c->d_known_time, c->S * sizeof(double)));                // owned by the context
    HIPCHK(c, hipMemset(c->d_known_time, 0, c->S * sizeof(double)));
    a.P = 0; a.D = 1; a.C = 0; a.pos = d_nopos; a.contig_lo = 0; a.filler = d_kn; a.known_time = c->d_known_time;
    a.bits = d_nobits; a.row_bytes = 0; a.gaps = nullptr; a.out = d_row; a.stride = rowlen; a.out_dt = DT_F64;
    int rc = tx_launch(c, a, 1, 0);
    if (rc != GF3_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(0));
    std::vector<double> h(c->S);
    HIPCHK(c, hipMemcpy(h.data(), d_row + c->Lc, c->S * sizeof(double), hipMemcpyDeviceToHost));
    for (auto& x : h) x *= 0.5;                          // stored before the x2 gain
    HIPCHK(c, hipMemcpy(c->d_known_time, h.data(), c->S * sizeof(double), hipMemcpyHostToDevice));
    return GF3_OK;
}

extern "C" int gf3_tx_frames(gf3_ctx* c, const uint8_t* d_bits_packed, const void* d_filler_c128, const int64_t* d_gaps,
                             int64_t F, void* d_out, int64_t stride, int32_t out_dtype, void* stream) {
    DeviceGuard dg(c);
    if (c && F == 0) return GF3_OK;
    if (!c || !d_bits_packed || !d_filler_c128 || !d_out || F < 0 || (out_dtype != GF3_F32 && out_dtype != GF3_F64))
        return fail(c, GF3_EINVAL, "gf3_tx_frames: bad argument");
    const gf3_config& g = c->cfg;
    if (stride < (int64_t)c->Lc + (int64_t)(2 * g.P + g.D) * c->S) return fail(c, GF3_EINVAL, "gf3_tx_frames: stride shorter than a packet");
    hipStream_t st = (hipStream_t)stream;
    TxArgs a{};
    a.t = {c->d_tw, c->d_twn};
    a.CP = g.CP; a.S = c->S; a.K = c->K; a.mu = g.mu; a.M = g.M; a.Lc = c->Lc;
    a.pos = c->d_pos; a.cre = c->d_cre; a.cim = c->d_cim; a.idx_of_label = c->d_idx_of_label; a.chirp = c->d_chirp;
    a.P = g.P; a.D = g.D; a.C = g.C; a.contig_lo = c->contig_lo; a.filler = (const cplx*)d_filler_c128;
    a.known_time = c->d_known_time; a.bits = d_bits_packed; a.row_bytes = c->row_bytes; a.gaps = d_gaps;
    a.out = d_out; a.stride = stride; a.out_dt = out_dtype == GF3_F32 ? DT_F32 : DT_F64;
    return tx_launch(c, a, F, st);
}

// ============================================================================
// Schmidl & Cox timing metric (receiver.schmidlcox_method, OFDM.py:376-387)
//   P[0] = 0, P[d+1] = P[d] + r[d+L] r[d+2L] - r[d] r[d+L]; answer = first argmax |P| + N - 1
// One workgroup walks the search range in chunks: per-thread terms -> wave shuffle scan -> carry;
// the running arg-max keeps (|P|, smallest index) and is reduced across the block at the end.
// ============================================================================
struct ScArgs { const void* in; int dt; int64_t S; int L; int N; int64_t* out; };

__global__ __launch_bounds__(1024) void schmidl_cox_kernel(ScArgs a) {
    __shared__ double wsum[16];
    __shared__ double bval[16];
    __shared__ long long bidx[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int ITEMS = 4;
    double best = 0.0;                 // |P[0]| = 0 at index 0
    long long besti = 0;
    double carry = 0.0;
    for (int64_t base = 0; base < a.S - 1; base += 1024 * ITEMS) {
        const int64_t d0 = base + (int64_t)tid * ITEMS;
        double t[ITEMS], run = 0.0;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int64_t d = d0 + k;
            double x = 0.0;
            if (d < a.S - 1) {
                const double r0 = load_sample(a.in, d, a.dt), r1 = load_sample(a.in, d + a.L, a.dt),
                             r2 = load_sample(a.in, d + 2 * a.L, a.dt);
                x = r1 * r2 - r0 * r1;
            }
            run += x;
            t[k] = run;                // inclusive prefix inside the thread
        }
        double incl = run;             // block-wide inclusive scan of the per-thread totals
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const double y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        double woff = 0.0, tot = 0.0;
        for (int w = 0; w < 16; ++w) { if (w < wave) woff += wsum[w]; tot += wsum[w]; }
        const double before = carry + woff + (incl - run);
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int64_t d = d0 + k;
            if (d < a.S - 1) {
                const double v = fabs(before + t[k]);      // |P[d+1]|
                if (v > best) { best = v; besti = d + 1; }
            }
        }
        carry += tot;
        __syncthreads();
    }
    // arg-max with first-index tie rule: wave reduction, then across waves
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double ov = __shfl_xor(best, o, 64);
        const long long oi = __shfl_xor(besti, o, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if (lane == 0) { bval[wave] = best; bidx[wave] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (bval[w] > best || (bval[w] == best && bidx[w] < besti)) { best = bval[w]; besti = bidx[w]; }
        a.out[0] = besti + a.N - 1;
    }
}

extern "C" int gf3_schmidl_cox(gf3_ctx* c, const void* d_r, int64_t n, int64_t search_len, int64_t* d_index, void* stream) {
    DeviceGuard dg(c);
    if (!c || !d_r || !d_index || search_len < 2) return fail(c, GF3_EINVAL, "gf3_schmidl_cox: bad argument");
    const int L = c->K + 1;
    if (n < search_len - 1 + 2 * (int64_t)L) return fail(c, GF3_EINVAL, "gf3_schmidl_cox: stream shorter than search length + 2L");
    ScArgs a{d_r, c->cfg.in_dtype, search_len, L, 2 * c->NC, d_index};
    hipLaunchKernelGGL(schmidl_cox_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    HIPCHK(c, hipGetLastError());
    return GF3_OK;
}


extern "C" int64_t gf3_sync_frames_workspace_bytes(const gf3_ctx* c, int64_t F) {
    if (!c || F < 0) return 0;
    return (int64_t)((size_t)F * sizeof(int) + 64);           // [count | pad | unresolved window numbers]
}

// mode 0: all fp64 (corr_kernel on every window).  mode 1: fp32 screen with a proven bound per window (gf3rx_fscreen.h);
// the windows it cannot decide are listed and corr_kernel runs on those.  The starts are the same either way.
static int sync_frames_impl(gf3_ctx* c, const void* d_in, int64_t n_in, int64_t F, int64_t stride, int32_t win_lo, int32_t win_hi,
                            int64_t* d_starts, double* d_peak, int32_t mode, void* d_work, float* dbg_y32, float* dbg_err, int* dbg_cls,
                            bool screen_only, void* stream) {
    DeviceGuard dg(c);
    if (c && F == 0) return GF3_OK;
    if (!c || !d_in || !d_starts || F < 0 || mode < 0 || mode > 1) return fail(c, GF3_EINVAL, "gf3_sync_frames: bad argument");
    const int W = win_hi - win_lo;
    const CorrPlan& pl = c->frames_plan;
    if (W < 3 || W > pl.W) return fail(c, GF3_EINVAL, "gf3_sync_frames: window %d outside [3, %d]", W, pl.W);
    CorrArgs a{};
    a.t = pl.t; a.in = d_in; a.n_in = n_in; a.dt = c->cfg.in_dtype;
    a.Hq = pl.d_Hq; a.Q = pl.Q; a.Lp = pl.Lp; a.Lc = c->Lc; a.Wmax = W;
    a.stride = stride; a.win_lo = win_lo; a.W = W; a.starts = d_starts; a.peak = d_peak; a.thresh = c->cfg.thresh;
    const auto& fp = c->fscr;
    // the screen serves index-only calls within its plan's window; a caller that wants the fp64 peak VALUE gets the fp64 kernel
    const bool screened = mode == 1 && d_work && fp.ok && W <= fp.wmax && !d_peak && F <= 0x7fffffff;
    if (screen_only && !screened) return fail(c, GF3_EINVAL, "gf3_debug_frames_screen: no screening plan for this window (max_window %d)", fp.wmax);
    if (!screened) {
        HIPCHK(c, run_corr(c, pl, a, F, (hipStream_t)stream));
        return GF3_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    int* count = (int*)d_work;
    int* list = (int*)((char*)d_work + 64);
    HIPCHK(c, hipMemsetAsync(count, 0, 64, st));
    FScreenArgs fa{d_in, n_in, c->cfg.in_dtype, fp.d_tw, fp.d_twn, fp.d_Hs, fp.d_H0N, fp.d_Hinf, fp.Q, fp.Lp, c->Lc, W,
                   stride, win_lo, W, (float)c->cfg.thresh, d_starts, list, count, dbg_y32, dbg_err, dbg_cls};
    HIPCHK(c, launch_fscreen(c, fa, F, st));
    if (screen_only) return GF3_OK;
    a.list = list; a.count = count;
    HIPCHK(c, run_corr(c, pl, a, F, st, true));               // (grid = the list's capacity; workgroups past its length return at once)
    return GF3_OK;
}
extern "C" int gf3_sync_frames(gf3_ctx* c, const void* d_in, int64_t n_in, int64_t F, int64_t stride,
                               int32_t win_lo, int32_t win_hi, int64_t* d_starts, double* d_peak, void* stream) {
    return sync_frames_impl(c, d_in, n_in, F, stride, win_lo, win_hi, d_starts, d_peak, 0, nullptr, nullptr, nullptr, nullptr, false, stream);
}
extern "C" int gf3_sync_frames_ex(gf3_ctx* c, const void* d_in, int64_t n_in, int64_t F, int64_t stride,
                                  int32_t win_lo, int32_t win_hi, int64_t* d_starts, double* d_peak, int32_t mode, void* d_work, void* stream) {
    return sync_frames_impl(c, d_in, n_in, F, stride, win_lo, win_hi, d_starts, d_peak, mode, d_work, nullptr, nullptr, nullptr, false, stream);
}
// tests: the screening pass alone -- fp32 lags [F][W], the bound per window, the verdict per window (0 resolved with a
// detection, 1 resolved without, 2 unresolved: d_starts is then left alone), the unresolved windows in d_work
extern "C" int gf3_debug_frames_screen(gf3_ctx* c, const void* d_in, int64_t n_in, int64_t F, int64_t stride, int32_t win_lo, int32_t win_hi,
                                       int64_t* d_starts, float* d_y32, float* d_err, int32_t* d_cls, void* d_work, void* stream) {
    return sync_frames_impl(c, d_in, n_in, F, stride, win_lo, win_hi, d_starts, nullptr, 1, d_work, d_y32, d_err, d_cls, true, stream);
}

static int run_demap(gf3_ctx* c, const void* d_sym, int64_t n, uint8_t* bits, uint8_t* idx, float* llr, double nv, void* stream);
__global__ void zf_bins_kernel(const int* pos, int K, int* bins) {      // bins[pos[k]] = k + 1 for every data carrier
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K && pos[k] >= 0) bins[pos[k]] = k + 1;
}
// ============================================================================
// known-channel zero forcing (the reference's older flow, `Weekend Challenge.ipynb` cells 9-15: H = fft(h, N),
// symbols = FFT(rx) / H on bins 1..N/2-1).  Not on receive()'s path and without a surviving reference function:
// parity is pinned by the formula only (oracle.zf_known_h).
// ============================================================================
struct ZfArgs { const cplx* X; const cplx* H; const int* bins; int64_t n_sym; int C, NC; cplx* eq; };
__global__ void zf_kernel(ZfArgs a) {
    const int64_t total = a.n_sym * a.C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / a.C;
        const int b = a.bins[i - s * a.C];
        a.eq[i] = cdiv_np(a.X[s * (a.NC + 1) + b], a.H[b]);              // complex128 division as NumPy performs it
    }
}
extern "C" int64_t gf3_known_h_workspace_bytes(const gf3_ctx* c, int64_t n_sym) {
    if (!c || n_sym < 0) return 0;
    return (int64_t)((size_t)(n_sym + 1) * (c->NC + 1) * sizeof(cplx) + (size_t)2 * c->NC * sizeof(double) + 16 + (size_t)c->cfg.C * 4 + 256);
}
extern "C" int gf3_equalise_known_h(gf3_ctx* c, const void* d_in, int64_t n_in, const int64_t* d_offsets, int64_t n_sym,
                                    const double* d_h, int32_t n_taps, void* d_eq, uint8_t* d_bits, uint8_t* d_idx,
                                    void* d_work, void* stream) {
    DeviceGuard dg(c);
    if (c && n_sym == 0) return GF3_OK;
    if (!c || !d_in || !d_offsets || !d_h || !d_eq || !d_bits || !d_work || n_sym < 0 || n_taps < 1 || n_taps > 2 * c->NC)
        return fail(c, GF3_EINVAL, "gf3_equalise_known_h: bad argument (1 <= n_taps <= N)");
    hipStream_t st = (hipStream_t)stream;
    const int NC = c->NC, N = 2 * NC;
    char* base = (char*)d_work;
    cplx* X = (cplx*)base;                                               // [n_sym][NC+1]
    cplx* H = X + (size_t)n_sym * (NC + 1);                              // [NC+1]
    double* hpad = (double*)(H + (NC + 1));                              // [N] taps, zero padded (np.fft.fft(h, N))
    int64_t* zero = (int64_t*)(hpad + N);                                // offset 0 of the padded taps
    int* bins = (int*)(zero + 2);
    HIPCHK(c, hipMemsetAsync(hpad, 0, (size_t)N * sizeof(double) + 16, st));
    HIPCHK(c, hipMemcpyAsync(hpad, d_h, (size_t)n_taps * sizeof(double), hipMemcpyDeviceToDevice, st));
    // data-carrier bins in output order (the context keeps the carrier -> position map; invert it on the device)
    hipLaunchKernelGGL(zf_bins_kernel, dim3((c->K + 255) / 256), dim3(256), 0, st, (const int*)c->d_pos, c->K, bins);
    HIPCHK(c, run_rfft_nc(NC, FftTables{c->d_tw, c->d_twn}, hpad, N, DT_F64, zero, 1, H, st));
    HIPCHK(c, run_rfft(c, d_in, n_in, c->cfg.in_dtype, d_offsets, n_sym, X, st));
    ZfArgs a{X, H, bins, n_sym, c->cfg.C, NC, (cplx*)d_eq};
    int64_t grid = (n_sym * c->cfg.C + 255) / 256;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(zf_kernel, dim3((unsigned)grid), dim3(256), 0, st, a);
    HIPCHK(c, hipGetLastError());
    return run_demap(c, d_eq, n_sym * c->cfg.C, d_bits, d_idx, nullptr, 1.0, stream);
}

static int run_demap(gf3_ctx* c, const void* d_sym, int64_t n, uint8_t* bits, uint8_t* idx, float* llr, double nv, void* stream) {
    DemapArgs a{(const cplx*)d_sym, n, c->cfg.M, c->cfg.mu, c->d_cre, c->d_cim, c->d_clab, bits, llr, nv > 0 ? 1.0 / nv : 0.0, idx, c->sep};
    int64_t grid = (n + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    if (grid < 1) return GF3_OK;
    int hI = 0, hQ = 0;
    if (bits) hipLaunchKernelGGL(demap_hard_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    else if (c->sep.nI > 0 && sep_is_binary(c->sep, c->cfg.mu, hI, hQ) && hI <= 3) {
        switch (hI) {                                  // QPSK, 16-QAM, 64-QAM: straight-line minima
            case 1: hipLaunchKernelGGL((soft_demap_bin_kernel<1, 1>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a); break;
            case 2: hipLaunchKernelGGL((soft_demap_bin_kernel<2, 2>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a); break;
            default: hipLaunchKernelGGL((soft_demap_bin_kernel<3, 3>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a); break;
        }
    } else if (c->sep.nI > 0) {
        switch (c->cfg.mu) {
            case 1: hipLaunchKernelGGL(soft_demap_sep_kernel<1>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a); break;
            case 2: hipLaunchKernelGGL(soft_demap_sep_kernel<2>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a); break;
            case 3: hipLaunchKernelGGL(soft_demap_sep_kernel<3>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a); break;
            case 4: hipLaunchKernelGGL(soft_demap_sep_kernel<4>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a); break;
            case 5: hipLaunchKernelGGL(soft_demap_sep_kernel<5>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a); break;
            case 6: hipLaunchKernelGGL(soft_demap_sep_kernel<6>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a); break;
            case 7: hipLaunchKernelGGL(soft_demap_sep_kernel<7>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a); break;
            default: hipLaunchKernelGGL(soft_demap_sep_kernel<8>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a); break;
        }
    } else hipLaunchKernelGGL(soft_demap_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    HIPCHK(c, hipGetLastError());
    return GF3_OK;
}
// ============================================================================
// PS + decode (OFDM.py:504-505, 541-544): packed decisions -> the int64 0/1 array the reference returns, whitening
// mask applied.  A thread owns two consecutive output elements (one 16-byte store; a wave writes 1 KB contiguously),
// which is what lets the destination be pinned HOST memory written over PCIe by the kernel itself.
// ============================================================================
struct UnpackArgs { const uint8_t* packed; int row_bytes; int64_t bpf, total; const uint8_t* mask; int n_mask; long long* out; };
__global__ __launch_bounds__(256) void unpack_bits_kernel(UnpackArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; 2 * t < a.total; t += stride) {
        long long v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t e = 2 * t + h;                               // output index = global bit number
            const int64_t f = e / a.bpf, j = e - f * a.bpf;            // packet, bit inside the packet
            unsigned bit = e < a.total ? (a.packed[f * a.row_bytes + (j >> 3)] >> (7 - (int)(j & 7))) & 1u : 0u;
            if (a.mask) bit ^= a.mask[e % a.n_mask] & 1u;              // tile(mask)[:len] runs over the whole stream
            v[h] = (long long)bit;
        }
        if (2 * t + 1 < a.total) *(longlong2*)(a.out + 2 * t) = make_longlong2(v[0], v[1]);
        else a.out[2 * t] = v[0];
    }
}
extern "C" int gf3_unpack_bits(gf3_ctx* c, const uint8_t* d_bits, int64_t F, const uint8_t* d_mask, int32_t n_mask, void* out, void* stream) {
    DeviceGuard dg(c);
    if (c && F == 0) return GF3_OK;
    if (!c || !d_bits || !out || F < 0 || (d_mask && n_mask < 1) || ((uintptr_t)out & 15)) return fail(c, GF3_EINVAL, "gf3_unpack_bits: bad argument (out must be 16-byte aligned)");
    // where does `out` live?  Device memory is used as it is; pinned host memory through its device-side address; anything
    // else (pageable memory) cannot be written by a kernel
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, out) != hipSuccess) { (void)hipGetLastError(); return fail(c, GF3_EINVAL, "gf3_unpack_bits: out is neither device memory nor pinned host memory"); }
    void* dst = out;
    if (at.type == hipMemoryTypeHost) {
        dst = at.devicePointer;
        if (!dst) return fail(c, GF3_EINVAL, "gf3_unpack_bits: the pinned host buffer is not mapped into the device's address space");
    } else if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged && at.type != hipMemoryTypeUnified)
        return fail(c, GF3_EINVAL, "gf3_unpack_bits: out is neither device memory nor pinned host memory");
    const int64_t bpf = (int64_t)c->cfg.D * c->cfg.C * c->cfg.mu;
    UnpackArgs a{d_bits, c->row_bytes, bpf, F * bpf, d_mask, n_mask, (long long*)dst};
    int64_t grid = (a.total / 2 + 255) / 256;
    if (grid > 8 * (int64_t)c->n_cu) grid = 8 * (int64_t)c->n_cu;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(unpack_bits_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    HIPCHK(c, hipGetLastError());
    return GF3_OK;
}

extern "C" int gf3_demap_hard(gf3_ctx* c, const void* d_sym, int64_t n, uint8_t* d_bits, uint8_t* d_idx, void* stream) {
    DeviceGuard dg(c);
    if (c && n == 0) return GF3_OK;
    if (!c || !d_sym || !d_bits || n < 0) return fail(c, GF3_EINVAL, "gf3_demap_hard: bad argument");
    return run_demap(c, d_sym, n, d_bits, d_idx, nullptr, 1.0, stream);
}
extern "C" int gf3_soft_demap(gf3_ctx* c, const void* d_sym, int64_t n, double noise_var, float* d_llr, void* stream) {
    DeviceGuard dg(c);
    if (c && n == 0) return GF3_OK;
    if (!c || !d_sym || !d_llr || n < 0 || !(noise_var > 0)) return fail(c, GF3_EINVAL, "gf3_soft_demap: bad argument");
    return run_demap(c, d_sym, n, nullptr, nullptr, d_llr, noise_var, stream);
}
