// Host-side plumbing shared by the translation units of libgf3rx: kernel argument blocks, the context, error
// reporting, launch helpers and the launchers each kernel family exports to the ABI layer (gf3rx_abi.hip, gf3rx_sync.hip).
// Kernels live in per-family translation units compiled in parallel (gf3_audio_modem_amd/build.py):
//   gf3rx_fft.hip            rfft_kernel, tx_kernel
//   gf3rx_demod_{qpsk,scan,full}.hip   the three modes of demod_kernel (gf3rx_demod.h)
//   gf3rx_demod_split.hip, gf3rx_dsplit_{qpsk,scan,full}.hip   the two-phase demodulation of long packets
//   gf3rx_corr.hip           corr_kernel, spec_kernel, ols_kernel
//   gf3rx_screen.hip         scr_ring_kernel, scr_ols_kernel, scr_refine_kernel (gf3rx_screen.h)
//   gf3rx_fscreen.hip        corr_screen_kernel: the opt-in fp32 screen of the frames-mode sync (gf3rx_fscreen.h)
//   gf3rx_stamp.cpp          the build stamp (source hash), recompiled on every change
//   gf3rx_sync.hip           pk_*, ck_*, scr list kernels + gf3_sync_stream*, gf3_sync_chunk, gf3_sync_decide
//   gf3rx_abi.hip            context, plans, the remaining entry points, demappers, zero forcing, Schmidl-Cox
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <vector>

#include "gf3rx.h"
#include "gf3rx_device.h"
#include "gf3rx_screen_defs.h"

// ============================================================================
// kernel argument blocks
// ============================================================================
// Grid constellation whose bit labels split per axis (square Gray QAM, the reference QPSK):
// point = lvI[a] + j lvQ[b], label = labI[a] | labQ[b].  Distances then separate per axis, so
// hard decisions and max-log LLRs cost O(levels) instead of O(points).  n = 0: not separable.
struct SepTab {
    int nI, nQ;
    double lvI[8], lvQ[8];
    int labI[8], labQ[8];       // label bits contributed by each level (already in label position)
    int maskI;                  // label bits owned by the I axis
};

// The same kind of table when, in addition, the levels of each axis are equally spaced (square Gray QAM, the reference
// QPSK): the nearest level is then one rint() away.  Levels are numbered in ascending order here; pack holds the
// label bits the i-th level contributes, one byte each (mu <= 8).  nI = 0: not such a table.
struct UniGrid {
    int nI, nQ;
    double loI, invI, loQ, invQ;             // lowest level and 1 / spacing per axis
    unsigned long long packI, packQ;
};

struct FftTables {
    const cplx* tw;    // [NC]      exp(-2 pi i m / NC)
    const cplx* twn;   // [NC/2+1]  exp(-2 pi i k / N)
};

struct RfftArgs {
    FftTables t;
    const void* in; int64_t n_in; const int64_t* off; int dt;
    cplx* out;
};

struct DemodArgs {
    FftTables t;
    const void* in; int64_t n_in; const int64_t* off; int dt;
    int CP, S, P, D, K, C, mu, M;
    const cplx* inv_known;    // [K] 1/known symbol
    const int* pos;           // [K] data-carrier position or -1
    int contig_lo;            // >0: data bins are contig_lo .. contig_lo+C-1 in order (no table look-up)
    int ring;                 // symbols held by the decision-byte ring in LDS (power of two, see demod_ring)
    const double* cre; const double* cim; const int* clab;   // [M]
    int fit_lo, fit_hi;       // effective python-slice bounds, fit_hi <= K
    double xbar, inv_sxx;
    uint8_t* bits; int row_bytes;
    cplx* eq; cplx* Hs; cplx* He; double* slope; cplx* Hest; int* status;
    // spectra mode (receiver.equalise as a stand-alone stage): frequency-domain inputs
    const cplx* sp_data;      // [F, D, K]
    const cplx* sp_start;     // [F, P, K]
    const cplx* sp_end;       // [F, P, K]
    cplx* eq_all;             // [F*D, K] equalised symbols on all carriers
    double qpsk_q;            // >0: table is the reference QPSK table (+-q +-qj): decide by signs away from ties
    UniGrid ug;
    unsigned long long* stamps;   // diagnostic build only (-DGF3_STAMPS): [F][8] s_memtime per phase
    // two-phase form for long packets (gf3rx_demod_split.hip): the data stage runs nchunk workgroups per packet, each on
    // Dc consecutive data symbols, from the channel state the estimate stage left in Hs / He / slope
    int Dc, nchunk;
    const double* psum;       // [F][2][N] time-domain sums of each side's P pilot symbols (input of the estimate stage)
};

#ifdef GF3_STAMPS
#define GF3_STAMP(i) do { if (a.stamps && threadIdx.x == 0) { unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); a.stamps[blockIdx.x * 8 + (i)] = t_; } } while (0)
// slots 6 / 7: s_memrealtime (100 MHz) next to the first / last s_memtime stamp -> shader clock = d(memtime) / d(memrealtime) x 100 MHz
#define GF3_STAMP_RT(i) do { if (a.stamps && threadIdx.x == 0) { unsigned long long t_; \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); a.stamps[blockIdx.x * 8 + (i)] = t_; } } while (0)
#else
#define GF3_STAMP(i) do { } while (0)
#define GF3_STAMP_RT(i) do { } while (0)
#endif

// occupancy targets: min waves per SIMD handed to __launch_bounds__ (blocks of NC/8 threads).
// 2048 -> 3 blocks of 4 waves per CU (<=168 VGPRs), 4096 -> 1 block of 8 waves (<=256).
template <int NC> struct Occ { static constexpr int WPS = 2; };

struct CorrArgs {
    FftTables t;
    const void* in; int64_t n_in; int dt;
    const cplx* Hq;           // [Q][NC+1] spectra of the zero-padded chirp partitions
    int Q, Lp, Lc, Wmax;
    int64_t stride; int win_lo; int W;      // window f = chirp-start lags [f*stride + win_lo, +W)
    int64_t* starts; double* peak; double thresh;
    const int* list; const int* count;      // LISTED launches (the screened sync's unresolved windows): window = list[blockIdx.x], blockIdx.x < *count
};

// ============================================================================
// stream-mode matched filter: uniformly partitioned overlap-save with a spectral delay line
// (convolve(r, chirp[::-1]) over a whole stream, OFDM.py:357-358).
//   hop H = Lp (partition length).  Window j = samples [jH - (Lc-1), +N) is transformed ONCE
//   (spec_kernel); output block b = lags [bH, (b+1)H) of P is
//       P_b = irfft( sum_q  X_{b+q} . conj(H_q) )        (ols_kernel)
//   so the cost per H lags is one forward and one inverse transform plus Q spectrum MACs,
//   instead of Q+1 transforms.
// ============================================================================
#define OLS_B 3      /* output blocks per ols_kernel workgroup */
struct OlsArgs {
    FftTables t;
    const void* in; int64_t n_in; int dt;
    const cplx* Hq; int Q, H, Lc;
    cplx* spec;               // [NWIN][NC+1]
    int64_t nwin, plen;
    double* corr;
    double* part;             // [ols work items] maximum of the lags each workgroup wrote (the global max is max over these)
    int64_t nitems;           // logical work items of the launch (windows for spec_kernel, groups of OLS_B blocks for ols_kernel)
};

// ============================================================================
// transmit-side synthesiser: one packet per workgroup
// transmitter.map / build_OFDM_symbol / ifft / add_cp / send_to_stream (OFDM.py:196-259)
//   row f = [gap_f zeros | chirp Lc | P known symbols | D data symbols | P known symbols | zeros]
//   data symbol = 2 * irfft(X), X[bin] = point(label) on data carriers, filler on the others
// ============================================================================
struct TxArgs {
    FftTables t;
    int CP, S, P, D, K, C, mu, M, Lc;
    const int* pos;               // [K] data position of a carrier or -1
    int contig_lo;
    const double* cre; const double* cim; const int* idx_of_label;   // label -> table index
    const cplx* filler;           // [K] value of a non-data carrier (indexed by carrier)
    const double* chirp;          // [Lc]
    const double* known_time;     // [S] one pilot symbol with its prefix, before the x2 gain
    const uint8_t* bits; int row_bytes;       // [F, row_bytes] packed payload (np.packbits order)
    const int64_t* gaps;          // [F] leading zeros of each row (may be null)
    void* out; int64_t stride; int out_dt;    // [F, stride] samples, f32 or f64
};

// ============================================================================
// host side: context + C ABI
// ============================================================================
// a correlation plan has its own FFT size: the chirp search need not use the OFDM symbol's N
struct CorrPlan { int NC = 0, Q = 0, Lp = 0, W = 0; cplx* d_Hq = nullptr; FftTables t{nullptr, nullptr}; };

struct gf3_ctx {
    gf3_config cfg;
    int NC, K, S, Lc, row_bytes;
    int fit_lo, fit_hi;
    double xbar, inv_sxx;
    cplx *d_tw = nullptr, *d_twn = nullptr, *d_known = nullptr;
    cplx *d_tw_x[2] = {nullptr, nullptr}, *d_twn_x[2] = {nullptr, nullptr};   // twiddles of plans whose FFT size != N
    int nc_x[2] = {0, 0};
    int *d_pos = nullptr, *d_clab = nullptr;
    double *d_cre = nullptr, *d_cim = nullptr;
    CorrPlan frames_plan, stream_plan;
    double qpsk_q = 0.0;
    int* d_idx_of_label = nullptr;
    double* d_chirp = nullptr;
    double* d_chirp_t = nullptr;  // the same taps in the order scr_refine_kernel's lanes consume them (RefineArgs::chirp_t)
    double* d_known_time = nullptr;     // one pilot symbol in the time domain (transmit side)
    SepTab sep{};
    UniGrid ug{};
    unsigned long long* stamps = nullptr;
    int contig_lo = 0;
    int device = 0;                     // HIP device the context (tables, plans) lives on
    int n_cu = 256;                     // its compute units (grid sizing of the persistent-style kernels)
    // single-precision screening plan of the stream-mode sync (gf3rx_screen.h); ok = false: always the fp64 path
    struct { bool ok = false; int Q = 0, H = 0; cf *d_tw = nullptr, *d_twn = nullptr; float4* d_Hs = nullptr;
             float *d_H0N = nullptr, *d_Hinf = nullptr;
             bool ring = false; float4* d_Hb = nullptr; float* d_ecoef = nullptr; int R_forced = 0; } scr;   // band-limited kernel (scr_ring_kernel)
    // single-precision screening plan of the frames-mode sync (gf3rx_fscreen.h): 2048-sample transforms, partitions of 2048 - wmax + 1 taps
    struct { bool ok = false; int Q = 0, Lp = 0, wmax = 0; cf *d_tw = nullptr, *d_twn = nullptr; float4* d_Hs = nullptr;
             float *d_H0N = nullptr, *d_Hinf = nullptr; } fscr;
    // The ONLY field a call may write after gf3_ctx_create: the default evaluation mode of the legacy entry point
    // gf3_sync_stream (gf3_sync_stream_mode sets it; gf3_sync_stream_ex takes the mode per call and never reads it).
    // 0: by stream length (screen from GF3_SCR_MIN_SAMPLES on); 1: fp64 only; 2: screen whenever a plan exists; 3: as 2 with the general kernel
    std::atomic<int> default_stream_mode{0};
    std::vector<double> chirp;
    std::vector<cplx> known_pts;
};

// ============================================================================
// errors, device selection, launches
// ============================================================================
// Message of the calling thread's last failure.  One buffer per host thread, none in the context: concurrent calls
// on one context (different streams, different threads) cannot overwrite each other's text, and a failing call
// writes nothing into the context it was given.  (gf3rx_abi.hip owns the buffer and `fail`.)
int fail(const gf3_ctx*, int code, const char* fmt, ...);
#define HIPCHK(c, x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    return fail(c, GF3_EHIP, "%s failed: %s", #x, hipGetErrorString(e_)); } while (0)

// Every entry point that launches work runs on the context's device, whatever device the calling thread had
// current (kernels, copies and frees of a context created on cuda:1 must not land on cuda:0's streams).
struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(const gf3_ctx* c) {
        if (!c) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

inline size_t fft_lds_bytes(int NC) {          // == FftGeom<NC>::LDS_ELEMS
    return (size_t)((NC == 1024 || NC == 2048) ? 2 * NC : NC + NC / 8) * sizeof(cplx);
}

template <typename Kern, typename Args>
static hipError_t launch(Kern k, int64_t grid, int threads, size_t lds, hipStream_t st, const Args& a) {
    if (grid <= 0) return hipSuccess;
#ifdef GF3_LDS_PAD      /* diagnostic builds only: extra dynamic LDS to force a lower occupancy */
    lds += GF3_LDS_PAD;
#endif
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(threads), lds, st, a);
    return hipGetLastError();
}

#ifdef GF3_DEV_BUILD   /* developer iteration: only N=4096 with f32/f64 samples */
#define DISPATCH_DT(DTv, CALL)                                            \
    switch (DTv) {                                                        \
        case DT_F64: { constexpr int DTC = DT_F64; CALL; break; }         \
        default:     { constexpr int DTC = DT_F32; CALL; break; }         \
    }
#define DISPATCH_NC(NCv, DTv, CALL) { constexpr int NCC = 2048; DISPATCH_DT(DTv, CALL); }
#else
#define DISPATCH_DT(DTv, CALL)                                            \
    switch (DTv) {                                                        \
        case DT_F64: { constexpr int DTC = DT_F64; CALL; break; }         \
        case DT_F32: { constexpr int DTC = DT_F32; CALL; break; }         \
        case DT_I16: { constexpr int DTC = DT_I16; CALL; break; }         \
        default:     { constexpr int DTC = DT_U8;  CALL; break; }         \
    }
#define DISPATCH_NC(NCv, DTv, CALL)                                       \
    switch (NCv) {                                                        \
        case 512:  { constexpr int NCC = 512;  DISPATCH_DT(DTv, CALL); break; }   \
        case 1024: { constexpr int NCC = 1024; DISPATCH_DT(DTv, CALL); break; }   \
        case 2048: { constexpr int NCC = 2048; DISPATCH_DT(DTv, CALL); break; }   \
        default:   { constexpr int NCC = 4096; DISPATCH_DT(DTv, CALL); break; }   \
    }
#endif

// ============================================================================
// launchers exported by the kernel translation units
// ============================================================================
// gf3rx_fft.hip
hipError_t run_rfft_nc(int NCv, FftTables t, const void* d_in, int64_t n_in, int dt, const int64_t* d_off,
                       int64_t n_sym, cplx* d_out, hipStream_t st);
inline hipError_t run_rfft(const gf3_ctx* c, const void* d_in, int64_t n_in, int dt, const int64_t* d_off,
                           int64_t n_sym, cplx* d_out, hipStream_t st) {
    return run_rfft_nc(c->NC, FftTables{c->d_tw, c->d_twn}, d_in, n_in, dt, d_off, n_sym, d_out, st);
}
int tx_launch(gf3_ctx* c, const TxArgs& a, int64_t F, hipStream_t st);
// gf3rx_demod_{qpsk,scan,full}.hip: one packet per workgroup (time-domain input); `full` also serves the spectra mode
hipError_t launch_demod_qpsk(const gf3_ctx* c, const DemodArgs& a, int64_t F, hipStream_t st);
hipError_t launch_demod_scan(const gf3_ctx* c, const DemodArgs& a, int64_t F, hipStream_t st);
hipError_t launch_demod_full(const gf3_ctx* c, const DemodArgs& a, int64_t F, hipStream_t st);
hipError_t launch_demod_spectra(const gf3_ctx* c, const DemodArgs& a, int64_t F, hipStream_t st);
// gf3rx_demod_split.hip + gf3rx_dsplit_{qpsk,scan,full}.hip: the two-phase form for long packets, few at a time
bool demod_wants_split(const gf3_ctx* c, int64_t F, int mode);
int demod_split(const gf3_ctx* c, DemodArgs a, int64_t F, void* d_work, hipStream_t st);
hipError_t launch_dsplit_qpsk(const gf3_ctx* c, const DemodArgs& a, int64_t grid, hipStream_t st);
hipError_t launch_dsplit_scan(const gf3_ctx* c, const DemodArgs& a, int64_t grid, hipStream_t st);
hipError_t launch_dsplit_full(const gf3_ctx* c, const DemodArgs& a, int64_t grid, hipStream_t st);
// gf3rx_corr.hip
hipError_t run_corr(const gf3_ctx* c, const CorrPlan& pl, const CorrArgs& a, int64_t grid, hipStream_t st, bool listed = false);
// gf3rx_fscreen.hip: the screened frames sync (fp32 with a bound per window; unresolved windows are listed for corr_kernel)
struct FScreenArgs;
hipError_t launch_fscreen(const gf3_ctx* c, const FScreenArgs& a, int64_t F, hipStream_t st);
hipError_t run_spec_ols(const CorrPlan& pl, OlsArgs a, int64_t nwin, int64_t nblk, hipStream_t st);   // spec_kernel, then ols_kernel
// gf3rx_screen.hip
hipError_t launch_screen(const gf3_ctx* c, ScreenArgs a, bool general, hipStream_t st);
hipError_t launch_refine(const gf3_ctx* c, const RefineArgs& a, int64_t cap_cells, hipStream_t st);
