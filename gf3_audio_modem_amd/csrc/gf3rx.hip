// libgf3rx -- MI355X (gfx950) OFDM receive-path engine: kernels + C ABI.
// See include/gf3rx.h for the boundary and DESIGN.md for the layout.
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
#include "gf3rx_screen.h"

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
};

// ============================================================================
// standalone batched real FFT  (remove_cp + np.fft.fft, OFDM.py:407-408,593)
// ============================================================================
template <int NC, int DT>
__global__ __launch_bounds__(NC / 8, Occ<NC>::WPS) void rfft_kernel(RfftArgs a) {
    extern __shared__ double2 smem[];
    constexpr int T = NC / 8;
    const int tid = threadIdx.x;
    const int64_t sym = blockIdx.x;
    const int64_t off = a.off[sym];
    cplx* out = a.out + sym * (int64_t)(NC + 1);
    FftTw<NC> ft;
    ft.init(tid, a.t.tw);
    const cplx wb = a.t.twn[tid];
    cplx v[8];
    const bool ok = off >= 0 && off + 2 * NC <= a.n_in;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        RawPair<DT> raw;
        if (ok) raw.load(a.in, off + 2 * (int64_t)(tid + r * T)); else raw.zero();
        v[r] = raw.get();
    }
    cplx z0;
    rfft_regs<NC, false>(v, smem, ft, wb, tid, z0, 0);      // single in-place buffer: twice the resident workgroups
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2)
        if (Spec<NC>::live(tid, s2)) out[Spec<NC>::bin(tid, s2)] = v[s2];
    if (tid == 0) {
        out[0] = cmk(z0.x + z0.y, 0.0);
        out[NC] = cmk(z0.x - z0.y, 0.0);
    }
}

// ============================================================================
// fused demodulation of one packet per workgroup
// (get_symbols..PS, OFDM.py:391-505; equalise :422-480 is the bulk)
//
// Per thread: 8 carriers ("slots", Spec<NC>), whose channel state (unit phasor of Hs,
// |Hs|, |He|-|Hs|) stays in registers for the whole packet.  Symbols are processed
// start pilots -> end pilots -> data, the next symbol's raw samples being fetched
// while the current one is transformed.
// ============================================================================
// Decision of the reference's argmin over its QPSK table (+q,+q) (+q,-q) (-q,-q) (-q,+q) with labels 00 10 11 01
// (OFDM.py:72-77, 493-496; first minimum wins) for exact arithmetic, from the signs of e (or of any
// positive multiple of e): the scan's first-minimum rule breaks the four axis ties as
// Re=0 -> Re>=0 side, Im=0 -> (Re<0 ? Im<0 side : Im>=0 side); NaN/Inf -> first point.
GF3_DEV uint32_t qpsk_sign_rule(cplx e) {
    // common case: both components are non-zero finite numbers -> the label is the two sign bits
    const uint32_t hx = (uint32_t)__double2hiint(e.x), hy = (uint32_t)__double2hiint(e.y);
    uint32_t lab = ((hy >> 31) << 1) | (hx >> 31);
    // v_cmp_class: NaN (0x3), -inf (0x4), -0 (0x20), +0 (0x40), +inf (0x200)
    const bool odd = __builtin_amdgcn_class(e.x, 0x267) || __builtin_amdgcn_class(e.y, 0x267);
    if (odd) {
        const bool fin = (fabs(e.x) < INFINITY) && (fabs(e.y) < INFINITY);
        const uint32_t b1 = e.x < 0.0 ? 1u : 0u;
        const uint32_t b0 = (e.y < 0.0 || (e.y == 0.0 && e.x < 0.0)) ? 2u : 0u;
        lab = fin ? (b0 | b1) : 0u;
    }
    return lab;
}

// First-minimum scan over the whole table, deciding as `argmin(abs(sym - table))` does (OFDM.py:490-496).
// Squared distances order the points exactly as the reference's distances do unless two of them agree to within
// rounding; then (margin 1e-12 relative, four orders above the rounding of either form) the contenders are
// re-measured with the reference's own |.| (np_cabs, bit-identical) in table order and the first minimum wins --
// which also covers exact mid-points, where different squared distances round to the SAME |.|.
GF3_DEV int scan_table(cplx e, const double* cre, const double* cim, int M) {
    int best = 0;
    double dx = e.x - cre[0], dy = e.y - cim[0];
    double bd = dx * dx + dy * dy;
    for (int c = 1; c < M; ++c) {
        dx = e.x - cre[c]; dy = e.y - cim[c];
        const double d = dx * dx + dy * dy;
        if (d < bd) { bd = d; best = c; }
    }
    const double lim = bd * (1.0 + 1e-12);             // (NaN: every comparison false -> point 0, as argmin gives)
    bool tie = false;
    for (int c = 0; c < M; ++c) {
        dx = e.x - cre[c]; dy = e.y - cim[c];
        tie = tie || (c != best && dx * dx + dy * dy <= lim);
    }
    if (tie) {
        double hb = INFINITY;
        best = -1;
        for (int c = 0; c < M; ++c) {
            dx = e.x - cre[c]; dy = e.y - cim[c];
            if (dx * dx + dy * dy <= lim) {
                const double h = np_cabs(dx, dy);
                if (best < 0 || h < hb) { hb = h; best = c; }
            }
        }
        if (best < 0) best = 0;
    }
    return best;
}
// Nearest of n equally spaced levels lo, lo + 1/inv, ...  The caller hands in tp = (x - lo) inv + 1/2 (one fma on the
// un-normalised symbol: x = ep / mag, so tp = ep (inv / mag) + (1/2 - lo inv)); the level index is trunc(tp) clamped to
// the grid (v_cvt_i32_f64 truncates; below the grid it is clamped to 0 anyway, inside it trunc = floor) and its label
// byte is picked out of the packed table by one v_perm_b32.  `clear` is false within 1e-9 of a spacing of a decision
// boundary -- tp within 1e-9 of an integer -- and for NaN / Inf / |tp| >= 2^52, where the caller falls back to the
// literal scan; everywhere else the per-axis choice IS the argmin over the grid, with a margin five orders above the
// error of tp (one Newton step on v_rcp_f64: ~1e-14 relative).  (Beyond the outermost levels tp may be flagged although
// the edge level is certain: a spurious, harmless visit of the literal scan.)  The flag is formed by the caller from the
// two axes' `off` in one comparison: clear <=> max(|offI|, |offQ|) < 1/2 - 1e-9 (NaN compares false).
GF3_DEV uint32_t uni_axis(double tp, int n, unsigned long long pack, double& off) {
    off = __builtin_amdgcn_fract(tp) - 0.5;                                // |off| -> 1/2 at a boundary (the caller tests both axes at once)
    int r = (int)tp;
    asm("v_med3_i32 %0, %0, 0, %1" : "+v"(r) : "v"(n > 1 ? n - 1 : 0));    // clamp to the grid (never negative: r indexes `pack`)
    // byte r of the packed label table; the selector's upper bytes are zero and pick byte 0 into the result's upper
    // bytes, which nobody looks at: the label is stored with a byte store
    return __builtin_amdgcn_perm((uint32_t)(pack >> 32), (uint32_t)pack, (uint32_t)r);
}
// 1/x to ~1e-14 relative: v_rcp_f64 seed (2^-23) + one Newton step
GF3_DEV double rcp_n1(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    return fma(y, fma(-x, y, 1.0), y);
}
// MODE_FULL : per-symbol dumps (eq, eq_all, Hest) + literal table scan on the equalised symbol
// MODE_SCAN : bits only, any constellation: normalise and scan literally
// MODE_QPSK : bits only, reference QPSK table.  The channel magnitude model
//             (1-f)|Hs| + f|He| is positive, so X/Hest and X*conj(u*rot) have the same signs and
//             the decision needs neither the division nor |Hs|, |He|: per-carrier state is u alone.
enum { MODE_FULL = 0, MODE_SCAN = 1, MODE_QPSK = 2 };
#ifndef GF3_DEMOD_WPS
#define GF3_DEMOD_WPS 2
#endif
#ifndef GF3_ABL
#define GF3_ABL 0             /* timing-only ablations of the table modes (WRONG results): 1 no magnitude reads, 2 also ping-pong buffers */
#endif
// lean modes at GF3_DEMOD_WPS waves/SIMD; 3 needs the single in-place FFT buffer to fit 3 workgroups of LDS
// MODE_QPSK keeps the two ping-pong FFT buffers.  The table modes carry two more doubles of state per carrier (the
// magnitude model a0 + da f_l), which do not fit the register file next to the transform at two workgroups per CU;
// they live in LDS ([8][T] pairs, one conflict-free 16-byte read per carrier per symbol), and the room comes from the
// single in-place FFT buffer (one more barrier per exchange).
template <int NC, int MODE> struct DemodOcc {
    static constexpr int WPS = (MODE == MODE_QPSK && NC <= 2048) ? GF3_DEMOD_WPS : 2;
    static constexpr bool MAG_LDS = (MODE != MODE_QPSK);
#if GF3_ABL >= 2
    static constexpr bool PP = FftGeom<NC>::PINGPONG && WPS <= 2;
#else
    static constexpr bool PP = FftGeom<NC>::PINGPONG && WPS <= 2 && !MAG_LDS;
#endif
    static constexpr int LDS_ELEMS = PP ? FftGeom<NC>::LDS_ELEMS : FftGeom<NC>::LDS_ELEMS_INPLACE;
    static constexpr int MAG_ELEMS = MAG_LDS ? NC : 0;             // double2 (a0, da) per slot per thread: 8 * NC/8
};

template <int NC, int DT, bool SPECTRA, int MODE>
__global__ __launch_bounds__(NC / 8, (DemodOcc<NC, MODE>::WPS)) void demod_kernel(DemodArgs a) {
    constexpr bool FULL = (MODE == MODE_FULL);
    extern __shared__ double2 smem[];
    constexpr int T = NC / 8;
    // LDS: [scratch 32 doubles | start-up rotation tables | FFT buffer | decision bytes | (fit-range overflow)]
    double* scratch = (double*)smem;
    cplx* rtab = (cplx*)(scratch + 32);                                   // [2][64 + NC/64 + 1]
    cplx* lds = rtab + 2 * (64 + NC / 64 + 1);                            // FFT buffer, DemodOcc::LDS_ELEMS points
    uint8_t* labs = (uint8_t*)(lds + DemodOcc<NC, MODE>::LDS_ELEMS);      // [ring][C] decisions, one byte each
    // [8][T] (a0, da) of slot s of thread t (table modes), behind the decision bytes: written only after the
    // channel-estimate stage, whose fit-range arrays may run over this region
    double2* mags = (double2*)(labs + ((a.ring * a.C + 15) & ~15));
    const int tid = threadIdx.x;
    const int64_t f = blockIdx.x;
    const int K = a.K, P = a.P, D = a.D, S = a.S;
    const int Bs = a.C * a.mu;                                            // bits per data symbol
    uint8_t* row = a.bits + f * (int64_t)a.row_bytes;

    int64_t off = 0;
    if constexpr (!SPECTRA) {
        off = a.off[f];
        const bool ok = off >= 0 && off + (int64_t)(2 * P + D) * S <= a.n_in;
        if (!ok) {                                                        // ragged packet
            for (int i = tid; i < a.row_bytes; i += T) row[i] = 0;
            if (tid == 0 && a.status) atomicOr(a.status, 1);
            return;
        }
    }

    FftTw<NC> ft;
    ft.init(tid, a.t.tw);
    cplx wb = a.t.twn[tid];
    const int tq = tid;       // 32-bit per-thread indices derived from it may be hoisted: cheap in registers
    auto bin_of = [&](int s) { return Spec<NC>::bin(tq, s); };
    auto live_of = [&](int s) { return Spec<NC>::live(tq, s); };
    auto pos_of = [&](int s) {
        if (!live_of(s)) return -1;
        const int bn = bin_of(s);
        if (a.contig_lo > 0) return (bn >= a.contig_lo && bn < a.contig_lo + a.C) ? bn - a.contig_lo : -1;
        return a.pos[bn - 1];
    };

    // symbol order: start pilots, end pilots, data (position in the packet)
    auto sym_pos = [&](int i) { return i < P ? i : (i < 2 * P ? D + i : i - P); };
    RawPair<DT> nxt[8];
    auto fetch = [&](int i) {
        typedef typename RawT<DT>::E E;
        const E* base = (const E*)a.in + (off + (int64_t)sym_pos(i) * S + a.CP);      // wave-uniform
        const unsigned t2 = 2u * (unsigned)launder(tid);
#pragma unroll
        for (int r = 0; r < 8; ++r) nxt[r].load_u(base, t2 + 2u * (unsigned)(r * T));
    };
    const int Msym = 2 * P + D;
    cplx v[8], z0;
    auto transform = [&](int i) {                     // nxt -> spectrum slots in v; prefetch i+1
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = nxt[r].get();
        if (i + 1 < Msym) fetch(i + 1);
        ft.refresh();
        asm volatile("" : "+v"(wb.x), "+v"(wb.y));
        rfft_regs<NC, DemodOcc<NC, MODE>::PP, true>(v, lds, ft, wb, tq, z0, i & 1);
    };
    auto load_spectra = [&](const cplx* sp) {         // SPECTRA mode: slots straight from memory
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) v[s2] = sp[bin_of(s2) - 1];
    };
    GF3_STAMP(0);
    GF3_STAMP_RT(6);
    if constexpr (!SPECTRA) fetch(0);

    // ---- pilots: Hs, He = mean over P symbols / known  (OFDM.py:443-451).
    // The mean of the P pilot spectra is the spectrum of the mean pilot symbol (the DFT is linear), so
    // the P symbols of a side are summed in the time domain, sample by sample as they arrive, and ONE
    // transform per side replaces P (differs from the reference's order of additions by rounding only).
    cplx Hs[8], He[8];
    if constexpr (SPECTRA) {
#pragma unroll
        for (int s = 0; s < 8; ++s) Hs[s] = He[s] = cmk(0.0, 0.0);
        for (int i = 0; i < P; ++i) {
            load_spectra(a.sp_start + ((int64_t)f * P + i) * K);
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) Hs[s2] = cadd(Hs[s2], v[s2]);
            load_spectra(a.sp_end + ((int64_t)f * P + i) * K);
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) He[s2] = cadd(He[s2], v[s2]);
        }
    } else {
        for (int side = 0; side < 2; ++side) {
            cplx sum[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) sum[r] = cmk(0.0, 0.0);
            for (int i = side * P; i < (side + 1) * P; ++i) {
#pragma unroll
                for (int r = 0; r < 8; ++r) sum[r] = cadd(sum[r], nxt[r].get());
                if (i + 1 < Msym) fetch(i + 1);                    // next pilot, or the first data symbol
            }
            if (side == 0) GF3_STAMP(1);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = sum[r];
            ft.refresh();
            asm volatile("" : "+v"(wb.x), "+v"(wb.y));
            rfft_regs<NC, DemodOcc<NC, MODE>::PP, true>(v, lds, ft, wb, tq, z0, side);
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) { if (side) He[s2] = v[s2]; else Hs[s2] = v[s2]; }
        }
    }

    // ---- per carrier: H = mean / known; equaliser state
    //      Hest = (a0 + da f_l) u exp(j slope n f_l),  u = Hs/|Hs| = exp(j angle(Hs))
    // Phase slope (OFDM.py:454-462): y_n = unwrap(angle He)_n - unwrap(angle Hs)_n fitted over
    // n in [fit_lo, fit_hi).  With c_n = sum_{i<=n} e_i the cumulative unwrap corrections,
    //   sum_n xm_n c_n = sum_i e_i S_i,  S_i = sum_{n>=i} xm_n = j (L - j) / 2,  j = i - fit_lo,
    // so the fit needs no prefix scan, and corrections before fit_lo (a common offset of every
    // fitted point) drop out: only carriers inside the fit range need their angles.
    GF3_STAMP(2);
    lds_barrier();                                    // FFT buffer is free: reuse it for the fit-range carriers
    const int L = a.fit_hi - a.fit_lo;
    // [L] Hs of carrier fit_lo + j, later its angle in .x.  The two arrays start at the FFT buffer and may run on
    // over the decision bytes (not in use before the first data symbol) and beyond: demod_lds_bytes sizes it.
    cplx* hsl = lds;
    cplx* hel = hsl + L;                              // [L] same for He
    cplx u[8];
    double a0[8], da[8];                              // (table modes; parked in LDS once the fit is done)
    // The transforms of this kernel leave 2 X in the slots (rfft_regs<.., TWICE>): XS is that factor (1 when the
    // spectra come from memory).  It is divided out of the pilots here and carried by the magnitudes a0, da, so the
    // channel estimates are true-scale and X/Hest is unchanged -- bit for bit, powers of two being exact.
    constexpr double XS = SPECTRA ? 1.0 : 2.0;
    const double invP = (1.0 / (double)P) / XS;
    // (a) straight-line over the 8 slots (independent chains overlap): H = mean/known, unit phasor, magnitudes
    cplx ik[8];                                       // 1/known (L2 latency covered by the other resident workgroup)
#pragma unroll
    for (int s = 0; s < 8; ++s) ik[s] = a.inv_known[bin_of(s) - 1];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        Hs[s] = cmul(cscale(Hs[s], invP), ik[s]);
        He[s] = cmul(cscale(He[s], invP), ik[s]);
        const double m2 = Hs[s].x * Hs[s].x + Hs[s].y * Hs[s].y;
        const double ia = rsq_nr(m2);                                 // 1/|Hs|
        if constexpr (MODE != MODE_QPSK) {
            const double e2 = He[s].x * He[s].x + He[s].y * He[s].y;
            const double ah = m2 * ia;                                // |Hs|
            a0[s] = XS * ah;
            da[s] = XS * (e2 * rsq_nr(e2) - ah);                      // XS (|He| - |Hs|)
        }
        u[s] = cmk(Hs[s].x * ia, Hs[s].y * ia);
    }
    // (b) optional dumps; the carriers inside the fit range go to LDS, where the angles are taken by
    //     whichever thread the carrier falls to (2 L angles per packet instead of 16 per thread)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int bn = bin_of(s);
        if (live_of(s)) {
            if (a.Hs) a.Hs[f * K + bn - 1] = Hs[s];
            if (a.He) a.He[f * K + bn - 1] = He[s];
            const int j = bn - 1 - a.fit_lo;
            if (j >= 0 && j < L) { hsl[j] = Hs[s]; hel[j] = He[s]; }
        }
    }
    lds_barrier();
    for (int j = launder(tid); j < L; j += T) {
        const cplx h0 = hsl[j], h1 = hel[j];
        hsl[j].x = atan2_fast(h0.y, h0.x);
        hel[j].x = atan2_fast(h1.y, h1.x);
    }
    lds_barrier();
    double slope;
    {
        double acc = 0.0;
        for (int j = launder(tid); j < L; j += T) {
            const double q0 = hsl[j].x, q1 = hel[j].x;
            acc += ((double)j - a.xbar) * (q1 - q0);
            if (j > 0) {
                const double e0 = unwrap_corr(q0 - hsl[j - 1].x);
                const double e1 = unwrap_corr(q1 - hel[j - 1].x);
                acc += (e1 - e0) * (0.5 * (double)j * (double)(L - j));
            }
        }
        slope = block_sum(acc, scratch + 16) * a.inv_sxx;
    }
    if (tid == 0 && a.slope) a.slope[f] = slope;
    GF3_STAMP(3);
    if constexpr (DemodOcc<NC, MODE>::MAG_LDS && GF3_ABL == 0) {      // (block_sum's barriers: every thread is done with the fit-range arrays)
#pragma unroll
        for (int s = 0; s < 8; ++s) mags[s * T + tid] = make_double2(a0[s], da[s]);
    }

    // ---- data symbols: FFT -> /Hest -> demap -> bit-pack (OFDM.py:466-478, 487-505)
    // Decisions are staged as one byte per data carrier in a ring of `ring` symbols in LDS and
    // packed into output words one symbol later (after the next FFT's barriers), so the
    // packing needs no atomics and no barrier of its own.  While symbol l is being decided, the words
    // completed by symbol l-1 are packed; the first of them starts up to 31 bits before that symbol, i.e.
    // ceil(32 / (C mu)) symbols back, and none of those slots may be the one symbol l is written to:
    // ring >= ceil(32 / (C mu)) + 2 (rounded up to a power of two on the host, demod_ring).
    const int C = a.C, mu = a.mu;
    const int RC = a.ring * C;
    auto pack_words = [&](int l, bool tail) {
        const int wlo = (l * Bs) >> 5, whi = ((l + 1) * Bs) >> 5;
        const int nlab = D * C;                                            // labels in the packet
        if (mu == 2 && (C & 1) == 0) {
            // QPSK, even C: a word is 16 labels = 4 aligned dwords of the ring; one multiply moves
            // the four 2-bit labels of a dword into a byte (label i -> bits 7-2i of it)
            for (int w = wlo + launder(tid); w < whi + (tail ? 1 : 0); w += T) {
                const int i0 = 16 * w;
                int r = i0 % RC;
                uint32_t x = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t d4 = (i0 + 4 * j < nlab) ? *(const uint32_t*)(labs + r) : 0u;
                    if (i0 + 4 * j + 4 > nlab) d4 &= 0xffffffffu >> (8 * (i0 + 4 * j + 4 - nlab));   // C even => whole pairs
                    x = (x << 8) | ((d4 * 0x40100401u) >> 24);
                    r += 4; if (r >= RC) r -= RC;
                }
                if (w < whi) {
                    if ((a.row_bytes & 3) == 0) ((uint32_t*)row)[w] = __builtin_bswap32(x);
                    else { row[4 * w] = x >> 24; row[4 * w + 1] = x >> 16; row[4 * w + 2] = x >> 8; row[4 * w + 3] = x; }
                } else {
                    const int rem = (D * Bs) & 31;
                    for (int bb = 0; bb < ((rem + 7) >> 3); ++bb) row[4 * w + bb] = (uint8_t)(x >> (24 - 8 * bb));
                }
            }
            return;
        }
        for (int w = wlo + launder(tid); w < whi + (tail ? 1 : 0); w += T) {
            int i = (32 * w) / mu;                                         // first label touching the word
            const int skip = 32 * w - i * mu;
            int r = i % RC;
            uint64_t acc = 0;
            int nb = 0;
            while (nb < skip + 32) {
                const uint32_t lb = (i < nlab) ? labs[r] : 0u;
                acc = (acc << mu) | lb;
                nb += mu; ++i;
                if (++r == RC) r = 0;
            }
            const uint32_t x = (uint32_t)(acc >> (nb - skip - 32));
            if (w < whi) {
                if ((a.row_bytes & 3) == 0) ((uint32_t*)row)[w] = __builtin_bswap32(x);
                else { row[4 * w] = x >> 24; row[4 * w + 1] = x >> 16; row[4 * w + 2] = x >> 8; row[4 * w + 3] = x; }
            } else {                                                       // partial last word of the packet
                const int rem = (D * Bs) & 31;
                for (int bb = 0; bb < ((rem + 7) >> 3); ++bb) row[4 * w + bb] = (uint8_t)(x >> (24 - 8 * bb));
            }
        }
    };
    const double denom = (double)(D + P);
    // Channel-model phasor per carrier: Hest = mag * g_l,  g_l = u exp(j slope n f_l),  f_l = (l + P/2)/(D+P)
    // is linear in l, so g_{l+1} = g_l * exp(j slope n / (D+P)): one complex multiply per carrier per symbol
    // and no sin/cos inside the symbol loop.  The two start-up rotations exp(j phi0 n), exp(j dphi n) come
    // from small two-level tables (n + 1 = 64 h + i  ->  T[64 + h] * T[i]) built once per packet.
    constexpr int NTH = NC / 64 + 1, NRT = 64 + NTH;
    {
        const double phi0 = slope * ((0.5 * (double)P) / denom), dphi = slope / denom;
        for (int i = tid; i < NRT; i += T) {
            const double nn = (double)(i < 64 ? i - 1 : 64 * (i - 64));
            rtab[i] = cis_fast(phi0 * nn);
            rtab[NRT + i] = cis_fast(dphi * nn);
        }
    }
    lds_barrier();
    cplx gstep[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int n1 = bin_of(s);                                          // n + 1
        const cplx r0 = cmul(rtab[64 + (n1 >> 6)], rtab[n1 & 63]);
        gstep[s] = cmul(rtab[NRT + 64 + (n1 >> 6)], rtab[NRT + (n1 & 63)]);
        u[s] = cmul(u[s], r0);                                             // u now holds g_0
    }
    // data position of every slot, resolved once: a lookup inside the symbol loop would put a vmcnt(0) wait
    // behind the packed-word stores and the next symbol's prefetch
    // (QPSK mode; the table modes have no registers to spare for it -- even packed two to a register the positions push
    //  the kernel from 240 VGPRs to 256 and into spills -- and recompute the position per symbol: plain arithmetic for a
    //  contiguous band)
    int psl[8];
    if constexpr (MODE == MODE_QPSK) {
#pragma unroll
        for (int s = 0; s < 8; ++s) psl[s] = pos_of(s);
    }
    for (int l = 0; l < D; ++l) {
        if constexpr (SPECTRA) { lds_barrier(); load_spectra(a.sp_data + ((int64_t)f * D + l) * K); }
        else transform(2 * P + l);
        if (l > 0) pack_words(l - 1, false);
        const double fl = ((double)l + 0.5 * (double)P) / denom;          // (l + P/2)/(D+P)
        uint8_t* lab_l = labs + (l & (a.ring - 1)) * C;
        if constexpr (MODE == MODE_QPSK) {
            // all eight carriers in one straight line: rotate, advance the phasors, take the sign bits; the exact
            // tie / NaN / Inf rule is one rarely taken branch for the whole group instead of one per carrier
            cplx ep[8];
            uint32_t lab[8];
            bool odd = false;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const cplx g = u[s];
                ep[s] = cmul_conj(v[s], g);
                u[s] = cmul(g, gstep[s]);
                lab[s] = (((uint32_t)__double2hiint(ep[s].y) >> 31) << 1) | ((uint32_t)__double2hiint(ep[s].x) >> 31);
                odd = odd || __builtin_amdgcn_class(ep[s].x, 0x267) || __builtin_amdgcn_class(ep[s].y, 0x267);
            }
            if (odd) {
#pragma unroll
                for (int s = 0; s < 8; ++s) lab[s] = qpsk_sign_rule(ep[s]);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) if (psl[s] >= 0) lab_l[psl[s]] = (uint8_t)lab[s];
        } else {
            // Fast path, straight-line over the eight carriers: equalise, then the nearest grid point per axis
            // (decide_fast).  A decision within 1e-9 of a spacing of a boundary, a NaN / Inf symbol or a table that
            // is not a uniform grid is only MARKED here; the marked carriers are re-decided below by the literal scan.
            uint32_t unclear = 0;
            const double cI = 0.5 - a.ug.loI * a.ug.invI, cQ = 0.5 - a.ug.loQ * a.ug.invQ;      // (wave-uniform)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int n = bin_of(s) - 1;
                const cplx g = u[s];                                       // unit phasor of Hest for this symbol
                const cplx ep = cmul_conj(v[s], g);                        // X / g  = e * mag
                u[s] = cmul(g, gstep[s]);                                  // ... and for the next one
                const int ps = pos_of(s);
#if GF3_ABL >= 1
                const double2 md = make_double2(1.0 + 1e-3 * s, 1e-4);
#else
                const double2 md = mags[s * T + tid];
#endif
                const double mag = fma(md.y, fl, md.x);
                // bits only: the decision needs (ep / mag - lo) inv to far less than full precision (the margin of `clear`
                // is 1e-9), so one Newton step serves and the quotient itself is never formed; the dumps of MODE_FULL
                // get the fully rounded reciprocal
                const double rm = FULL ? rcp_nr(mag) : rcp_n1(mag);
                if constexpr (FULL) {
                    const cplx e = cscale(ep, rm);
                    if (live_of(s)) {
                        if (a.Hest) a.Hest[((int64_t)f * D + l) * K + n] = cscale(g, mag * (1.0 / XS));
                        if (a.eq_all) a.eq_all[((int64_t)f * D + l) * K + n] = e;
                    }
                    if (ps >= 0 && a.eq) a.eq[((int64_t)f * D + l) * C + ps] = e;
                }
                if (ps >= 0) {
                    if (a.ug.nI > 0) {                                     // (wave-uniform) a grid with equally spaced levels
                        double oi, oq;
                        const uint32_t li = uni_axis(fma(ep.x, rm * a.ug.invI, cI), a.ug.nI, a.ug.packI, oi);
                        const uint32_t lq = uni_axis(fma(ep.y, rm * a.ug.invQ, cQ), a.ug.nQ, a.ug.packQ, oq);
                        lab_l[ps] = (uint8_t)(li | lq);
                        if (!(fmax(fabs(oi), fabs(oq)) < 0.5 - 1e-9)) unclear |= 1u << s;
                    } else unclear |= 1u << s;                             // any other table: every data carrier takes the literal scan
                }
            }
            // Rare path, one marked carrier at a time (no unrolling: nothing of the transform is live here, and the
            // slot's operands are picked out of the register arrays by selects).  The symbol is rebuilt from the
            // spectrum still in v[]: the phasor has already been advanced, g = u * conj(gstep) undoes that.
            for (uint32_t m = unclear; m; m &= m - 1) {
                const int s = __ffs((int)m) - 1;
                cplx vs = v[0], us = u[0], gs = gstep[0];
                const double2 md = mags[s * T + tid];
#pragma unroll
                for (int k = 1; k < 8; ++k) {
                    const bool hit = (s == k);
                    vs = cmk(hit ? v[k].x : vs.x, hit ? v[k].y : vs.y);
                    us = cmk(hit ? u[k].x : us.x, hit ? u[k].y : us.y);
                    gs = cmk(hit ? gstep[k].x : gs.x, hit ? gstep[k].y : gs.y);
                }
                const int bn = Spec<NC>::bin(tid, s);                      // (a marked slot is live and a data carrier)
                const int ps = a.contig_lo > 0 ? bn - a.contig_lo : a.pos[bn - 1];
                const cplx g = cmul_conj(us, gs);
                const cplx e = cscale(cmul_conj(vs, g), rcp_nr(fma(md.y, fl, md.x)));
                lab_l[ps] = (uint8_t)a.clab[scan_table(e, a.cre, a.cim, a.M)];
            }
        }
    }
    GF3_STAMP(4);
    lds_barrier();
    pack_words(D - 1, ((D * Bs) & 31) != 0);
    GF3_STAMP(5);
    GF3_STAMP_RT(7);
}

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
template <int NC, int DT>
__global__ __launch_bounds__(NC / 8, (NC <= 2048 ? GF3_CORR_WPS : 2)) void corr_kernel(CorrArgs a) {
    extern __shared__ double2 smem[];
    constexpr int T = NC / 8;
    constexpr bool PP = GF3_CORR_PP && FftGeom<NC>::PINGPONG;
    cplx* lds = smem;
    double* scratch = (double*)(smem + (PP ? FftGeom<NC>::LDS_ELEMS : FftGeom<NC>::LDS_ELEMS_INPLACE));
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
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

// ============================================================================
// stream-mode matched filter: uniformly partitioned overlap-save with a spectral delay line
// (convolve(r, chirp[::-1]) over a whole stream, OFDM.py:357-358).
//   hop H = Lp (partition length).  Window j = samples [jH - (Lc-1), +N) is transformed ONCE
//   (spec_kernel); output block b = lags [bH, (b+1)H) of P is
//       P_b = irfft( sum_q  X_{b+q} . conj(H_q) )        (ols_kernel)
//   so the cost per H lags is one forward and one inverse transform plus Q spectrum MACs,
//   instead of Q+1 transforms.
// ============================================================================
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

#define OLS_B 3      /* output blocks per workgroup */
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

template <int NC>
__global__ __launch_bounds__(NC / 8, 2) void tx_kernel(TxArgs a) {
    extern __shared__ double2 smem[];
    constexpr int T = NC / 8, N = 2 * NC;
    cplx* lds = smem;
    const int tid = threadIdx.x;
    const int64_t f = blockIdx.x;
    const int S = a.S, P = a.P, D = a.D;
    const int64_t g = a.gaps ? a.gaps[f] : 0;
    auto put = [&](int64_t i, double x) {
        if (a.out_dt == DT_F32) ((float*)a.out)[f * a.stride + i] = (float)x;
        else ((double*)a.out)[f * a.stride + i] = x;
    };
    const int64_t body = g + a.Lc;
    const int64_t used = body + (int64_t)(2 * P + D) * S;
    for (int64_t i = tid; i < a.stride; i += T) {
        if (i < g || i >= used) put(i, 0.0);
        else if (i < body) put(i, a.chirp[i - g]);
    }
    for (int p = 0; p < 2 * P; ++p) {                       // known symbols, x2 gain (OFDM.py:253-256)
        const int64_t s0 = body + (int64_t)(p < P ? p : D + p) * S;
        for (int i = tid; i < S; i += T) put(s0 + i, 2.0 * a.known_time[i]);
    }
    FftTw<NC> ft;
    ft.init(tid, a.t.tw);
    cplx wb = a.t.twn[tid];
    const uint8_t* brow = a.bits + f * (int64_t)a.row_bytes;
    auto point_of = [&](int l, int bn) -> cplx {            // value of FFT bin bn (1..K) of data symbol l
        int ps;
        if (a.contig_lo > 0) ps = (bn >= a.contig_lo && bn < a.contig_lo + a.C) ? bn - a.contig_lo : -1;
        else ps = a.pos[bn - 1];
        if (ps < 0) return a.filler[bn - 1];
        const int o = (l * a.C + ps) * a.mu;                // first bit of the label, MSB-first stream
        const int b0 = o >> 3, bl = a.row_bytes - 1;          // a label never extends past the row; clamp the look-ahead
        uint32_t w = ((uint32_t)brow[b0] << 16) | ((uint32_t)brow[min(b0 + 1, bl)] << 8) | (uint32_t)brow[min(b0 + 2, bl)];
        const uint32_t lab = (w >> (24 - a.mu - (o & 7))) & ((1u << a.mu) - 1u);
        const int ix = a.idx_of_label[lab];
        return cmk(a.cre[ix], a.cim[ix]);
    };
    for (int l = 0; l < D; ++l) {
        const int tq = launder(tid);
        lds_barrier();                                      // previous symbol fully written out
        // Hermitian half-spectrum X[0..NC] (X[0] = X[NC] = 0) -> packed spectrum Z of the NC-point
        // complex IFFT whose output interleaves even/odd samples (inverse of real_split)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = Spec<NC>::bin(tq, 2 * r);
            const cplx A = point_of(l, k);
            const cplx Bc = (k == NC - k) ? A : point_of(l, NC - k);
            const cplx B = cconj(Bc);
            const cplx E = cscale(cadd(A, B), 0.5);
            const cplx Op = cmul_conj(cscale(csub(A, B), 0.5), Spec<NC>::pair_tw(tq, r, wb));
            const cplx Zk = cadd(E, mul_posi(Op));
            const cplx Zm = cadd(cconj(E), mul_posi(cconj(Op)));
            lds[k] = cconj(Zk);
            if (Spec<NC>::live(tq, 2 * r + 1)) lds[NC - k] = cconj(Zm);
        }
        if (tid == 0) lds[0] = cmk(0.0, 0.0);
        lds_barrier();
        cplx v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = lds[tid + r * T];
        lds_barrier();
        ft.refresh();
        cplx* yb = fft_core<NC>(v, lds, ft, launder(tid));
        // y[2n] = Re z / NC, y[2n+1] = -Im z / NC (z = conj of the forward FFT of conj Z); gain 2
        const double sc = 2.0 / (double)NC;
        const int64_t s0 = body + (int64_t)(P + l) * S;
        for (int i = tid; i < NC; i += T) {
            const cplx z = yb[i];
            const double y0 = z.x * sc, y1 = -z.y * sc;
            put(s0 + a.CP + 2 * i, y0);
            put(s0 + a.CP + 2 * i + 1, y1);
            const int j = 2 * i - (N - a.CP);               // cyclic prefix = last CP samples (OFDM.py:221-226)
            if (j >= 0) put(s0 + j, y0);
            if (j + 1 >= 0) put(s0 + j + 1, y1);
        }
    }
}

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
    // The ONLY field a call may write after gf3_ctx_create: the default evaluation mode of the legacy entry point
    // gf3_sync_stream (gf3_sync_stream_mode sets it; gf3_sync_stream_ex takes the mode per call and never reads it).
    // 0: by stream length (screen from GF3_SCR_MIN_SAMPLES on); 1: fp64 only; 2: screen whenever a plan exists; 3: as 2 with the general kernel
    std::atomic<int> default_stream_mode{0};
    std::vector<double> chirp;
    std::vector<cplx> known_pts;
};

// Message of the calling thread's last failure.  One buffer per host thread, none in the context: concurrent calls
// on one context (different streams, different threads) cannot overwrite each other's text, and a failing call
// writes nothing into the context it was given.
static thread_local char g_err[512] = "";
// ... and the diagnostics of the calling thread's last gf3_sync_stream / gf3_sync_stream_ex (gf3_sync_stream_info)
static thread_local int64_t g_last_info[4] = {0, 0, 0, 0};
// Where a call reads its few result words back to: 256 bytes of PINNED host memory per calling thread (allocated on the
// thread's first call, never freed: a copy into pageable memory goes through the runtime's staging path, which costs a call
// tens of microseconds).  nullptr if the allocation fails -- the caller then copies into a local variable as before.
static void* readback_buffer() {
    static thread_local void* p = nullptr;
    static thread_local bool tried = false;
    if (!tried) { tried = true; if (hipHostMalloc(&p, 256, hipHostMallocPortable) != hipSuccess) { p = nullptr; (void)hipGetLastError(); } }
    return p;
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

// Every entry point that launches work runs on the context's device, whatever device the calling thread had
// current (kernels, copies and frees of a context created on cuda:1 must not land on cuda:0's streams).
struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(const gf3_ctx* c);
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

static int fail(const gf3_ctx*, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(c, x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    return fail(c, GF3_EHIP, "%s failed: %s", #x, hipGetErrorString(e_)); } while (0)

DeviceGuard::DeviceGuard(const gf3_ctx* c) {
    if (!c) return;
    if (hipGetDevice(&prev) == hipSuccess && prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
}

template <typename T> static hipError_t upload(T** dptr, const T* h, size_t n) {
    hipError_t e = hipMalloc((void**)dptr, n * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*dptr, h, n * sizeof(T), hipMemcpyHostToDevice);
}

static size_t fft_lds_bytes(int NC) {          // == FftGeom<NC>::LDS_ELEMS
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

static hipError_t run_rfft_nc(int NCv, FftTables t, const void* d_in, int64_t n_in, int dt, const int64_t* d_off,
                              int64_t n_sym, cplx* d_out, hipStream_t st) {
    RfftArgs a{t, d_in, n_in, d_off, dt, d_out};
    hipError_t e = hipSuccess;
#ifdef GF3_DEV_BUILD
    if (NCv == 1024) {
        if (dt == DT_F64) return launch((rfft_kernel<1024, DT_F64>), n_sym, 128, (size_t)(1024 + 128) * sizeof(cplx), st, a);
        return launch((rfft_kernel<1024, DT_F32>), n_sym, 128, (size_t)(1024 + 128) * sizeof(cplx), st, a);
    }
#endif
    DISPATCH_NC(NCv, dt, e = launch((rfft_kernel<NCC, DTC>), n_sym, NCC / 8, (size_t)(NCC + NCC / 8) * sizeof(cplx), st, a));
    return e;
}
static hipError_t run_rfft(const gf3_ctx* c, const void* d_in, int64_t n_in, int dt, const int64_t* d_off,
                           int64_t n_sym, cplx* d_out, hipStream_t st) {
    return run_rfft_nc(c->NC, FftTables{c->d_tw, c->d_twn}, d_in, n_in, dt, d_off, n_sym, d_out, st);
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

extern "C" const char* gf3_version(void) { return GF3RX_VERSION; }

// SHA-256 of the sources and flags this binary was built from (gf3_audio_modem_amd/build.py passes it in; "unknown" for a
// build made by hand).  The loader compares it with the sources as they are now -- the library carries its own stamp,
// no side file -- and finds it by the marker without loading the library.
#ifndef GF3_SRC_HASH
#define GF3_SRC_HASH "unknown"
#endif
extern "C" const char gf3_src_hash_marker[] = "GF3_SRC_HASH=" GF3_SRC_HASH;
extern "C" const char* gf3_source_hash(void) { return gf3_src_hash_marker + 13; }

extern "C" const char* gf3_last_error(const gf3_ctx*) { return g_err; }

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
    if (rc != GF3_OK) { gf3_ctx_destroy(c); return rc; }
    *out = c;
    return GF3_OK;
}

extern "C" void gf3_ctx_destroy(gf3_ctx* c) {
    if (!c) return;
    DeviceGuard dg(c);
    void* ptrs[] = {c->d_tw_x[0], c->d_twn_x[0], c->d_tw_x[1], c->d_twn_x[1], c->d_tw, c->d_twn, c->d_known, c->d_pos, c->d_clab, c->d_cre, c->d_cim,
                    c->frames_plan.d_Hq, c->stream_plan.d_Hq, c->d_idx_of_label, c->d_chirp, c->d_chirp_t, c->d_known_time,
                    c->scr.d_tw, c->scr.d_twn, c->scr.d_Hs, c->scr.d_H0N, c->scr.d_Hinf, c->scr.d_Hb, c->scr.d_ecoef};
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

// symbols in the decision-byte ring of demod_kernel: ceil(32 / (C mu)) + 2, rounded up to a power of two
// (>= 4: the QPSK packer reads the ring one aligned dword at a time, so ring * C must be a multiple of 4)
static int demod_ring(const gf3_ctx* c) {
    const int Bs = c->cfg.C * c->cfg.mu;
    const int need = (32 + Bs - 1) / Bs + 2;
    int r = 4;
    while (r < need) r <<= 1;
    return r;
}
// lean = MODE_QPSK (ping-pong FFT buffers); the table modes use the in-place buffer and NC (a0, da) pairs.
// Layout: [scratch 32 doubles | rotation tables | FFT buffer | decision bytes | (a0, da) pairs]; everything from the
// FFT buffer on is overlaid by Hs, He of the fit range during the channel-estimate stage (which may need more).
static size_t demod_lds_bytes(const gf3_ctx* c, bool lean = false) {
    const bool inplace = !lean || (GF3_DEMOD_WPS > 2 && c->NC <= 2048);
    const size_t fft = inplace ? (size_t)(c->NC + c->NC / 8) * sizeof(cplx) : fft_lds_bytes(c->NC);
    const size_t mags = lean ? 0 : (size_t)c->NC * sizeof(double2);
    const size_t tail = fft + (size_t)((demod_ring(c) * c->cfg.C + 15) & ~15) + mags;
    const size_t fit = (size_t)2 * (c->fit_hi - c->fit_lo) * sizeof(cplx);
    return 32 * sizeof(double) + (size_t)2 * (64 + c->NC / 64 + 1) * sizeof(cplx) + (fit > tail ? fit : tail);
}

extern "C" int gf3_demod_frames(gf3_ctx* c, const void* d_in, int64_t n_in, const int64_t* d_off, int64_t F,
                                uint8_t* d_bits, void* d_eq, void* d_Hs, void* d_He, double* d_slope, void* d_Hest,
                                int32_t* d_status, void* stream) {
    DeviceGuard dg(c);
    if (c && F == 0) return GF3_OK;
    if (!c || !d_in || !d_off || !d_bits || F < 0) return fail(c, GF3_EINVAL, "gf3_demod_frames: bad argument");
    const gf3_config& g = c->cfg;
    DemodArgs a{{c->d_tw, c->d_twn}, d_in, n_in, d_off, g.in_dtype,
                g.CP, c->S, g.P, g.D, c->K, g.C, g.mu, g.M,
                c->d_known, c->d_pos, c->contig_lo, demod_ring(c), c->d_cre, c->d_cim, c->d_clab,
                c->fit_lo, c->fit_hi, c->xbar, c->inv_sxx,
                d_bits, c->row_bytes, (cplx*)d_eq, (cplx*)d_Hs, (cplx*)d_He, d_slope, (cplx*)d_Hest, d_status,
                nullptr, nullptr, nullptr, nullptr, c->qpsk_q, c->ug, c->stamps};
    hipError_t e = hipSuccess;
    if (d_eq || d_Hest) {
        DISPATCH_NC(c->NC, g.in_dtype, e = launch((demod_kernel<NCC, DTC, false, MODE_FULL>), F, NCC / 8, demod_lds_bytes(c), (hipStream_t)stream, a));
    } else if (c->qpsk_q > 0.0) {
        DISPATCH_NC(c->NC, g.in_dtype, e = launch((demod_kernel<NCC, DTC, false, MODE_QPSK>), F, NCC / 8, demod_lds_bytes(c, true), (hipStream_t)stream, a));
    } else {
        DISPATCH_NC(c->NC, g.in_dtype, e = launch((demod_kernel<NCC, DTC, false, MODE_SCAN>), F, NCC / 8, demod_lds_bytes(c, GF3_ABL >= 2), (hipStream_t)stream, a));
    }
    HIPCHK(c, e);
    return GF3_OK;
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
                (const cplx*)d_data, (const cplx*)d_start, (const cplx*)d_end, (cplx*)d_eq_all, c->qpsk_q, c->ug, nullptr};
    hipError_t e = hipSuccess;
    switch (c->NC) {
#ifndef GF3_DEV_BUILD
        case 512:  e = launch((demod_kernel<512, DT_F64, true, MODE_FULL>), F, 64, demod_lds_bytes(c), (hipStream_t)stream, a); break;
        case 1024: e = launch((demod_kernel<1024, DT_F64, true, MODE_FULL>), F, 128, demod_lds_bytes(c), (hipStream_t)stream, a); break;
        case 4096: e = launch((demod_kernel<4096, DT_F64, true, MODE_FULL>), F, 512, demod_lds_bytes(c), (hipStream_t)stream, a); break;
#endif
        default:   e = launch((demod_kernel<2048, DT_F64, true, MODE_FULL>), F, 256, demod_lds_bytes(c), (hipStream_t)stream, a); break;
    }
    HIPCHK(c, e);
    return GF3_OK;
}

// known pilot symbol in the time domain (with prefix), built with the TX kernel itself:
// a one-symbol packet whose "filler" is the known-symbol vector and which has no data carriers.
static int tx_launch(gf3_ctx* c, const TxArgs& a, int64_t F, hipStream_t st) {
    const size_t lds = fft_lds_bytes(c->NC);
    hipError_t e = hipSuccess;
    switch (c->NC) {
#ifndef GF3_DEV_BUILD
        case 512:  e = launch(tx_kernel<512>, F, 64, lds, st, a); break;
        case 1024: e = launch(tx_kernel<1024>, F, 128, lds, st, a); break;
        case 4096: e = launch(tx_kernel<4096>, F, 512, lds, st, a); break;
#endif
        default:   e = launch(tx_kernel<2048>, F, 256, lds, st, a); break;
    }
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

static hipError_t run_corr(const gf3_ctx* c, const CorrPlan& pl, const CorrArgs& a, int64_t grid, hipStream_t st) {
    const int NCp = pl.NC;
    const size_t lds = (GF3_CORR_PP ? fft_lds_bytes(NCp) : (size_t)(NCp + NCp / 8) * sizeof(cplx)) + 32 * sizeof(double);
    hipError_t e = hipSuccess;
#ifdef GF3_DEV_BUILD
    if (NCp == 1024) {
        if (a.dt == DT_F64) return launch((corr_kernel<1024, DT_F64>), grid, 128, lds, st, a);
        return launch((corr_kernel<1024, DT_F32>), grid, 128, lds, st, a);
    }
#endif
    DISPATCH_NC(NCp, a.dt, e = launch((corr_kernel<NCC, DTC>), grid, NCC / 8, lds, st, a));
    return e;
}

extern "C" int gf3_sync_frames(gf3_ctx* c, const void* d_in, int64_t n_in, int64_t F, int64_t stride,
                               int32_t win_lo, int32_t win_hi, int64_t* d_starts, double* d_peak, void* stream) {
    DeviceGuard dg(c);
    if (c && F == 0) return GF3_OK;
    if (!c || !d_in || !d_starts || F < 0) return fail(c, GF3_EINVAL, "gf3_sync_frames: bad argument");
    const int W = win_hi - win_lo;
    const CorrPlan& pl = c->frames_plan;
    if (W < 3 || W > pl.W) return fail(c, GF3_EINVAL, "gf3_sync_frames: window %d outside [3, %d]", W, pl.W);
    CorrArgs a{};
    a.t = pl.t; a.in = d_in; a.n_in = n_in; a.dt = c->cfg.in_dtype;
    a.Hq = pl.d_Hq; a.Q = pl.Q; a.Lp = pl.Lp; a.Lc = c->Lc; a.Wmax = W;
    a.stride = stride; a.win_lo = win_lo; a.W = W; a.starts = d_starts; a.peak = d_peak; a.thresh = c->cfg.thresh;
    HIPCHK(c, run_corr(c, pl, a, F, (hipStream_t)stream));
    return GF3_OK;
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

// The screening pass: the band-limited ring kernel when the plan allows it (and `general` is not asked for), else
// the general kernel.  R, the ring kernel's blocks per workgroup: every workgroup transforms Q - 1 windows without
// finishing a block, so R is as large as leaves a whole number of rounds of 2 workgroups per CU (swept on the
// config-3 stream when it was 83 582 blocks and every block ran its inverse transform: R = 164 -> 510 workgroups
// 1.71 ms, R = 82 1.72, R = 32 2.0, and 2.2 at R = 110 = 1.5 rounds).
static hipError_t launch_screen(const gf3_ctx* c, ScreenArgs a, bool general, hipStream_t st) {
    const auto& sp = c->scr;
    hipError_t e = hipSuccess;
    if (sp.ring && !general) {
        const int64_t slots = 2 * (int64_t)c->n_cu;                                       // workgroups resident at once
        const int64_t rounds = (a.nblk + slots * 170 - 1) / (slots * 170);
        int64_t R = sp.R_forced > 0 ? sp.R_forced : (a.nblk + slots * rounds - 1) / (slots * rounds);
        if (R < 1) R = 1;                                                                 // (short streams: one block per workgroup, Q windows each, all at once)
        a.Hb = sp.d_Hb; a.ecoef = sp.d_ecoef; a.R = (int)R;
        const size_t lds = (size_t)2 * GF3_SCR_NC * sizeof(cf) + (64 + 64 + 8 + 2 + 2 * GF3_SCR_RQ + 2) * sizeof(float);
        const int64_t grid = (((a.nblk + R - 1) / R + 7) / 8) * 8;                        // padded to the 8 XCDs (xcd_order)
        DISPATCH_DT(a.dt, e = launch((scr_ring_kernel<DTC>), grid, GF3_SCR_T, lds, st, a));
    } else {
        const size_t lds = (size_t)2 * GF3_SCR_NC * sizeof(cf) + (128 + 11 * GF3_SCR_B + 4) * sizeof(float);
        const int64_t grid = (((a.nblk + GF3_SCR_B - 1) / GF3_SCR_B + 7) / 8) * 8;
        DISPATCH_DT(a.dt, e = launch((scr_ols_kernel<DTC>), grid, GF3_SCR_T, lds, st, a));
    }
    return e;
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
        const int64_t slots = 2 * (int64_t)c->n_cu;                         // (the LDS staging allows two workgroups per CU)
        const int64_t wgs = (w.s_cap + 3) / 4;                               // (a wave per cell at a time)
        const unsigned grid = (unsigned)(wgs < slots ? wgs : slots);
#if GF3_REFINE_MFMA
        const int64_t wg4 = 4 * slots;                                       // (no LDS staging: more resident waves, a wave per cell)
        DISPATCH_DT(dt, hipLaunchKernelGGL((scr_refine_mfma_kernel<DTC>), dim3((unsigned)(wg4 < wgs ? wg4 : wgs)), dim3(SCR_REF_THREADS), 0, st, a));
        (void)grid;
#else
        DISPATCH_DT(dt, hipLaunchKernelGGL((scr_refine_kernel<DTC>), dim3(grid), dim3(SCR_REF_THREADS), 0, st, a));
#endif
        HIPCHK(c, hipGetLastError());
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
        const size_t lds = fft_lds_bytes(pl.NC);
        hipError_t e = hipSuccess;
        auto pad8 = [](int64_t x) { return (x + 7) / 8 * 8; };              // grids padded to the 8 XCDs (xcd_order)
        a.nitems = w.nwin;
        DISPATCH_NC(pl.NC, a.dt, e = launch((spec_kernel<NCC, DTC>), pad8(w.nwin), NCC / 8, lds, st, a));
        HIPCHK(c, e);
        a.nitems = (w.nblk + OLS_B - 1) / OLS_B;
        switch (pl.NC) {
#ifndef GF3_DEV_BUILD
            case 512:  e = launch(ols_kernel<512>, pad8(a.nitems), 64, lds, st, a); break;
            case 1024: e = launch(ols_kernel<1024>, pad8(a.nitems), 128, lds, st, a); break;
            case 4096: e = launch(ols_kernel<4096>, pad8(a.nitems), 512, lds, st, a); break;
#endif
            default:   e = launch(ols_kernel<2048>, pad8(a.nitems), 256, lds, st, a); break;
        }
        HIPCHK(c, e);
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
    if (!c || !d_buf || !d_run_max || !d_idx || !d_val3 || !n_listed || !d_work || n < 3 || cap < 0)
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
        const size_t lds = fft_lds_bytes(pl.NC);
        hipError_t e = hipSuccess;
        auto pad8 = [](int64_t x) { return (x + 7) / 8 * 8; };
        a.nitems = w.nwin;
        DISPATCH_NC(pl.NC, a.dt, e = launch((spec_kernel<NCC, DTC>), pad8(w.nwin), NCC / 8, lds, st, a));
        HIPCHK(c, e);
        a.nitems = (w.nblk + OLS_B - 1) / OLS_B;
        switch (pl.NC) {
#ifndef GF3_DEV_BUILD
            case 512:  e = launch(ols_kernel<512>, pad8(a.nitems), 64, lds, st, a); break;
            case 1024: e = launch(ols_kernel<1024>, pad8(a.nitems), 128, lds, st, a); break;
            case 4096: e = launch(ols_kernel<4096>, pad8(a.nitems), 512, lds, st, a); break;
#endif
            default:   e = launch(ols_kernel<2048>, pad8(a.nitems), 256, lds, st, a); break;
        }
        HIPCHK(c, e);
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
