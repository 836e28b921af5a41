// demod_kernel: the fused demodulation of one packet per workgroup, and the decision helpers it shares with the
// stand-alone demappers.  A header because its three modes are compiled in three translation units
// (gf3rx_demod_qpsk.hip, gf3rx_demod_scan.hip, gf3rx_demod_full.hip) and the two-phase kernels of long packets
// (gf3rx_demod_split.h) reuse its pieces.
#pragma once
#include "gf3rx_host.h"

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
// two axes' `off` in one comparison: clear <=> max(|offI|, |offQ|) < 1/2 - 1e-9.  (A NaN on BOTH axes compares false;
// fmax drops a NaN that sits on one axis only -- which cannot happen here: tp comes from the two parts of one complex
// product ep = X conj(g), and a non-finite X or g makes both parts non-finite.)
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
    // N = 8192 (NC = 4096, 512 threads): the register file allows one workgroup per CU whatever the LDS does, and two
    // 64 KB buffers fit beside the decision ring (148 KB of 160) -- one barrier per exchange instead of two, with no other
    // workgroup on the CU to run under a barrier.  Four passes end in the second buffer and the next transform starts in
    // the first, so the buffers need not be flipped (rfft_regs passes flip = 0 for the unfused sizes).
    static constexpr bool PP_SIZE = FftGeom<NC>::PINGPONG || NC == 4096;
#if GF3_ABL >= 2
    static constexpr bool PP = PP_SIZE && WPS <= 2;
#else
    static constexpr bool PP = PP_SIZE && WPS <= 2 && !MAG_LDS;
#endif
    static constexpr int LDS_ELEMS = PP ? 2 * NC : FftGeom<NC>::LDS_ELEMS_INPLACE;
    static constexpr int MAG_ELEMS = MAG_LDS ? NC : 0;             // double2 (a0, da) per slot per thread: 8 * NC/8
};

// STAGE: the whole packet in one workgroup (STAGE_ALL, the batch path), or -- for long packets, few at a time -- its two
// halves as separate launches (gf3rx_demod_split.hip): STAGE_EST stops after the channel estimate (input: the time-domain
// pilot sums of pilot_sum_kernel; output: Hs, He, slope), STAGE_DATA takes that state from memory and demodulates the
// data symbols [chunk Dc, (chunk + 1) Dc) of packet blockIdx.x / nchunk.  The discarded branches vanish: STAGE_ALL is
// the kernel it was.
enum { STAGE_ALL = 0, STAGE_EST = 1, STAGE_DATA = 2 };
template <int NC, int DT, bool SPECTRA, int MODE, int STAGE = STAGE_ALL>
__global__ __launch_bounds__(NC / 8, (DemodOcc<NC, MODE>::WPS)) void demod_kernel(DemodArgs a) {
    constexpr bool FULL = (MODE == MODE_FULL);
    static_assert(STAGE == STAGE_ALL || !SPECTRA, "the two-phase form takes time-domain input");
    static_assert(STAGE != STAGE_EST || DT == DT_F64, "the estimate stage reads the fp64 pilot sums");
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
    int64_t f = blockIdx.x;
    const int K = a.K, P = a.P, D = a.D, S = a.S;
    int l_lo = 0, l_hi = D;                                               // data symbols this workgroup demodulates
    if constexpr (STAGE == STAGE_DATA) {
        const unsigned pk = blockIdx.x / (unsigned)a.nchunk;
        f = pk;
        l_lo = __builtin_amdgcn_readfirstlane((int)(blockIdx.x - pk * (unsigned)a.nchunk) * a.Dc);     // (wave-uniform: kept in scalar registers)
        l_hi = __builtin_amdgcn_readfirstlane(min(D, l_lo + a.Dc));
    }
    const int Bs = a.C * a.mu;                                            // bits per data symbol
    uint8_t* row = a.bits + f * (int64_t)a.row_bytes;

    int64_t off = 0;
    if constexpr (!SPECTRA) {
        off = a.off[f];
        const bool ok = off >= 0 && off + (int64_t)(2 * P + D) * S <= a.n_in;
        if (!ok) {                                                        // ragged packet
            if constexpr (STAGE == STAGE_DATA) return;                    // (the estimate stage has zeroed the row)
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
        if constexpr (STAGE == STAGE_EST) base = (const E*)a.psum + ((int64_t)f * 2 + i) * (2 * NC);   // "symbol" i = side i's pilot sum
        const unsigned t2 = 2u * (unsigned)launder(tid);
#pragma unroll
        for (int r = 0; r < 8; ++r) nxt[r].load_u(base, t2 + 2u * (unsigned)(r * T));
    };
    const int Msym = STAGE == STAGE_EST ? 2 : (STAGE == STAGE_DATA ? 2 * P + l_hi : 2 * P + D);      // (nothing is fetched from here on)
    const int Pl = STAGE == STAGE_EST ? 1 : P;                            // symbols summed per side
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
    if constexpr (!SPECTRA) fetch(STAGE == STAGE_DATA ? 2 * P + l_lo : 0);

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
    } else if constexpr (STAGE == STAGE_DATA) {
        // the estimate stage's Hs, He (true scale, divided by the known symbols): the same doubles the one-launch kernel
        // holds in registers at this point, so u, a0, da below come out bit for bit the same.  (He only feeds the
        // magnitude model: the sign mode never looks at it.)
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
            Hs[s2] = a.Hs[f * K + bin_of(s2) - 1];
            He[s2] = cmk(0.0, 0.0);                   // (the table modes fetch He slot by slot below, next to its one use)
        }
    } else {
        for (int side = 0; side < 2; ++side) {
            cplx sum[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) sum[r] = cmk(0.0, 0.0);
            for (int i = side * Pl; i < (side + 1) * Pl; ++i) {
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
    if constexpr (STAGE != STAGE_DATA) lds_barrier();  // FFT buffer is free: reuse it for the fit-range carriers
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
    if constexpr (STAGE != STAGE_DATA) {
#pragma unroll
        for (int s = 0; s < 8; ++s) ik[s] = a.inv_known[bin_of(s) - 1];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if constexpr (STAGE != STAGE_DATA) {
            Hs[s] = cmul(cscale(Hs[s], invP), ik[s]);
            He[s] = cmul(cscale(He[s], invP), ik[s]);
        }
        const double m2 = Hs[s].x * Hs[s].x + Hs[s].y * Hs[s].y;
        const double ia = rsq_nr(m2);                                 // 1/|Hs|
        if constexpr (MODE != MODE_QPSK) {
            if constexpr (STAGE == STAGE_DATA) He[s] = a.He[f * K + bin_of(s) - 1];
            const double e2 = He[s].x * He[s].x + He[s].y * He[s].y;
            const double ah = m2 * ia;                                // |Hs|
            a0[s] = XS * ah;
            da[s] = XS * (e2 * rsq_nr(e2) - ah);                      // XS (|He| - |Hs|)
        }
        u[s] = cmk(Hs[s].x * ia, Hs[s].y * ia);
    }
    // (b) optional dumps; the carriers inside the fit range go to LDS, where the angles are taken by
    //     whichever thread the carrier falls to (2 L angles per packet instead of 16 per thread)
    double slope;
    if constexpr (STAGE == STAGE_DATA) slope = a.slope[f];
    else {
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
    }
    if constexpr (STAGE == STAGE_EST) return;         // Hs, He, slope are in memory: the data stage takes it from there
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
        // (the data stage of the two-phase form starts its phasors at its first symbol: f_l at l = l_lo)
        const double phi0 = STAGE == STAGE_DATA ? slope * (((double)l_lo + 0.5 * (double)P) / denom) : slope * ((0.5 * (double)P) / denom);
        const double dphi = slope / denom;
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
    for (int l = l_lo; l < l_hi; ++l) {
        if constexpr (SPECTRA) { lds_barrier(); load_spectra(a.sp_data + ((int64_t)f * D + l) * K); }
        else transform(2 * P + l);
        if (l > l_lo) pack_words(l - 1, false);
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
    pack_words(l_hi - 1, l_hi == D && ((D * Bs) & 31) != 0);
    GF3_STAMP(5);
    GF3_STAMP_RT(7);
}

// ============================================================================
// host side: LDS layout of demod_kernel
// ============================================================================
// symbols in the decision-byte ring of demod_kernel: ceil(32 / (C mu)) + 2, rounded up to a power of two
// (>= 4: the QPSK packer reads the ring one aligned dword at a time, so ring * C must be a multiple of 4)
inline int demod_ring(const gf3_ctx* c) {
    const int Bs = c->cfg.C * c->cfg.mu;
    const int need = (32 + Bs - 1) / Bs + 2;
    int r = 4;
    while (r < need) r <<= 1;
    return r;
}
// lean = MODE_QPSK (ping-pong FFT buffers); the table modes use the in-place buffer and NC (a0, da) pairs.
// Layout: [scratch 32 doubles | rotation tables | FFT buffer | decision bytes | (a0, da) pairs]; everything from the
// FFT buffer on is overlaid by Hs, He of the fit range during the channel-estimate stage (which may need more).
inline size_t demod_lds_bytes(const gf3_ctx* c, bool lean = false) {
    const bool inplace = !lean || (GF3_DEMOD_WPS > 2 && c->NC <= 2048);
    const bool pp_size = c->NC == 1024 || c->NC == 2048 || c->NC == 4096;        // == DemodOcc::PP_SIZE
    const size_t fft = (inplace || !pp_size) ? (size_t)(c->NC + c->NC / 8) * sizeof(cplx) : (size_t)2 * c->NC * sizeof(cplx);
    const size_t mags = lean ? 0 : (size_t)c->NC * sizeof(double2);
    const size_t tail = fft + (size_t)((demod_ring(c) * c->cfg.C + 15) & ~15) + mags;
    const size_t fit = (size_t)2 * (c->fit_hi - c->fit_lo) * sizeof(cplx);
    return 32 * sizeof(double) + (size_t)2 * (64 + c->NC / 64 + 1) * sizeof(cplx) + (fit > tail ? fit : tail);
}
