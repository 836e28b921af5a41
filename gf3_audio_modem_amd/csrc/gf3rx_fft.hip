// libgf3rx -- transforms on their own: the batched real FFT (remove_cp + np.fft.fft) and the transmit-side synthesiser.
#include "gf3rx_host.h"

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

hipError_t run_rfft_nc(int NCv, FftTables t, const void* d_in, int64_t n_in, int dt, const int64_t* d_off,
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

int tx_launch(gf3_ctx* c, const TxArgs& a, int64_t F, hipStream_t st) {
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
