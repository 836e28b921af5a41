// libgf3rx -- the two-phase demodulation of LONG packets, FEW at a time (the reference's own geometry: no_pilots = 20,
// packet_length = 180, OFDM.py:18, three packets in its published test, Final System Test.ipynb cell 7).
//
// demod_kernel gives a packet to ONE workgroup, which is what a batch of thousands wants; three packets of 220 symbols
// are three workgroups on a 256-CU chip, each walking 182 transforms one after the other.  But the triple loop this
// replaces (OFDM.py:466-478) is independent over (symbol, carrier) once Hs, He and the phase slope exist, and those
// three are outputs of the path anyway (Hest_start, Hest_end, OFDM.py:443-462).  So, for such calls, three launches:
//   1. pilot_sum_kernel   grid F x 2 x N/512: the P pilot symbols of each side summed in the time domain, sample by
//                          sample, in the one-launch kernel's order of additions (bit-identical sums); every thread has
//                          its P loads in flight at once;
//   2. demod_kernel<.., STAGE_EST>   grid F: the two transforms, Hs = mean / known, He, the slope fit -> Hs, He, slope
//                          in memory (the caller's dump arrays, or the workspace);
//   3. demod_kernel<.., STAGE_DATA>  grid F x ceil(D / Dc): each workgroup re-derives the per-carrier state from those
//                          arrays (the same doubles -> the same u, a0, da), starts its phasors at symbol l0 from the
//                          two-level rotation table, u exp(j slope n f_l0), and demodulates Dc symbols into ITS bit range:
//                          Dc is a multiple of 32 / gcd(C mu, 32), so chunks meet on word boundaries of the packed row.
// What differs from the one-launch path is the rounding of the phasor at a chunk's first symbol (computed directly
// there, reached by l0 recurrence steps here: <= 1e-13 relative); bits are the same off exact decision ties.
#include "gf3rx_demod.h"

struct PilotSumArgs {
    const void* in; int64_t n_in; const int64_t* off;
    int CP, S, P, D, NC;
    double* psum;             // [F][2][2 NC]
};

template <int DT>
__global__ __launch_bounds__(256) void pilot_sum_kernel(PilotSumArgs a) {
    const int per = a.NC / 256;                                     // workgroups per (packet, side): a sample pair per thread
    const unsigned blk = blockIdx.x;
    const int64_t f = blk / (unsigned)(2 * per);
    const int rem = (int)(blk - (unsigned)f * (unsigned)(2 * per));
    const int side = rem / per;
    const int j = (rem - side * per) * 256 + threadIdx.x;           // pair index inside the symbol
    const int64_t off = a.off[f];
    const bool ok = off >= 0 && off + (int64_t)(2 * a.P + a.D) * a.S <= a.n_in;
    cplx sum = cmk(0.0, 0.0);
    if (ok) {
        const int64_t base = off + (int64_t)(side ? a.P + a.D : 0) * a.S + a.CP + 2 * j;
        int p = 0;
        for (; p + 8 <= a.P; p += 8) {                               // eight loads in flight, added in order
            RawPair<DT> raw[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) raw[k].load(a.in, base + (int64_t)(p + k) * a.S);
#pragma unroll
            for (int k = 0; k < 8; ++k) sum = cadd(sum, raw[k].get());
        }
        for (; p < a.P; ++p) { RawPair<DT> raw; raw.load(a.in, base + (int64_t)p * a.S); sum = cadd(sum, raw.get()); }
    }
    ((cplx*)a.psum)[((int64_t)f * 2 + side) * a.NC + j] = sum;
}

extern "C" int64_t gf3_demod_workspace_bytes(const gf3_ctx* c, int64_t F) {
    if (!c || F < 0) return 0;
    // [F][2][N] pilot sums | Hs [F][K] | He [F][K] | slope [F]   (the last three only when the caller keeps no dumps)
    return (int64_t)((size_t)F * ((size_t)4 * c->NC * sizeof(double) + (size_t)2 * c->K * sizeof(cplx) + sizeof(double)) + 256);
}

// Chunk length of the data stage: a multiple of q = 32 / gcd(C mu, 32) -- chunks then meet on word boundaries of the
// packed row -- near D F / (2 CUs), so that the launch fills the chip about once, and not below DC_MIN symbols (a
// workgroup's start-up -- state, rotation tables -- costs about one transform).
static void split_geometry(const gf3_ctx* c, int64_t F, int& Dc, int& nchunk) {
    const int Bs = c->cfg.C * c->cfg.mu, D = c->cfg.D;
    int g = 32, b = Bs;
    while (b) { const int t = g % b; g = b; b = t; }                 // gcd(32, Bs)
    const int q = 32 / g;
    constexpr int DC_MIN = 2;
    int64_t want = ((int64_t)D * F + 2 * c->n_cu - 1) / (2 * (int64_t)c->n_cu);
    if (want < DC_MIN) want = DC_MIN;
    Dc = (int)((want + q - 1) / q) * q;
    if (Dc > D) Dc = D;
    nchunk = (D + Dc - 1) / Dc;
}

// mode 0: by geometry; 1: always the one-launch kernel; 2: the two-phase form whenever a workspace is there
bool demod_wants_split(const gf3_ctx* c, int64_t F, int mode) {
    if (mode == 1) return false;
    int Dc, nchunk;
    split_geometry(c, F, Dc, nchunk);
    if (mode == 2) return true;
    // Measured over F = 1 ... 1024 on four geometries (tools/ab/time_split.py): wherever a packet cuts into two chunks or more
    // and one packet per workgroup cannot fill the chip's 2 x CUs slots, the two-phase form wins -- A2: 0.03 vs 0.49 ms at F = 3,
    // 0.22 vs 0.51 at F = 128, 0.42 vs 0.56 at F = 256; with a single chunk per packet it only adds two launches.
    return nchunk >= 2 && F <= 2 * (int64_t)c->n_cu;
}

extern "C" int gf3_demod_split_plan(const gf3_ctx* c, int64_t F, int32_t mode, int32_t* h_Dc, int32_t* h_nchunk) {
    if (!c || F < 0 || mode < 0 || mode > 2) return 0;
    int Dc = 0, nchunk = 0;
    split_geometry(c, F, Dc, nchunk);
    if (h_Dc) *h_Dc = Dc;
    if (h_nchunk) *h_nchunk = nchunk;
    return F > 0 && demod_wants_split(c, F, mode) ? 1 : 0;
}

int demod_split(const gf3_ctx* c, DemodArgs a, int64_t F, void* d_work, hipStream_t st) {
    const int NC = c->NC, K = c->K;
    char* w = (char*)d_work;
    double* psum = (double*)w;
    w += (size_t)F * 4 * NC * sizeof(double);
    if (!a.Hs) a.Hs = (cplx*)w;
    w += (size_t)F * K * sizeof(cplx);
    if (!a.He) a.He = (cplx*)w;
    w += (size_t)F * K * sizeof(cplx);
    if (!a.slope) a.slope = (double*)w;
    a.psum = psum;
    split_geometry(c, F, a.Dc, a.nchunk);
    {
        PilotSumArgs pa{a.in, a.n_in, a.off, a.CP, a.S, a.P, a.D, NC, psum};
        const int64_t grid = F * 2 * (NC / 256);
        DISPATCH_DT(a.dt, hipLaunchKernelGGL((pilot_sum_kernel<DTC>), dim3((unsigned)grid), dim3(256), 0, st, pa));
        HIPCHK(c, hipGetLastError());
    }
    {
        hipError_t e = hipSuccess;
        const size_t lds = demod_lds_bytes(c, true);
        switch (NC) {
#ifndef GF3_DEV_BUILD
            case 512:  e = launch((demod_kernel<512, DT_F64, false, MODE_QPSK, STAGE_EST>), F, 64, lds, st, a); break;
            case 1024: e = launch((demod_kernel<1024, DT_F64, false, MODE_QPSK, STAGE_EST>), F, 128, lds, st, a); break;
            case 4096: e = launch((demod_kernel<4096, DT_F64, false, MODE_QPSK, STAGE_EST>), F, 512, lds, st, a); break;
#endif
            default:   e = launch((demod_kernel<2048, DT_F64, false, MODE_QPSK, STAGE_EST>), F, 256, lds, st, a); break;
        }
        HIPCHK(c, e);
    }
    hipError_t e = hipSuccess;
    if (a.eq || a.Hest) e = launch_dsplit_full(c, a, F * a.nchunk, st);
    else if (c->qpsk_q > 0.0) e = launch_dsplit_qpsk(c, a, F * a.nchunk, st);
    else e = launch_dsplit_scan(c, a, F * a.nchunk, st);
    HIPCHK(c, e);
    return GF3_OK;
}
