// The small bookkeeping kernels of the screened stream sync (gf3rx_screen.h): lower bound of the maximum, the work
// list of cells, the reference's rule on the re-evaluated fp64 values.  Non-template kernels: included by exactly one
// translation unit (gf3rx_sync.hip).
#pragma once
#include "gf3rx_screen_defs.h"

// One list serves both questions.  With Mlo <= M:  a lag that could be the maximum has an upper bound >= Mlo, a lag
// that could pass the threshold has one >= thresh M (1 - 1e-6) >= thresh Mlo (1 - 1e-6); so every lag that matters has
// an upper bound >= lim = min(Mlo, thresh Mlo (1 - 1e-6)), known BEFORE any fp64 value is.  The cells under those lags
// are re-evaluated once; M is the largest of their fp64 values and the rule is then applied to the same values.
// (A few dozen workgroups: one workgroup's loop over 80 000 blocks was 40 us of load latency.  The last one to
//  contribute turns the key into Mlo and lim.)
__global__ __launch_bounds__(SCR_MLO_THREADS) void scr_mlo_kernel(const float* blk_max, const float* blk_err, int64_t nblk, ScrMisc* misc, double thresh) {
    __shared__ double scratch[16];
    double m = -INFINITY;                                                     // (fmax drops NaN blocks here; their lags are kept by scr_flag_kernel)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nblk; i += (int64_t)gridDim.x * blockDim.x)
        m = fmax(m, (double)blk_max[i] - (double)blk_err[i]);
    m = block_max(m, scratch);
    if (threadIdx.x == 0) {
        if (m > -INFINITY) atomicMax(&misc->mlo_key, scr_key(m));
        __threadfence();
        if (atomicAdd(&misc->mlo_done, 1u) == gridDim.x - 1) {
            __threadfence();
            m = scr_unkey(atomicMax(&misc->mlo_key, 0ull));
            misc->Mlo = m;
            // (the prefilter of pk_candidates: only a positive finite maximum and threshold exclude anything)
            const bool filt = m > 0.0 && thresh > 0.0 && m < INFINITY && thresh < INFINITY;
            misc->lim = filt ? fmin(m, thresh * m * (1.0 - 1e-6)) : -INFINITY;
        }
    }
}

// Cells: cell c = centre lags m = 1 + 14 c .. 14 + 14 c of the full correlation (zeros-indices i = m - 1); its
// refinement evaluates the 16 lags 14 c .. 14 c + 15.  A cell is listed when one of its lags (centres, plus lag 0
// for cell 0 and the last lag for the last cell) has an upper bound P32 + E_b that reaches misc->lim.
// scr_flag_kernel looks at the lags once: a workgroup covers 64 segments of 64 cells, leaves one 64-bit hit mask per
// segment and the number of hits; after the scan of those numbers scr_scatter_kernel turns the masks into the ascending
// list of cell numbers.  The four waves of a flag workgroup take every fourth segment and never wait for one another
// inside the loop (a stream with a chirp every 78 000 samples has one or two active blocks, four to eight active
// segments, under a workgroup).
__global__ __launch_bounds__(SCR_LIST_THREADS) void scr_flag_kernel(const float* __restrict__ P32, const float* __restrict__ blk_max,
                                                                    const float* __restrict__ blk_err, int H, int64_t plen, int64_t ncell,
                                                                    const ScrMisc* misc, unsigned long long* masks, int64_t* counts) {
    __shared__ int wsum[SCR_LIST_THREADS / 64];
    __shared__ unsigned long long blkmask;
    const double level = misc->lim;
    const bool all = !(level == level);                                      // NaN level: nothing can be excluded
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Which of the (at most 64) output blocks under this workgroup's lags can reach the level at all: one parallel
    // look, shared through LDS.  Most lags of a stream lie under blocks that cannot.
    // (first block by a double multiplication and a correction: a 64-bit division is a hundred instructions, and most
    //  workgroups do nothing else)
    const int64_t L0 = GF3_SCR_CELL * (int64_t)blockIdx.x * SCR_LIST_SEGS * 64;       // < 2^53: exact as a double
    int64_t b_first = (int64_t)((double)L0 * (1.0 / (double)H));
    if (b_first * (int64_t)H > L0) --b_first;
    if ((b_first + 1) * (int64_t)H <= L0) ++b_first;
    if (wave == 0) {
        const int64_t bb = b_first + lane;
        const bool act = bb * (int64_t)H < plen && ((double)blk_max[bb] + (double)blk_err[bb] >= level);
        const unsigned long long m = __ballot(act);
        if (lane == 0) blkmask = all ? ~0ull : m;
    }
    __syncthreads();
    const unsigned long long bm = blkmask;
    if (bm == 0) { if (threadIdx.x == 0) counts[blockIdx.x] = 0; return; }   // (the masks are not read when the count is 0)
    // Which of this wave's 16 segments touch a block that can reach the level: lane i answers for segment wave + 4 i
    // (the blocks under its lags [14 c0, 14 (c0 + 64) + 2), 32-bit arithmetic relative to the workgroup's first block),
    // then the wave walks the set bits only.
    unsigned todo;
    {
        const int64_t c0 = ((int64_t)blockIdx.x * SCR_LIST_SEGS + wave + 4 * (lane & 15)) * 64;
        const unsigned rel0 = (unsigned)(GF3_SCR_CELL * c0 - b_first * (int64_t)H);
        const int r0 = (int)(rel0 / (unsigned)H), r1 = (int)((rel0 + GF3_SCR_CELL * 64 + 1) / (unsigned)H);
        const unsigned long long span = (r1 >= 63 ? ~0ull : ((1ull << (r1 + 1)) - 1ull)) & ~((1ull << r0) - 1ull);
        todo = (unsigned)(__ballot(lane < 16 && c0 < ncell && (bm & span) != 0) & 0xffffull);
    }
    unsigned long long keep = 0;                                             // lane i: mask of segment wave + 4 i
    while (todo) {                                                           // (uniform over the wave)
        const int si = __ffs((int)todo) - 1;
        todo &= todo - 1;
        const int64_t c0 = ((int64_t)blockIdx.x * SCR_LIST_SEGS + wave + 4 * si) * 64;
        const int64_t c = c0 + lane;
        bool hit = false;
        if (c < ncell) {
            int64_t lo = GF3_SCR_CELL * c + 1, hi = lo + GF3_SCR_CELL;       // centres [lo, hi)
            if (c == 0) lo = 0;
            if (c == ncell - 1) hi = plen;
            if (hi > plen) hi = plen;
            // block of the first lag by one 32-bit division relative to the workgroup's first block; at most one
            // boundary can fall inside a cell (H >= 1024 > 16)
            const unsigned rel = (unsigned)(lo - b_first * (int64_t)H);
            int64_t bb = b_first + rel / (unsigned)H;
            int64_t bend = (bb + 1) * (int64_t)H;
            const float e0 = blk_err[bb];
            const float e1 = (bend < hi) ? blk_err[bb + 1] : e0;
            // a lag counts only if its own block can reach the level: blocks that cannot were possibly never written
            const bool a0 = all | ((double)blk_max[bb] + (double)e0 >= level);
            const bool a1 = (bend < hi) ? (all | ((double)blk_max[bb + 1] + (double)e1 >= level)) : a0;
            // all (at most 16) lags are fetched before any is looked at -- a short-circuiting loop would serialise
            // sixteen HBM round trips -- and fetched whether or not their block was written (what an unwritten block
            // holds is ignored below), so that these loads do not wait for the block bounds above either
            float pv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) pv[j] = P32[lo + j < hi ? lo + j : hi - 1];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double up = (double)pv[j] + (double)((lo + j >= bend) ? e1 : e0);
                hit = hit | ((lo + j < hi) & ((lo + j >= bend) ? a1 : a0) & ((up >= level) | !(up == up)));  // (a NaN keeps the lag)
            }
            hit = hit || all;
        }
        const unsigned long long bal = __ballot(hit);
        if (lane == si) keep = bal;
    }
    if (lane < SCR_LIST_SEGS / 4) masks[(int64_t)blockIdx.x * SCR_LIST_SEGS + wave + 4 * lane] = keep;
    int n = __popcll(keep);
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) n += __shfl_xor(n, d, 64);              // (lanes 16 .. 63 hold 0)
    if (lane == 0) wsum[wave] = n;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = (int64_t)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// One wave per flag workgroup: segment masks + the workgroup's offset -> cell numbers in ascending order.  Workgroup 0
// also publishes the list length and raises the overflow flag.
__global__ __launch_bounds__(64) void scr_scatter_kernel(const unsigned long long* masks, const int64_t* counts, const int64_t* offsets,
                                                         const int64_t* total, ScrMisc* misc, int64_t* cells, int64_t cap) {
    const long long n = total[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) { misc->ncell = n; if (n > cap) misc->status |= 1; }
    if (n > cap || counts[blockIdx.x] == 0) return;
    const int lane = threadIdx.x;
    unsigned long long m = masks[(int64_t)blockIdx.x * SCR_LIST_SEGS + lane];
    const int pc = __popcll(m);
    int x = pc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    int64_t o = offsets[blockIdx.x] + (x - pc);
    const int64_t c0 = ((int64_t)blockIdx.x * SCR_LIST_SEGS + lane) * 64;
    while (m) {
        const int j = __ffsll((long long)m) - 1;
        cells[o++] = c0 + j;                                                 // (o < n <= cap)
        m &= m - 1;
    }
}

// candidates of every listed cell: the reference's rule on the fp64 values, division by the maximum first
// (OFDM.py:359-361).  One thread per cell; bit j of the mask: zeros-index 14 c + j is a candidate.
__global__ void scr_decide_kernel(const int64_t* cells, const double* cell_val, ScrMisc* misc, int64_t nz, double thresh,
                                  unsigned* cell_mask, int64_t* cell_cnt) {
    if (misc->status & 1) return;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const double M = misc->m_nan ? NAN : scr_unkey(misc->m_key);          // (np.amax propagates NaN)
    if (i == 0) misc->M = M;
    if (i >= misc->ncell) return;
    const int64_t m0 = GF3_SCR_CELL * cells[i];
    double p[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) p[j] = cell_val[i * 16 + j] / M;
    unsigned mk = 0;
#pragma unroll
    for (int j = 0; j < GF3_SCR_CELL; ++j) {
        const bool cand = (m0 + j < nz) && ((p[j + 1] - p[j]) * (p[j + 2] - p[j + 1]) <= 0.0) && (p[j + 1] > thresh);
        mk |= cand ? (1u << j) : 0u;
    }
    cell_mask[i] = mk;
    cell_cnt[i] = __popc(mk);
    if (mk) atomicAdd((unsigned long long*)&misc->nhit, 1ull);
}

// ordered expansion of the cell masks into zeros-indices
__global__ void scr_expand_kernel(const int64_t* cells, const unsigned* cell_mask, const int64_t* offsets, const ScrMisc* misc,
                                  int64_t* cand, int64_t cap) {
    if (misc->status & 1) return;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= misc->ncell) return;
    unsigned m = cell_mask[i];
    int64_t o = offsets[i];
    const int64_t base = GF3_SCR_CELL * cells[i];
    while (m) {
        const int j = __ffs((int)m) - 1;
        if (o < cap) cand[o] = base + j;
        ++o;
        m &= m - 1;
    }
}

