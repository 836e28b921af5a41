// Constants and argument blocks of the stream-mode screening pass (gf3rx_screen.h has the kernels and the method;
// gf3rx_screen_list.h the small list kernels).  Split out so that the host-side translation units (plan building,
// workspace layout) see the sizes without compiling the kernels.
#pragma once
#include "gf3rx_device.h"

typedef float2 cf;
#define GF3_SCR_NC 4096              /* complex points of the screening transform (8192 real samples per window) */
#define GF3_SCR_T 256                /* threads per workgroup: 16 points each */
#define GF3_SCR_B 4                  /* adjacent output blocks per workgroup (share their windows' transforms) */
#define GF3_SCR_CELL 14              /* centre lags per refinement cell (16 fp64 values with the two neighbours) */
#define GF3_SCR_GAMMA (256.0f * 5.9604645e-8f)
// Samples below 1e-19 have squares that underflow in fp32: a sum of 8192 rounded squares can miss 8192 x 1.4e-45 of
// the true energy.  Added under the root, the energy stays an UPPER bound of |x|_2^2 whatever the samples' size (a stream
// that small then has bounds far above its own correlation, lists everything and takes the fp64 path).
#define GF3_SCR_UFLOW 2e-41f
#define GF3_SCR_KS 6                 /* slots t + 256 r, r < KS, of the half spectrum are kept (even: read in pairs) */
#define GF3_SCR_RQ 8                 /* ring depth = largest Q */

struct ScreenArgs {
    const void* in; int64_t n_in; int dt;
    const cf* tw;              // [4096] exp(-2 pi i m / 4096)
    const cf* twn;             // [2049] exp(-2 pi i k / 8192)
    const float4* Hs;          // [Q][8][256]: (H_q[k], H_q[4096 - k]), k = t + 256 r  (thread 0, r = 0: bin 2048 twice)
    const float* H0N;          // [Q][2]: H_q[0], H_q[4096] (real)
    const float* Hinf;         // [Q] max_k |H_q[k]|, rounded up
    int Q, H, Lc;
    int64_t nblk, plen;
    float* P32;                // [plen]
    float* blk_max;            // [nblk] max of the block's P32
    float* blk_err;            // [nblk] bound on |P32 - P| for every lag of the block
    int* run_lo;               // optional: running lower bound of the maximum (float bits, > 0), shared by the grid
    float thresh;              // 0 < thresh < 1 enables skipping the store of blocks that cannot matter
    // band-limited kernel (scr_ring_kernel) only:
    const float4* Hb;          // [Q][GF3_SCR_KS / 2][256]: (H_q[k], H_q[k + 256]), k = t + 512 p  -- the bins below 256 GF3_SCR_KS
    const float* ecoef;        // [2][Q] error per unit |x|_2: GF3_SCR_GAMMA (max|H_q| + |h_q,out|_2); per unit |x_out|_2: |h_q,out|_2
    int R;                     // output blocks per workgroup
    unsigned long long* bad;   // optional: bit 0 is set when a window's energy is not finite in fp32 (NaN / Inf samples, or
                               // finite ones beyond 1e19): the bounds mean nothing then and the caller takes the fp64 path
};

// ---------------------------------------------------------------- screening bookkeeping
struct ScrMisc {                  // device-resident scalars of one gf3_sync_stream call (zeroed by the host before the screen)
    double Mlo;                   // best lower bound of the maximum: max_b (blk_max - blk_err)
    double M;                     // the maximum (fp64 re-evaluation)
    double lim;                   // listing level: a lag whose upper bound stays below it can neither be the maximum nor pass the threshold
    long long ncell;              // cells listed (and re-evaluated)
    long long nhit;               // of those, cells that hold a candidate
    long long status;             // bit 0: the work list overflowed -> the caller falls back to the all-fp64 path
    unsigned long long mlo_key;   // running maximum of (blk_max - blk_err) as an ordered key (0: none yet)
    unsigned long long m_key;     // running maximum of the fp64 values, same encoding
    unsigned int mlo_done;        // workgroups of scr_mlo_kernel that have contributed
    unsigned int m_nan;           // a re-evaluated lag was NaN (np.amax then returns NaN)
    long long total;              // length of the list being scanned (cells, then candidates)
    long long np[2];              // pk_nms: peaks accepted, suppression status -- everything the host reads back is in this block
};
// order-preserving map double -> uint64 (every finite or infinite value maps above 0, so 0 can mean "nothing yet")
GF3_DEV unsigned long long scr_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
GF3_DEV double scr_unkey(unsigned long long k) {
    if (k == 0) return -INFINITY;
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}

#define SCR_MLO_THREADS 256
#define SCR_LIST_THREADS 256
#define SCR_LIST_SEGS 64             /* segments of 64 cells per workgroup: 4096 cells = 57 344 lags */

template <int DT> struct ScrStage { typedef float S; };      // what scr_refine_kernel stages a sample as
template <> struct ScrStage<DT_F64> { typedef double S; };
struct RefineArgs {
    const void* in; int64_t n_in; int dt;
    const double* chirp; int Lc;
    const int64_t* cells; ScrMisc* misc;
    int64_t plen;
    double* cell_val;             // [ncell][16] the cell's fp64 lags (their maximum goes to misc->m_key / m_nan)
    const double* chirp_t;        // the taps tiled for scr_refine_kernel: [step][q < 8][lane][2] = c[1024 step + 16 lane + 2 q + (0, 1)], 0 past Lc
    unsigned long long* stamps;   // diagnostic build only (-DGF3_STAMPS): [waves][8] s_memtime ticks summed per phase (tools/ab/refine_stamps.py)
};
#ifdef GF3_STAMPS
// s_memtime once everything the wave has in flight on the scalar / LDS side has returned; nothing is scheduled across it
#define SCR_TICK(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); \
                         __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SCR_TICK(t) do { } while (0)
#endif
#define SCR_REF_THREADS 256
#define SCR_REF_WT 1024                              /* taps per wave and step: 16 per lane */
