// Frames-mode chirp sync, screened: the search window of one packet evaluated in fp32 with a proven bound, the decision
// taken only where the bound decides it -- the frames counterpart of gf3rx_screen.h (opt-in: gf3_sync_frames_ex mode 1;
// the default and the bench's `value` stay on the all-fp64 corr_kernel).
//
// What the window rule of corr_kernel (the reference's peak rule applied to a window, OFDM.py:359-361) needs of the W lags
//   y[j] = sum_k r[s0 + j + k] c[k]:    M = max_j y[j];   first j in [1, W-2] with y[j]/M > thresh and
//                                         (y[j] - y[j-1]) (y[j+1] - y[j]) <= 0
// is an INDEX.  With fp32 lags y32 and a bound E >= |y32 - y| for the whole window (same analysis and the same constant as
// gf3rx_screen.h: E = GAMMA sum_q max_k|H_q[k]| |x_q|_2 over the partitions' segments), every lag is classified
//   NOT    y32 + E < thresh (max y32 - E)(1 - 1e-6)            -- cannot pass the threshold, whatever the rounding
//          or both differences beyond 2E with the SAME sign      -- cannot be an extremum
//   YES    y32 - E > thresh (max y32 + E)(1 + 1e-6) and both differences beyond 2E with OPPOSITE signs
//   MAYBE  everything else
// and the window is RESOLVED when no MAYBE precedes the first YES (or there is neither): the answer is then the one exact
// arithmetic gives -- and the one corr_kernel gives, whose own rounding (1e-16) is eleven orders inside the 2E margins.
// An unresolved window (noise near the threshold, a flat top, a window whose maximum the bound cannot separate from zero,
// non-finite samples) is appended to a list, and corr_kernel itself runs on the listed windows: no decision is ever taken on
// an fp32 value that the bound does not back.
//
// Geometry: 2048-sample transforms (1024 complex points) held by ONE wave -- 16 points per lane, passes 16 x 16 x 4, two
// exchanges through an 8 KB private LDS buffer and no barrier anywhere (LDS serves a wave's instructions in order); the fp64
// kernel cannot do this (16 fp64 points + 16 accumulators do not fit the register file), fp32 can.  Partition
// length Lp = 2048 - Wmax + 1, Q = ceil(Lc / Lp) forward transforms, the chirp partitions' spectra multiplied in and
// accumulated in registers, one inverse transform.
#pragma once
#include "gf3rx_screen.h"

#define GF3_FS_NC 1024               /* complex points per transform (2048 real samples) */

struct FScreenArgs {
    const void* in; int64_t n_in; int dt;
    const cf* tw;              // [1024] exp(-2 pi i m / 1024)
    const cf* twn;             // [64]   exp(-2 pi i t / 2048), t < 64
    const float4* Hs;          // [Q][8][64]: (H_q[k], H_q[1024 - k]), k = t + 64 r  (lane 0, r = 0: bin 512 twice)
    const float* H0N;          // [Q][2]: H_q[0], H_q[1024] (real)
    const float* Hinf;         // [Q] max_k |H_q[k]|, rounded up
    int Q, Lp, Lc, Wmax;
    int64_t stride; int win_lo, W;
    float thresh;
    int64_t* starts;           // [F] first-pilot sample index, -1: nothing found (as corr_kernel writes them)
    int* unresolved;           // [F] windows the bound could not decide
    int* n_unresolved;
    float* y32;                // optional (tests): [F][W] the fp32 lags
    float* err;                // optional (tests): [F] the window's bound
    int* cls;                  // optional (tests): [F] 0 resolved with a detection, 1 resolved without, 2 unresolved
};

// ---- packed fp32 complex arithmetic: a complex number is one 64-bit register pair and an addition, a subtraction, a
// multiplication by -i folded into the addition that follows, or half of a complex multiply is ONE v_pk_*_f32 with op_sel /
// neg operand modifiers.  Written as inline assembly: the compiler's own SLP packing of scalar code surrounds such
// instructions with register moves (which is why the library is built with -fno-slp-vectorize), and at two waves per SIMD a
// packed instruction retires 1.4 x (fma) / 1.1 x (add) the operations of two scalar ones (tools/ubench/pk_f32_rate.hip) at
// half the instruction count.
typedef float pf __attribute__((ext_vector_type(2)));
GF3_DEV pf pfmk(float a, float b) { pf r = {a, b}; return r; }
GF3_DEV pf pf_add(pf a, pf b) { return a + b; }
GF3_DEV pf pf_sub(pf a, pf b) { return a - b; }
GF3_DEV pf pf_add_negi(pf a, pf b) { pf r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }   // a - i b
GF3_DEV pf pf_sub_negi(pf a, pf b) { pf r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }   // a + i b
GF3_DEV pf pf_add_conj(pf a, pf b) { pf r; asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }                                  // a + conj(b)
GF3_DEV pf pf_sub_conj(pf a, pf b) { pf r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }                                  // a - conj(b)
GF3_DEV pf pf_cmul(pf a, pf w) {                                                                                                                                  // a w
    pf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
GF3_DEV pf pf_cmul_conj(pf a, pf w) {                                                                                                                             // a conj(w)
    pf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
GF3_DEV pf pf_cfma(pf a, pf w, pf c) {                                                                                                                            // c + a w
    pf t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "v"(w), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
GF3_DEV pf pf_cfma_conj(pf a, pf w, pf c) {                                                                                                                       // c + a conj(w)
    pf t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t) : "v"(a), "v"(w), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
GF3_DEV pf pf_scale(pf a, float s) { return a * s; }

// 4-point DFT in place: eight packed additions (the -i of the odd half rides on two of them)
GF3_DEV void pf_dft4(pf& a, pf& b, pf& c, pf& d) {
    const pf s0 = pf_add(a, c), s1 = pf_sub(a, c), s2 = pf_add(b, d), s3 = pf_sub(b, d);
    a = pf_add(s0, s2); c = pf_sub(s0, s2); b = pf_add_negi(s1, s3); d = pf_sub_negi(s1, s3);
}
// 16-point DFT in place as 4 x 4; output X[m] is left in v[scr_perm(m)] (the layout of scr_dft16)
GF3_DEV void pf_dft16(pf (&v)[16]) {
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
#pragma unroll
    for (int a = 0; a < 4; ++a) pf_dft4(v[a], v[a + 4], v[a + 8], v[a + 12]);      // v[a + 4b] = T_a[b]
    // T_a[b] *= W16^(a b)
    v[1 + 4] = pf_cmul(v[1 + 4], pfmk(c1, -s1));   v[1 + 8] = pf_scale(pf_add_negi(v[1 + 8], v[1 + 8]), h);     v[1 + 12] = pf_cmul(v[1 + 12], pfmk(s1, -c1));
    v[2 + 4] = pf_scale(pf_add_negi(v[2 + 4], v[2 + 4]), h);   v[2 + 8] = pfmk(v[2 + 8].y, -v[2 + 8].x);        v[2 + 12] = pf_scale(pf_sub_negi(v[2 + 12], v[2 + 12]), -h);
    v[3 + 4] = pf_cmul(v[3 + 4], pfmk(s1, -c1));   v[3 + 8] = pf_scale(pf_sub_negi(v[3 + 8], v[3 + 8]), -h);    v[3 + 12] = pf_cmul(v[3 + 12], pfmk(-c1, s1));
#pragma unroll
    for (int b = 0; b < 4; ++b) pf_dft4(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);   // v[c + 4b] = X[b + 4c]
}
// v[r] *= w^r, r = 1..15; the powers by the three-term recurrence w^(r+1) = 2 cos(theta) w^r - w^(r-1): one packed fma per
// power (error ~ r^2 u: inside the bound's constant, which allows 8 u per pass)
GF3_DEV void pf_twiddle16(pf (&v)[16], pf w) {
    const float c2 = w.x + w.x;
    pf pm = pfmk(1.0f, 0.0f), p = w;
    v[1] = pf_cmul(v[1], p);
#pragma unroll
    for (int r = 2; r < 16; ++r) {
        const pf n = __builtin_elementwise_fma(pfmk(c2, c2), p, -pm);
        pm = p; p = n;
        v[r] = pf_cmul(v[r], p);
    }
}

GF3_DEV int fs_swz(int i) { return i ^ ((i >> 4) & 15); }          // XOR swizzle of the first exchange (8-byte elements)

struct FsTw { pf tw2; pf tw3; };                                  // pass 2: exp(-2 pi i (t & 15) / 256); pass 3: exp(-2 pi i ((t & 15) + 64 (t >> 4)) / 1024)
#ifndef GF3_FS_PERMLANE
#define GF3_FS_PERMLANE 1     /* 1: the second exchange runs through v_permlane32_swap / v_permlane16_swap instead of LDS */
#endif
// swap the upper 32 lanes of a with the lower 32 lanes of b / the odd rows (of 16 lanes) of a with the even rows of b
GF3_DEV void pf_swap32(pf& a, pf& b) {
    auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = pfmk(__uint_as_float(rx[0]), __uint_as_float(ry[0])); b = pfmk(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
}
GF3_DEV void pf_swap16(pf& a, pf& b) {
    auto rx = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    auto ry = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = pfmk(__uint_as_float(rx[0]), __uint_as_float(ry[0])); b = pfmk(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
}
// Forward complex FFT of 1024 points in ONE wave: in v[r] = z[t + 64 r]; out (GF3_FS_PERMLANE) v[i + 4 m] =
// Z[(t & 15) + 16 i + 64 (t >> 4) + 256 m], i, m < 4 -- fs_bin(t, s) -- else Z[t + 64 s] in v[s].
// L: the wave's private 1024-point LDS buffer (in place: a wave's LDS instructions execute in order).
GF3_DEV int fs_bin(int t, int s) {
#if GF3_FS_PERMLANE
    return (t & 15) + 16 * (s & 3) + 64 * (t >> 4) + 256 * (s >> 2);
#else
    return t + 64 * s;
#endif
}
GF3_DEV void fs_fft1024(pf (&v)[16], pf* L, const FsTw& tw, int t) {
    pf_dft16(v);                                                   // X[m] in v[scr_perm(m)]
#pragma unroll
    for (int m = 0; m < 16; ++m) L[(t * 16 + m) ^ (t & 15)] = v[scr_perm(m)];        // logical t*16 + m, swizzled: (i >> 4) & 15 = t & 15
    {
        pf x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = L[fs_swz(t + 64 * r)];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = x[r];
    }
    pf_twiddle16(v, tw.tw2);                                       // w = exp(-2 pi i (t & 15) / 256)
    pf_dft16(v);
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
#if GF3_FS_PERMLANE
    // Second exchange WITHOUT LDS.  Lane (row rho = t >> 4, column k = t & 15) holds, as X[m] in v[scr_perm(m)], element
    // 256 rho + k + 16 m of the pass's output; the radix-4 butterflies of the last pass combine the four ROWS of one column.
    // A 4 x 4 transpose of chunks of four m across the rows -- two stages of cross-row swaps, 32 instructions -- leaves lane
    // (rho, k) with v[c + 4 i] = element 256 c + k + 16 (4 rho + i): the four inputs of butterfly j_i = k + 16 i + 64 rho.
    // (chunk c of a lane = m in [4c, 4c + 4) = registers v[c], v[c + 4], v[c + 8], v[c + 12].)
#pragma unroll
    for (int i = 0; i < 4; ++i) { pf_swap32(v[0 + 4 * i], v[2 + 4 * i]); pf_swap32(v[1 + 4 * i], v[3 + 4 * i]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { pf_swap16(v[0 + 4 * i], v[1 + 4 * i]); pf_swap16(v[2 + 4 * i], v[3 + 4 * i]); }
    {
        const float c64[4] = {1.0f, 0.99518472667219688624f, 0.98078528040323044913f, 0.95694033573220886494f};      // cos(2 pi i / 64)
        const float s64[4] = {0.0f, 0.09801714032956060199f, 0.19509032201612826785f, 0.29028467725446236764f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const pf w1 = i == 0 ? tw.tw3 : pf_cmul(tw.tw3, pfmk(c64[i], -s64[i]));   // exp(-2 pi i j_i / 1024)
            const pf w2 = pf_cmul(w1, w1), w3 = pf_cmul(w2, w1);
            pf a0 = v[4 * i], a1 = pf_cmul(v[4 * i + 1], w1), a2 = pf_cmul(v[4 * i + 2], w2), a3 = pf_cmul(v[4 * i + 3], w3);
            pf_dft4(a0, a1, a2, a3);                               // outputs m = 0..3: Z[j_i + 256 m]
            v[4 * i] = a0; v[4 * i + 1] = a1; v[4 * i + 2] = a2; v[4 * i + 3] = a3;
        }
        // (slot order i + 4 m, as fs_bin has it)
        pf o[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m) o[i + 4 * m] = v[4 * i + m];
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) v[s2] = o[s2];
    }
#else
    {
        const int k = t & 15, base = (t - k) * 16 + k;
#pragma unroll
        for (int m = 0; m < 16; ++m) L[base + 16 * m] = v[scr_perm(m)];
    }
    // last pass: radix 4, NS = 256: butterflies j = t + 64 b, inputs L[j + 256 r], twiddle exp(-2 pi i j / 1024)^r
    {
        pf x[16];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[4 * b + r] = L[t + 64 * b + 256 * r];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const pf w16[4] = {pfmk(1.0f, 0.0f), pfmk(c1, -s1), pfmk(h, -h), pfmk(s1, -c1)};       // exp(-2 pi i b / 16)
            const pf w1 = b == 0 ? tw.tw3 : pf_cmul(tw.tw3, w16[b]);
            const pf w2 = pf_cmul(w1, w1), w3 = pf_cmul(w2, w1);
            pf a0 = x[4 * b], a1 = pf_cmul(x[4 * b + 1], w1), a2 = pf_cmul(x[4 * b + 2], w2), a3 = pf_cmul(x[4 * b + 3], w3);
            pf_dft4(a0, a1, a2, a3);                               // outputs m = 0..3: Z[j + 256 m]
            v[b] = a0; v[b + 4] = a1; v[b + 8] = a2; v[b + 12] = a3;   // slot q = b + 4 m
        }
    }
#endif
    (void)c1; (void)s1; (void)h;
}

#ifndef GF3_FS_WPS
#define GF3_FS_WPS 2
#endif
#ifndef GF3_FS_EARLY
#define GF3_FS_EARLY 1        /* 1: the next segment's samples are requested before this one's transform (32 registers in flight) */
#endif

template <int DT>
__global__ __launch_bounds__(64, GF3_FS_WPS) void corr_screen_kernel(FScreenArgs a) {
    __shared__ pf L[GF3_FS_NC];
    constexpr int NC = GF3_FS_NC;
    typedef typename RawT<DT>::E E;
    const int t = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int64_t s0 = b * a.stride + a.win_lo;      // absolute sample index of lag 0 of this window
    const int W = a.W;
    const float c8[8] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                         0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};
    const float s8[8] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                         0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f};
    // One base twiddle per pass and for the split; their powers are recomputed per transform (kept in registers across the
    // partition loop -- 12 + 8 complex -- they change nothing: same-box A/B 2.02 vs 1.96 ms; the kernel waits on LDS round
    // trips, not on VALU issue: SQ counters in profiles/r04_fscreen_study.md).
    const pf* twp = (const pf*)a.tw;
    FsTw tw;
    tw.tw2 = twp[(t & 15) * 4];
    tw.tw3 = twp[fs_bin(t, 0)];
    pf wb = ((const pf*)a.twn)[t];
    auto wsplit = [&](int r) { return (t == 0 && r == 0) ? pfmk(0.0f, -1.0f) : pf_cmul(wb, pfmk(c8[r], -s8[r])); };
    auto refresh = [&]() { asm volatile("" : "+v"(tw.tw2), "+v"(tw.tw3), "+v"(wb)); };
    // acc[2r] accumulates 2 Y[k]; acc[2r+1] accumulates conj(2 Y[1024 - k]) -- the form the inverse split wants, and the one
    // a plain (un-conjugated) multiply-add produces: conj(X conj(H)) = conj(X) H
    pf acc[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = pfmk(0.0f, 0.0f);
    float accDC = 0.0f, accNy = 0.0f, ebound = 0.0f;
    const int need = a.Lp + a.Wmax - 1;               // samples of a segment that reach valid lags
    RawPair<DT> nxt[16];
    auto fetch = [&](int q) {
        const int64_t seg = s0 + (int64_t)q * a.Lp;
        const bool inside = seg >= 0 && seg + 2 * NC <= a.n_in;
        if (inside) {                                 // (uniform) unguarded pair loads; what lies past the samples that reach valid lags is zeroed
            const E* base = (const E*)a.in + seg;
            const unsigned t2 = 2u * (unsigned)t;
#pragma unroll
            for (int r = 0; r < 16; ++r) nxt[r].load_u(base, t2 + 128u * (unsigned)r);
            if (need < 2 * NC) {                      // (uniform: a window narrower than the plan's)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = 2 * (t + 64 * r);
                    if (j >= need) nxt[r].v.a = 0;
                    if (j + 1 >= need) nxt[r].v.b = 0;
                }
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 2 * (t + 64 * r);
            nxt[r].zero();
            if (j < need && seg + j >= 0 && seg + j < a.n_in) nxt[r].v.a = ((const E*)a.in)[seg + j];
            if (j + 1 < need && seg + j + 1 >= 0 && seg + j + 1 < a.n_in) nxt[r].v.b = ((const E*)a.in)[seg + j + 1];
        }
    };
    fetch(0);
    pf v[16];
    for (int q = 0; q < a.Q; ++q) {
        pf e2p = pfmk(0.0f, 0.0f);
#pragma unroll
        for (int r = 0; r < 16; ++r) { v[r] = pfmk((float)nxt[r].v.a, (float)nxt[r].v.b); e2p = __builtin_elementwise_fma(v[r], v[r], e2p); }
        float e2 = e2p.x + e2p.y;
        // (f64 samples: 16 raw pairs are 64 registers, which do not fit beside the transform -- the next segment is then
        //  requested after this one's multiply-adds, its latency covered by the SIMD's other wave)
        constexpr bool EARLY = GF3_FS_EARLY && DT != DT_F64;
        if (EARLY && q + 1 < a.Q) fetch(q + 1);
        // this partition's spectrum (an L2-resident table): requested BEFORE the transform, whose LDS round trips then cover
        // the L2 latency -- asked for next to the multiply-adds it sat, exposed, between the split's reads and the first fma
        // (f64 samples: no registers for that; two batches after the transform)
        const float4* Hp = a.Hs + (int64_t)q * 8 * 64;
        float4 hq[8];
        if constexpr (EARLY) {
#pragma unroll
            for (int r = 0; r < 8; ++r) hq[r] = Hp[r * 64 + t];
        }
        const float h0 = a.H0N[2 * q], hN = a.H0N[2 * q + 1];
        refresh();
        fs_fft1024(v, L, tw, t);
#pragma unroll
        for (int m = 0; m < 16; ++m) L[fs_bin(t, m)] = v[m];               // natural order for the packed-real split
        const pf z0 = L[0];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
        if constexpr (!EARLY) {
#pragma unroll
            for (int r = 4 * half; r < 4 * half + 4; ++r) hq[r] = Hp[r * 64 + t];
        }
#pragma unroll
        for (int r = 4 * half; r < 4 * half + 4; ++r) {
            const bool self = (t == 0 && r == 0);
            const int k = self ? NC / 2 : t + 64 * r;
            const pf A = L[k], Bm = L[(NC - k) & (NC - 1)];
            const pf Ee = pf_add_conj(A, Bm), Dd = pf_sub_conj(A, Bm);       // (halvings folded into `inv`: 2 X below)
            const pf Pw = pf_cmul(Dd, wsplit(r));                               // D w; O = -i D w
            const pf Xk = pf_add_negi(Ee, Pw), Ym = pf_sub_negi(Ee, Pw);     // 2 X[k] = E + O;  conj(2 X[1024 - k]) = E - O
            acc[2 * r] = pf_cfma_conj(Xk, pfmk(hq[r].x, hq[r].y), acc[2 * r]);
            acc[2 * r + 1] = pf_cfma(Ym, pfmk(hq[r].z, hq[r].w), acc[2 * r + 1]);
        }
        }
        accDC = fmaf(2.0f * (z0.x + z0.y), h0, accDC);
        accNy = fmaf(2.0f * (z0.x - z0.y), hN, accNy);
        e2 = scr_wave_reduce<false>(e2);
        ebound = fmaf(a.Hinf[q], sqrtf(e2 + GF3_SCR_UFLOW) * 1.0001f, ebound);
        if (!EARLY && q + 1 < a.Q) fetch(q + 1);
    }
    const float Eb = ebound * GF3_SCR_GAMMA * 1.0001f + 1e-37f;
    // ---- inverse real FFT of the accumulated Hermitian spectrum (the construction of corr_kernel, fp32)
    refresh();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const bool self = (t == 0 && r == 0);
        const int k = self ? NC / 2 : t + 64 * r;
        const pf A = acc[2 * r], Bc = acc[2 * r + 1];                       // (acc[2r+1] already holds the conjugate)
        const pf Ee = pf_add(A, Bc), Dd = pf_sub(A, Bc);
        const pf Op = pf_cmul_conj(Dd, wsplit(r));                             // * exp(+2 pi i k / 2048)
        // Zk = E + i Op, Zm = conj(E) + i conj(Op); stored conjugated: conj(Zk) = conj(E) - i conj(Op), conj(Zm) = E - i Op
        const pf cE = pfmk(Ee.x, -Ee.y), cO = pfmk(Op.x, -Op.y);
        L[k] = pf_add_negi(cE, cO);
        if (!self) L[NC - k] = pf_add_negi(Ee, Op);
    }
    if (t == 0) L[0] = pfmk(accDC + accNy, -(accDC - accNy));               // conj(E + i Op), doubled like the rest
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = L[t + 64 * r];
    fs_fft1024(v, L, tw, t);
    // z = conj(FFT(conj Z)) / NC; y[2n] = Re z, y[2n+1] = Im z, n = t + 64 q -> as floats into the buffer
    const float inv = 0.25f / (float)NC;              // 1/NC of the inverse transform, 1/2 of each of the two splits
    float* y = (float*)L;
#pragma unroll
    for (int m = 0; m < 16; ++m) L[fs_bin(t, m)] = pfmk(v[m].x * inv, -v[m].y * inv);
    // ---- the window rule on bounded values
    float mx = -INFINITY;
    bool nonfinite = !(Eb < INFINITY);
    for (int j = t; j < W; j += 64) { const float yj = y[j]; mx = fmaxf(mx, yj); nonfinite = nonfinite || !(fabsf(yj) < INFINITY); }
    mx = scr_wave_reduce<true>(mx);
    if (a.y32) for (int j = t; j < W; j += 64) a.y32[b * (int64_t)W + j] = y[j];
    const float mlo = mx - Eb, mhi = mx + Eb;
    const float tlo = a.thresh * mlo * (1.0f - 1e-6f) * 0.9999f;            // below this (with the bound) a lag cannot pass
    const float thi = a.thresh * mhi * (1.0f + 1e-6f) * 1.0001f;            // above this (with the bound) it passes
    const float e2x = 2.0f * Eb * 1.001f + 1e-37f;
    int first_yes = 0x7fffffff, first_maybe = 0x7fffffff;
    for (int j = 1 + t; j < W - 1; j += 64) {
        const float ym = y[j - 1], y0 = y[j], yp = y[j + 1];
        if (y0 + Eb < tlo) continue;                                        // NOT: cannot reach the threshold
        const float d1 = y0 - ym, d2 = yp - y0;
        const bool up1 = d1 > e2x, dn1 = d1 < -e2x, up2 = d2 > e2x, dn2 = d2 < -e2x;
        if ((up1 && up2) || (dn1 && dn2)) continue;                         // NOT: surely no extremum
        const bool ext = (up1 && dn2) || (dn1 && up2);
        if (ext && (y0 - Eb > thi)) first_yes = min(first_yes, j);
        else first_maybe = min(first_maybe, j);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        first_yes = min(first_yes, __shfl_xor(first_yes, d, 64));
        first_maybe = min(first_maybe, __shfl_xor(first_maybe, d, 64));
    }
    const bool anybad = __any((int)nonfinite);
    // (a window whose maximum the bound cannot tell from zero, a threshold outside (0, inf): the rule's own special cases)
    const bool decidable = !anybad && mlo > 0.0f && a.thresh > 0.0f && a.thresh < INFINITY;
    const bool resolved = decidable && !(first_maybe < first_yes);       // no undecided lag before the first certain one (or neither exists)
    if (t == 0) {
        const bool found = first_yes != 0x7fffffff;
        if (resolved) a.starts[b] = found ? (s0 + first_yes + a.Lc) : -1;
        else a.unresolved[atomicAdd(a.n_unresolved, 1)] = (int)b;
        if (a.err) a.err[b] = Eb;
        if (a.cls) a.cls[b] = resolved ? (found ? 0 : 1) : 2;
    }
}
