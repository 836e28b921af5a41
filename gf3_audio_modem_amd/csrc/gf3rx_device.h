// Device-side building blocks of libgf3rx (gfx950 / CDNA4, wave64).
//
// Everything computes in fp64 (SURVEY.md §7 "Precision": fp32 fails both the
// bit-exact and the 1e-6 symbol bar on real recordings).  Samples are stored as
// f64/f32/i16/u8 and widened in registers.
//
// FFT: a length-N real symbol is packed as NC = N/2 complex points
// z[n] = x[2n] + i x[2n+1]; NC/8 threads each own 8 points and run Stockham
// autosort passes (radix 8 with a radix-4 tail) through one in-place LDS buffer;
// the packed-real split is then done on (k, NC-k) pairs held by ONE thread, so
// the per-carrier equaliser state lives in that thread's registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double2 cplx;

#define GF3_DEV __device__ __forceinline__

GF3_DEV cplx cmk(double a, double b) { return make_double2(a, b); }
GF3_DEV cplx cadd(cplx a, cplx b) { return cmk(a.x + b.x, a.y + b.y); }
GF3_DEV cplx csub(cplx a, cplx b) { return cmk(a.x - b.x, a.y - b.y); }
GF3_DEV cplx cconj(cplx a) { return cmk(a.x, -a.y); }
GF3_DEV cplx cscale(cplx a, double s) { return cmk(a.x * s, a.y * s); }
GF3_DEV cplx cmul(cplx a, cplx b) { return cmk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
GF3_DEV cplx cmul_conj(cplx a, cplx b) { return cmk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)
GF3_DEV cplx mul_negi(cplx a) { return cmk(a.y, -a.x); }   // a * (-i)
GF3_DEV cplx mul_posi(cplx a) { return cmk(-a.y, a.x); }   // a * (+i)

// complex division the way NumPy does it for complex128 (Smith's method)
GF3_DEV cplx cdiv_np(cplx a, cplx b) {
    const double br = fabs(b.x), bi = fabs(b.y);
    if (br >= bi) {
        if (br == 0.0 && bi == 0.0) return cmk(a.x / br, a.y / bi);
        const double rat = b.y / b.x;
        const double scl = 1.0 / (b.x + b.y * rat);
        return cmk((a.x + a.y * rat) * scl, (a.y - a.x * rat) * scl);
    }
    const double rat = b.x / b.y;
    const double scl = 1.0 / (b.y + b.x * rat);
    return cmk((a.x * rat + a.y) * scl, (a.y * rat - a.x) * scl);
}

// |x + iy| the way NumPy's complex128 `absolute` loop computes it (the reference's `abs(symbols - constellation)`,
// OFDM.py:490): larger * sqrt(fma(r, r, 1)) with r = smaller / larger, r = 0 when larger == 0 or smaller == inf;
// an infinite part makes both parts infinite, a NaN part makes both NaN.  Every step is correctly rounded here as
// there, so the value is bit-identical (tests/golden g5: 185 556 distances) -- which is what decides exact and
// near ties of the hard demapper: two different squared distances can round to the same |.|, and argmin then
// returns the first of them.
GF3_DEV double np_cabs(double x, double y) {
    double re = fabs(x), im = fabs(y);
    const bool rinf = re == INFINITY, iinf = im == INFINITY;
    if (rinf) im = INFINITY;
    if (iinf) re = INFINITY;
    const bool rnan = re != re, inan = im != im;
    if (rnan) im = re;
    if (inan) re = im;
    const double larger = fmax(re, im), smaller = fmin(im, re);      // (both NaN when either is)
    const bool nodiv = larger == 0.0 || smaller == INFINITY;
    const double ratio = nodiv ? 0.0 : smaller / larger;
    return sqrt(fma(ratio, ratio, 1.0)) * larger;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0),
// which would make every FFT barrier wait for the next symbol's prefetch loads; here global
// loads stay in flight across the barrier (their consumers get their own s_waitcnt).
GF3_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// XCD-aware work order for kernels whose neighbouring workgroups re-read each other's data: workgroups are dealt
// round-robin over the 8 XCDs (blockIdx % 8 share one XCD and its L2), so logical work item
// (blockIdx % 8) * (grid / 8) + blockIdx / 8 makes every XCD walk a contiguous eighth of the items.  The grid must be a
// multiple of 8 (pad; padded items return).  Placement is a speed matter only.
GF3_DEV int64_t xcd_order(unsigned bid, unsigned grid) { return (int64_t)(bid & 7u) * (int64_t)(grid >> 3) + (int64_t)(bid >> 3); }

// Opaque copy of a per-thread index.  Everything derived from the copy is recomputed where it
// is used instead of being hoisted out of the symbol loop and kept live (or spilled) across it.
GF3_DEV int launder(int x) { asm volatile("" : "+v"(x)); return x; }
GF3_DEV double launder(double x) { asm volatile("" : "+v"(x)); return x; }

enum { DT_F64 = 0, DT_F32 = 1, DT_I16 = 2, DT_U8 = 3 };

// Two consecutive samples as stored (one packed complex point), kept raw in
// registers while in flight so that a prefetched symbol costs 8-32 VGPRs.
// Sample offsets are arbitrary (sync decides them), so pair loads are only
// element-aligned; gfx950 global loads accept that.
template <int DT> struct RawT;
template <> struct RawT<DT_F64> { typedef double E; };
template <> struct RawT<DT_F32> { typedef float E; };
template <> struct RawT<DT_I16> { typedef int16_t E; };
template <> struct RawT<DT_U8>  { typedef uint8_t E; };
template <int DT> struct RawPair {
    typedef typename RawT<DT>::E E;
    struct __attribute__((packed, aligned(sizeof(E)))) P2 { E a, b; };
    P2 v;
    GF3_DEV void load(const void* p, int64_t i) { v = *(const P2*)((const E*)p + i); }
    // wave-uniform base + 32-bit per-lane element offset: selects the scalar-base addressing form of global_load
    // (one 32-bit VGPR offset) instead of 64-bit per-lane address arithmetic
    GF3_DEV void load_u(const E* base, unsigned off) { v = *(const P2*)(base + off); }
    GF3_DEV void zero() { v.a = 0; v.b = 0; }
    GF3_DEV cplx get() const { return cmk((double)v.a, (double)v.b); }
};
template <int DT> GF3_DEV double load_sample_t(const void* p, int64_t i) {
    return (double)((const typename RawT<DT>::E*)p)[i];
}

GF3_DEV double load_sample(const void* p, int64_t i, int dt) {
    switch (dt) {
        case DT_F64: return ((const double*)p)[i];
        case DT_F32: return (double)((const float*)p)[i];
        case DT_I16: return (double)((const int16_t*)p)[i];
        default:     return (double)((const uint8_t*)p)[i];
    }
}
GF3_DEV cplx load_pair(const void* p, int64_t i, int dt) {
    return cmk(load_sample(p, i, dt), load_sample(p, i + 1, dt));
}
GF3_DEV double load_sample_clamped(const void* p, int64_t i, int64_t n, int dt) {
    return (i >= 0 && i < n) ? load_sample(p, i, dt) : 0.0;
}

// ---------------------------------------------------------------- butterflies
#define GF3_SQRT1_2 0.70710678118654752440

GF3_DEV void bfly8(cplx* v) {
    const cplx a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]);
    const cplx a2 = cadd(v[2], v[6]), a3 = mul_negi(csub(v[2], v[6]));
    const cplx a4 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    const cplx a6 = cadd(v[3], v[7]), a7 = mul_negi(csub(v[3], v[7]));
    const cplx b0 = cadd(a0, a2), b1 = cadd(a1, a3), b2 = csub(a0, a2), b3 = csub(a1, a3);
    const cplx b4 = cadd(a4, a6), o1 = cadd(a5, a7), b6 = mul_negi(csub(a4, a6)), o3 = csub(a5, a7);
    // b5 = o1 (1-i)/sqrt2 and b7 = o3 (-1-i)/sqrt2 are never formed: the 1/sqrt2 rides on the fma of b +- b5, b +- b7
    const double p1 = o1.x + o1.y, m1 = o1.y - o1.x;            // b5 = (p1, m1) / sqrt2
    const double m3 = o3.y - o3.x, p3 = o3.x + o3.y;            // b7 = (m3, -p3) / sqrt2
    v[0] = cadd(b0, b4); v[4] = csub(b0, b4);
    v[1] = cmk(fma(GF3_SQRT1_2, p1, b1.x), fma(GF3_SQRT1_2, m1, b1.y));
    v[5] = cmk(fma(-GF3_SQRT1_2, p1, b1.x), fma(-GF3_SQRT1_2, m1, b1.y));
    v[2] = cadd(b2, b6); v[6] = csub(b2, b6);
    v[3] = cmk(fma(GF3_SQRT1_2, m3, b3.x), fma(-GF3_SQRT1_2, p3, b3.y));
    v[7] = cmk(fma(-GF3_SQRT1_2, m3, b3.x), fma(GF3_SQRT1_2, p3, b3.y));
}

// powers of a unit twiddle w = exp(-i t) by the three-term recurrence w^(k+1) = 2 cos(t) w^k - w^(k-1): two fma per
// power instead of the four operations of a complex multiply, no table traffic.  Its error grows like k^2 ulp
// (k <= 7 here: <= ~1e-14 relative, three orders inside the tightest parity bar).
GF3_DEV cplx tw_next(double c2, cplx wk, cplx wkm1) { return cmk(fma(c2, wk.x, -wkm1.x), fma(c2, wk.y, -wkm1.y)); }
// Twiddle multiplication fused into the first butterfly stage: with x' = x w, the pair (a + b', a - b') costs
// a complex fma (4) plus 2a - (a + b') (2) instead of a complex multiply, an add and a subtract (8); when a is
// itself twiddled, 10 instead of 12.
GF3_DEV cplx cfma(cplx a, cplx w, cplx c) {           // c + a w
    return cmk(fma(a.x, w.x, fma(-a.y, w.y, c.x)), fma(a.x, w.y, fma(a.y, w.x, c.y)));
}
GF3_DEV cplx twice_minus(cplx u, cplx s) { return cmk(fma(2.0, u.x, -s.x), fma(2.0, u.y, -s.y)); }   // 2u - s
GF3_DEV void bfly4_tw(cplx* v, cplx w) {              // v[r] *= w^r, then the radix-4 butterfly
    const double c2 = w.x + w.x;
    const cplx w2 = cmk(fma(c2, w.x, -1.0), c2 * w.y);
    const cplx w3 = tw_next(c2, w2, w);
    const cplx s0 = cfma(v[2], w2, v[0]), s1 = twice_minus(v[0], s0);
    const cplx u1 = cmul(v[1], w);
    const cplx s2 = cfma(v[3], w3, u1), s3 = mul_negi(twice_minus(u1, s2));
    v[0] = cadd(s0, s2); v[1] = cadd(s1, s3); v[2] = csub(s0, s2); v[3] = csub(s1, s3);
}
GF3_DEV void bfly8_tw(cplx* v, cplx w) {              // v[r] *= w^r, then the radix-8 butterfly
    const double c2 = w.x + w.x;
    const cplx w2 = cmk(fma(c2, w.x, -1.0), c2 * w.y);
    const cplx w3 = tw_next(c2, w2, w), w4 = tw_next(c2, w3, w2), w5 = tw_next(c2, w4, w3);
    const cplx w6 = tw_next(c2, w5, w4), w7 = tw_next(c2, w6, w5);
    const cplx a0 = cfma(v[4], w4, v[0]), a1 = twice_minus(v[0], a0);
    const cplx u2 = cmul(v[2], w2);
    const cplx a2 = cfma(v[6], w6, u2), a3 = mul_negi(twice_minus(u2, a2));
    const cplx u1 = cmul(v[1], w);
    const cplx a4 = cfma(v[5], w5, u1), a5 = twice_minus(u1, a4);
    const cplx u3 = cmul(v[3], w3);
    const cplx a6 = cfma(v[7], w7, u3), a7 = mul_negi(twice_minus(u3, a6));
    const cplx b0 = cadd(a0, a2), b1 = cadd(a1, a3), b2 = csub(a0, a2), b3 = csub(a1, a3);
    const cplx b4 = cadd(a4, a6), o1 = cadd(a5, a7), b6 = mul_negi(csub(a4, a6)), o3 = csub(a5, a7);
    const double p1 = o1.x + o1.y, m1 = o1.y - o1.x;
    const double m3 = o3.y - o3.x, p3 = o3.x + o3.y;
    v[0] = cadd(b0, b4); v[4] = csub(b0, b4);
    v[1] = cmk(fma(GF3_SQRT1_2, p1, b1.x), fma(GF3_SQRT1_2, m1, b1.y));
    v[5] = cmk(fma(-GF3_SQRT1_2, p1, b1.x), fma(-GF3_SQRT1_2, m1, b1.y));
    v[2] = cadd(b2, b6); v[6] = csub(b2, b6);
    v[3] = cmk(fma(GF3_SQRT1_2, m3, b3.x), fma(-GF3_SQRT1_2, p3, b3.y));
    v[7] = cmk(fma(-GF3_SQRT1_2, m3, b3.x), fma(GF3_SQRT1_2, p3, b3.y));
}
template <int R> GF3_DEV void bfly_tw(cplx* v, cplx w) { if constexpr (R == 8) bfly8_tw(v, w); else bfly4_tw(v, w); }

// ---------------------------------------------------------------- LDS FFT
// LDS footprint of one FFT buffer, in cplx elements (first exchange is padded
// by one element per 8 to break the stride-8 store conflict).
// Fused sizes (1024/2048): two NC-point buffers used ping-pong, so one barrier per exchange
// (a pass never stores into the buffer other waves may still be loading from); the first,
// stride-8, exchange is XOR-swizzled instead of padded.  Other sizes: one padded buffer, two
// barriers per exchange.
template <int NC> struct FftGeom {
    static constexpr int T = NC / 8;
    static constexpr bool PINGPONG = (NC == 1024 || NC == 2048);
    static constexpr int LDS_ELEMS = PINGPONG ? 2 * NC : NC + NC / 8;
    static constexpr int LDS_ELEMS_INPLACE = NC + NC / 8;
};

// Per-thread twiddle bases: the index k of every pass depends only on the thread,
// so one unit twiddle per pass is loaded once per workgroup and kept in registers.
template <int NC> struct FftTw {
    cplx b2, b3, b4, c4;
    // Made opaque once per transform: without it LLVM hoists every twiddle POWER
    // (w^2..w^7 of each pass, ~80 VGPRs) out of the symbol loop and keeps them live across it.
    // (in place: the empty asm "redefines" the registers it is given, no copies are made)
    GF3_DEV void refresh() {
        asm volatile("" : "+v"(b2.x), "+v"(b2.y), "+v"(b3.x), "+v"(b3.y));
        asm volatile("" : "+v"(b4.x), "+v"(b4.y), "+v"(c4.x), "+v"(c4.y));
    }
    // passes 2 and 3 only: a kernel with ~16 registers to spare lets the (few) last-pass powers be hoisted
    GF3_DEV void refresh_inner() { asm volatile("" : "+v"(b2.x), "+v"(b2.y), "+v"(b3.x), "+v"(b3.y)); }      // c4: step to the second butterfly of a 2-butterfly pass, or (fused
                              // sizes) the base twiddle of the mirrored butterfly of the last pass
    GF3_DEV void init(int tid, const cplx* __restrict__ tw);
};

template <int NC, int R, int NS>
GF3_DEV void fft_pass(cplx (&v)[8], const cplx* src, cplx* dst, cplx wbase, cplx wstep, int tid) {
    constexpr int T = NC / 8, NB = 8 / R;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = tid + b * T;
#pragma unroll
        for (int r = 0; r < R; ++r) v[b * R + r] = src[j + r * (NC / R)];
    }
    if (src == dst) lds_barrier();                          // in place: all loads before any store
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = tid + b * T;
        const int k = j & (NS - 1);
        cplx w = wbase;
        if (b == 1 && NS > T) w = cmul(wbase, wstep);      // k advances by T for the second butterfly
        bfly_tw<R>(&v[b * R], w);
        const int base = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) dst[base + r * NS] = v[b * R + r];
    }
    lds_barrier();
}

// Slot layout of one real-FFT result held in registers: thread t owns 8 bins,
// slot s = 2r + h  ->  h = 0: bin k_r,  h = 1: bin NC - k_r  (a mirrored pair per r).
//   fused sizes (NC = 1024, 2048: last pass is radix 4 with two butterflies per thread):
//       k_r = t + r*NC/4;   thread 0 (whose butterflies mirror onto themselves): NC/2, NC/4, NC/8, 3NC/8
//   other sizes: k_r = t + r*NC/8; thread 0, r = 0: NC/2
// Thread 0's slot 1 repeats bin NC/2 and is not "live".  DC/Nyquist come back separately (z0).
template <int NC> struct Spec {
    static constexpr bool FUSED = (NC == 1024 || NC == 2048);
    static constexpr int Q = NC / 4, T = NC / 8;
    GF3_DEV static int bin(int t, int s) {
        const int r = s >> 1;
        int k;
        if constexpr (FUSED) {
            const int k0 = (r == 0) ? NC / 2 : (r == 1 ? Q : (r == 2 ? Q / 2 : 3 * Q / 2));
            k = (t == 0) ? k0 : t + r * Q;
        } else {
            k = (t == 0 && r == 0) ? NC / 2 : t + r * T;
        }
        return (s & 1) ? NC - k : k;
    }
    GF3_DEV static bool live(int t, int s) { return !(t == 0 && s == 1); }
    // exp(-2 pi i k_r / N), N = 2 NC, from wb = exp(-2 pi i t / N)
    GF3_DEV static cplx pair_tw(int t, int r, cplx wb) {
        const double c8 = 0.92387953251128675613, s8 = 0.38268343236508977173;
        if constexpr (FUSED) {          // k advances by N/8 per r
            cplx w = wb;
            if (r == 1) w = cmk((wb.x + wb.y) * GF3_SQRT1_2, (wb.y - wb.x) * GF3_SQRT1_2);
            if (r == 2) w = mul_negi(wb);
            if (r == 3) w = cmk((wb.y - wb.x) * GF3_SQRT1_2, -(wb.x + wb.y) * GF3_SQRT1_2);
            if (t == 0) {
                if (r == 0) w = cmk(0.0, -1.0);
                if (r == 2) w = cmk(c8, -s8);
                if (r == 3) w = cmk(s8, -c8);
            }
            return w;
        } else {                        // k advances by N/16 per r
            const cplx r16[4] = {cmk(1.0, 0.0), cmk(c8, -s8), cmk(GF3_SQRT1_2, -GF3_SQRT1_2), cmk(s8, -c8)};
            if (r == 0) return t == 0 ? cmk(0.0, -1.0) : wb;
            return cmul(wb, r16[r]);
        }
    }
};

// packed-real split of one mirrored pair: A = Z[k], Bm = Z[NC-k], w = exp(-2 pi i k / N).
// TWICE = true returns 2 X[k], 2 X[N/2-k]: the two halvings are dropped (four multiplies per pair); callers whose
// results are ratios or signs of spectra fold the factor of two -- exact in binary -- into a constant they apply anyway.
template <bool TWICE = false>
GF3_DEV void real_split(cplx A, cplx Bm, cplx w, cplx& Xk, cplx& Xm) {
    const cplx B = cconj(Bm);
    const cplx E = TWICE ? cadd(A, B) : cscale(cadd(A, B), 0.5);
    const cplx D = TWICE ? csub(A, B) : cscale(csub(A, B), 0.5);
    Xk = cfma(mul_negi(D), w, E);                    // E + O,  O = -i D w
    Xm = cconj(twice_minus(E, Xk));                  // conj(E - O) = conj(2E - Xk)
}

// Passes through LDS.  In: v[r] = z[tid + r*NC/8].  ALL = false stops before the last pass
// (fused sizes).  `flip` selects which ping-pong buffer the first exchange uses; the caller
// alternates it per transform.  Returns the buffer that holds the result.
template <int NC, bool ALL, bool PP = FftGeom<NC>::PINGPONG>
GF3_DEV cplx* fft_passes(cplx (&v)[8], cplx* lds, const FftTw<NC>& ft, int tid, int flip) {
    constexpr int T = NC / 8;
    const cplx w8 = cmk(GF3_SQRT1_2, -GF3_SQRT1_2);              // exp(-i pi/4): T steps of the last pass
    cplx* A = PP ? lds + (flip ? NC : 0) : lds;
    cplx* B = PP ? lds + (flip ? 0 : NC) : lds;
    bfly8(v);
    if constexpr (PP) {
        // exchange 1, XOR-swizzled: logical i = tid*8 + r  ->  i ^ ((i >> 3) & 7)
#pragma unroll
        for (int r = 0; r < 8; ++r) A[tid * 8 + (r ^ (tid & 7))] = v[r];
        lds_barrier();
        const int ts = tid ^ ((tid >> 3) & 7);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = A[ts + r * T];
    } else {
        lds_barrier();                     // previous users of the buffer are done
#pragma unroll
        for (int r = 0; r < 8; ++r) A[tid * 9 + r] = v[r];        // logical tid*8+r, padded
        lds_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int i = tid + r * T; v[r] = A[i + (i >> 3)]; }
        lds_barrier();
    }
    {
        const int k = tid & 7;
        bfly8_tw(v, ft.b2);
        const int base = (tid - k) * 8 + k;
#pragma unroll
        for (int r = 0; r < 8; ++r) B[base + r * 8] = v[r];
    }
    lds_barrier();
    if constexpr (NC == 512) {
        fft_pass<NC, 8, 64>(v, B, A, ft.b3, w8, tid);
        return A;
    } else if constexpr (NC == 1024) {
        fft_pass<NC, 4, 64>(v, B, A, ft.b3, w8, tid);
        if constexpr (ALL) { fft_pass<NC, 4, 256>(v, A, B, ft.b4, w8, tid); return B; }
        return A;
    } else if constexpr (NC == 2048) {
        fft_pass<NC, 8, 64>(v, B, A, ft.b3, w8, tid);
        if constexpr (ALL) { fft_pass<NC, 4, 512>(v, A, B, ft.b4, w8, tid); return B; }
        return A;
    } else {
        static_assert(NC == 4096, "unsupported FFT size");
        fft_pass<NC, 8, 64>(v, B, A, ft.b3, w8, tid);
        fft_pass<NC, 8, 512>(v, A, B, ft.b4, w8, tid);
        return B;
    }
}

// Forward complex FFT of NC points; returns the LDS buffer holding Z[0..NC) in natural order.
// Ping-pong sizes: the caller must have a barrier between the last reads of both buffers by a
// previous user and this call (this entry point is only used after such a barrier).
template <int NC, bool PP = FftGeom<NC>::PINGPONG>
GF3_DEV cplx* fft_core(cplx (&v)[8], cplx* lds, const FftTw<NC>& ft, int tid) {
    return fft_passes<NC, true, PP>(v, lds, ft, tid, 0);
}

template <int NC> GF3_DEV void FftTw<NC>::init(int tid, const cplx* __restrict__ tw) {
    b2 = tw[(tid & 7) * (NC / 64)];                          // pass 2: radix 8, NS = 8
    if constexpr (NC == 1024) b3 = tw[(tid & 63) * (NC / 256)];      // radix 4, NS = 64
    else b3 = tw[(tid & 63) * (NC / 512)];                           // radix 8, NS = 64
    b4 = cmk(1.0, 0.0); c4 = cmk(1.0, 0.0);
    if constexpr (NC == 1024 || NC == 2048) {                // last pass radix 4, NS = NC/4: k = j
        b4 = tw[tid];
        c4 = tw[tid == 0 ? NC / 8 : NC / 4 - tid];           // mirrored butterfly j2 (fused split)
    } else if constexpr (NC == 4096) b4 = tw[tid & 511];
}

// Real FFT of one packed symbol with the spectrum left in registers in Spec<NC> slot order.
// In: v[r] = z[t + r*NC/8] (t = threadIdx.x); wb = exp(-2 pi i t / (2NC)); z0 (thread 0) = Z[0].
// TWICE: slots hold 2 X (see real_split); z0 is not affected.
// `flip` must alternate between consecutive calls in a workgroup (ping-pong hazard: the last
// pass of call i reads buffer A_i with no barrier after it; call i+1 starts by storing into the
// other buffer, which every wave finished reading before call i's final barrier).
template <int NC, bool PP = FftGeom<NC>::PINGPONG, bool TWICE = false>
GF3_DEV void rfft_regs(cplx (&v)[8], cplx* lds, const FftTw<NC>& ft, cplx wb, int t, cplx& z0, int flip) {
    if constexpr (Spec<NC>::FUSED) {
        constexpr int Q = NC / 4;
        const cplx* Z = fft_passes<NC, false, PP>(v, lds, ft, t, flip);
        // last pass (radix 4, NS = Q) on butterflies j1 = t and j2 = Q - t, whose outputs mirror
        // each other: Z[j1 + rQ] <-> Z[j2 + (3-r)Q].  Thread 0 takes the two self-mirrored ones.
        const int j2 = (t == 0) ? Q / 2 : Q - t;
        cplx a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { a[r] = Z[t + r * Q]; b[r] = Z[j2 + r * Q]; }
        bfly4_tw(a, ft.b4);
        bfly4_tw(b, ft.c4);
        z0 = a[0];
        // Only thread 0 pairs its outputs differently (its two butterflies mirror onto themselves).  The ~40 selects
        // that costs are confined to its wave by a scalar branch; the other waves run the plain pairing.
        if (__builtin_amdgcn_readfirstlane(t) < 64) {
            const bool t0 = (t == 0);
            auto sel = [&](cplx x, cplx y) { return cmk(t0 ? x.x : y.x, t0 ? x.y : y.y); };
            const cplx A[4] = {sel(a[2], a[0]), a[1], sel(b[0], a[2]), sel(b[1], a[3])};
            const cplx Bm[4] = {sel(a[2], b[3]), sel(a[3], b[2]), sel(b[3], b[1]), sel(b[2], b[0])};
#pragma unroll
            for (int r = 0; r < 4; ++r) real_split<TWICE>(A[r], Bm[r], Spec<NC>::pair_tw(t, r, wb), v[2 * r], v[2 * r + 1]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) real_split<TWICE>(a[r], b[3 - r], Spec<NC>::pair_tw(1, r, wb), v[2 * r], v[2 * r + 1]);
        }
    } else {
        const cplx* Z = fft_passes<NC, true, PP>(v, lds, ft, t, 0);
        z0 = Z[0];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = Spec<NC>::bin(t, 2 * r);
            const cplx A = Z[k], Bm = Z[NC - k];
            real_split<TWICE>(A, Bm, Spec<NC>::pair_tw(t, r, wb), v[2 * r], v[2 * r + 1]);
        }
    }
}

// ---------------------------------------------------------------- block collectives
GF3_DEV double wave_incl_scan(double x) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}
GF3_DEV double wave_sum(double x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}
GF3_DEV double wave_max(double x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x = fmax(x, __shfl_xor(x, d, 64));
    return x;
}
GF3_DEV int wave_min_i(int x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x = min(x, __shfl_xor(x, d, 64));
    return x;
}
// scratch: >= 16 doubles of LDS, not in use by anyone else; contains barriers
GF3_DEV double block_sum(double x, double* scratch) {
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    x = wave_sum(x);
    lds_barrier();
    if ((threadIdx.x & 63) == 0) scratch[wave] = x;
    lds_barrier();
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += scratch[i];
    return s;
}
GF3_DEV double block_max(double x, double* scratch) {
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    x = wave_max(x);
    lds_barrier();
    if ((threadIdx.x & 63) == 0) scratch[wave] = x;
    lds_barrier();
    double s = scratch[0];
    for (int i = 1; i < nw; ++i) s = fmax(s, scratch[i]);
    return s;
}
GF3_DEV int block_min_i(int x, int* scratch) {
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    x = wave_min_i(x);
    lds_barrier();
    if ((threadIdx.x & 63) == 0) scratch[wave] = x;
    lds_barrier();
    int s = scratch[0];
    for (int i = 1; i < nw; ++i) s = min(s, scratch[i]);
    return s;
}
// exclusive scan of per-thread totals (two independent scans at once)
GF3_DEV void block_excl_scan2(double a, double b, double* scratch, double& ea, double& eb) {
    const int wave = threadIdx.x >> 6;
    const double ia = wave_incl_scan(a), ib = wave_incl_scan(b);
    lds_barrier();
    if ((threadIdx.x & 63) == 63) { scratch[wave] = ia; scratch[8 + wave] = ib; }
    lds_barrier();
    double oa = 0.0, ob = 0.0;
    for (int i = 0; i < wave; ++i) { oa += scratch[i]; ob += scratch[8 + i]; }
    ea = oa + (ia - a);
    eb = ob + (ib - b);
}

// sin/cos by Cody-Waite reduction to [-pi/4, pi/4] (three-part pi/2, exact with fma for
// |x| < ~1e6) and the fdlibm kernel polynomials: <= 1 ulp there, ~30 fp64 ops.
GF3_DEV void sincos_fast(double x, double& s, double& c) {
    if (!(fabs(x) < 1.0e5)) { sincos(x, &s, &c); return; }
    const double fn = rint(x * 6.36619772367581382433e-01);
    double r = fma(-fn, 1.57079632673412561417e+00, x);
    r = fma(-fn, 6.07710050630396597660e-11, r);
    r = fma(-fn, 2.02226624879595063154e-21, r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                          2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03),
                          -1.66666666666666324348e-01);
    const double sr = fma(z * r, ps, r);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                          -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03),
                          4.16666666666666019037e-02);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)fn & 3;
    s = (q & 1) ? cr : sr;
    c = (q & 1) ? sr : cr;
    if (q & 2) s = -s;
    if ((q + 1) & 2) c = -c;
}
// 1/x: v_rcp_f64 seed + two Newton steps (full double precision for normal x)
GF3_DEV double rcp_nr(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
// 1/sqrt(x): v_rsq_f64 seed + two Newton steps
GF3_DEV double rsq_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * fma(-hx * y, y, 1.5);
    return y * fma(-hx * y, y, 1.5);
}
GF3_DEV cplx cis_fast(double x) { double s, c; sincos_fast(x, s, c); return cmk(c, s); }

// atan2 with fdlibm's atan kernel (break points 7/16, 11/16; 11-term odd polynomial),
// branch-free apart from selects: a = min/max in [0,1], reduce, evaluate, unfold octants.
// <= ~2 ulp; only feeds the phase-slope fit (tolerance 1e-11).
GF3_DEV double atan2_fast(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    const double a = mn * rcp_nr(mx);
    const bool lo = a < 0.4375, mid = a < 0.6875;
    const double num = lo ? a : (mid ? 2.0 * a - 1.0 : a - 1.0);
    const double den = lo ? 1.0 : (mid ? 2.0 + a : a + 1.0);
    const double t = num * rcp_nr(den);
    const double hi = lo ? 0.0 : (mid ? 4.63647609000806093515e-01 : 7.85398163397448278999e-01);
    const double lw = lo ? 0.0 : (mid ? 2.26987774529616870924e-17 : 3.06161699786838301793e-17);
    const double z = t * t, w = z * z;
    const double s1 = z * fma(w, fma(w, fma(w, fma(w, fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02),
                      6.66107313738753120669e-02), 9.09088713343650656196e-02), 1.42857142725034663711e-01),
                      3.33333333333329318027e-01);
    const double s2 = w * fma(w, fma(w, fma(w, fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02),
                      -7.69187620504482999495e-02), -1.11111104054623557880e-01), -1.99999999998764832476e-01);
    double r = hi - ((t * (s1 + s2) - lw) - t);
    if (!(mx > 0.0)) r = 0.0;                       // atan2(0, 0) = 0 (NaN stays NaN through mx)
    if (ay > ax) r = 1.57079632679489655800 - r;
    if (x < 0.0) r = 3.14159265358979311600 - r;
    return y < 0.0 ? -r : r;
}

// np.unwrap's correction for one phase step dd = p[n] - p[n-1]
GF3_DEV double unwrap_corr(double dd) {
    const double PI = 3.14159265358979323846, TWO_PI = 6.28318530717958647692;
    // np.mod(dd + pi, 2 pi): dd is a difference of two angles in [-pi, pi], so one conditional
    // add/subtract reproduces fmod + sign fix-up exactly (the subtraction is exact by Sterbenz)
    double m = dd + PI;
    if (m >= TWO_PI) m -= TWO_PI; else if (m < 0.0) m += TWO_PI;
    double ddmod = m - PI;
    if (ddmod == -PI && dd > 0.0) ddmod = PI;
    double corr = ddmod - dd;
    if (fabs(dd) < PI) corr = 0.0;
    return corr;
}
