// Sixteen points per thread, one wave per transform (gfx950, wave64).
//
// A 2048-point real transform packed as NC = 1024 complex points, held by ONE wave: thread t owns 16 points and the
// passes are 16 . 16 . 4 -- TWO exchanges through LDS instead of the three of the 8-points-per-thread core
// (gf3rx_device.h: 8 . 8 . 4 . 4 at this size), a third less LDS traffic per point.  And a workgroup that is a single
// wave needs no s_barrier at all: LDS serves one wave's instructions in order, so a store followed by another
// lane's load is ordered by issue, and the waves of a CU drift apart instead of meeting at every pass (one wave's
// exchange runs under another's butterflies).  The arithmetic per point is the same as the radix-8 core's (a
// twiddled radix-16 pass costs what a twiddled radix-8 plus half a radix-4 pass do).
//
// Stockham autosort, decimation in time, as in gf3rx_device.h:
//   pass 1: radix 16, NS = 1    butterfly j = t        in  z[t + 64 r]       out  [16 t + q]            (XOR-swizzled)
//   pass 2: radix 16, NS = 16   butterfly j = t        in  [t + 64 r]        out  [256 (t >> 4) + (t & 15) + 16 q]
//   pass 3: radix 4,  NS = 256  butterflies t, 256 - t, 64 + t, 192 - t  (thread 0: 0, 128, 64, 192), in [j + 256 r],
//           whose outputs Z[j + 256 r] are exactly the mirrored pairs (k, NC - k) the packed-real split needs, so the
//           split happens in registers: slots 0..7 are Spec<1024>'s (bins t + 256 r and their mirrors; thread 0's four
//           self-mirrored pairs permuted by selects), slots 8..15 hold bins 64 + t + 256 r and their mirrors.
#pragma once
#include "gf3rx_device.h"

#define GF3_C16 0.92387953251128675613      /* cos(pi/8) */
#define GF3_S16 0.38268343236508977173      /* sin(pi/8) */

// compiler-level ordering of one wave's LDS phases (no instruction: the hardware keeps a wave's LDS operations in
// issue order; this keeps the compiler from moving them across the phase boundary)
GF3_DEV void wave_lds_fence() { asm volatile("" ::: "memory"); }

// radix-4 butterfly, natural order out: y[q] = sum_r x[r] (-i)^(r q)
GF3_DEV void bfly4(cplx& x0, cplx& x1, cplx& x2, cplx& x3) {
    const cplx s0 = cadd(x0, x2), s1 = csub(x0, x2), s2 = cadd(x1, x3), s3 = mul_negi(csub(x1, x3));
    x0 = cadd(s0, s2); x1 = cadd(s1, s3); x2 = csub(s0, s2); x3 = csub(s1, s3);
}
// the same on x1 w1, x2 w2, x3 w3 (twiddle multiplications fused into the first stage as complex fma, gf3rx_device.h)
GF3_DEV void bfly4_tw3(cplx& x0, cplx& x1, cplx& x2, cplx& x3, cplx w1, cplx w2, cplx w3) {
    const cplx s0 = cfma(x2, w2, x0), s1 = twice_minus(x0, s0);
    const cplx u1 = cmul(x1, w1);
    const cplx s2 = cfma(x3, w3, u1), s3 = mul_negi(twice_minus(u1, s2));
    x0 = cadd(s0, s2); x1 = cadd(s1, s3); x2 = csub(s0, s2); x3 = csub(s1, s3);
}
// ... with the constant twiddles W16^q, W16^2q, W16^3q of the second stage of a radix-16 butterfly, q = 1, 2, 3
GF3_DEV void bfly4_w16_1(cplx& x0, cplx& x1, cplx& x2, cplx& x3) {            // W^1, W^2, W^3
    const double p2 = x2.x + x2.y, m2 = x2.y - x2.x;                           // x2 W^2 = (p2, m2) / sqrt2
    const cplx s0 = cmk(fma(GF3_SQRT1_2, p2, x0.x), fma(GF3_SQRT1_2, m2, x0.y));
    const cplx s1 = cmk(fma(-GF3_SQRT1_2, p2, x0.x), fma(-GF3_SQRT1_2, m2, x0.y));
    const cplx u1 = cmul(x1, cmk(GF3_C16, -GF3_S16));
    const cplx s2 = cfma(x3, cmk(GF3_S16, -GF3_C16), u1), s3 = mul_negi(twice_minus(u1, s2));
    x0 = cadd(s0, s2); x1 = cadd(s1, s3); x2 = csub(s0, s2); x3 = csub(s1, s3);
}
GF3_DEV void bfly4_w16_2(cplx& x0, cplx& x1, cplx& x2, cplx& x3) {            // W^2, W^4 = -i, W^6
    const cplx t2 = mul_negi(x2);
    const cplx s0 = cadd(x0, t2), s1 = csub(x0, t2);
    const double p1 = x1.x + x1.y, m1 = x1.y - x1.x;                           // x1 W^2 = (p1, m1) / sqrt2
    const double p3 = x3.x + x3.y, m3 = x3.y - x3.x;                           // x3 W^6 = (m3, -p3) / sqrt2
    const double ax = p1 + m3, ay = m1 - p3;                                   // s2 = (ax, ay) / sqrt2
    const double bx = m1 + p3, by = m3 - p1;                                   // s3 = -i (x1 W^2 - x3 W^6) = (bx, by) / sqrt2
    x0 = cmk(fma(GF3_SQRT1_2, ax, s0.x), fma(GF3_SQRT1_2, ay, s0.y));
    x2 = cmk(fma(-GF3_SQRT1_2, ax, s0.x), fma(-GF3_SQRT1_2, ay, s0.y));
    x1 = cmk(fma(GF3_SQRT1_2, bx, s1.x), fma(GF3_SQRT1_2, by, s1.y));
    x3 = cmk(fma(-GF3_SQRT1_2, bx, s1.x), fma(-GF3_SQRT1_2, by, s1.y));
}
GF3_DEV void bfly4_w16_3(cplx& x0, cplx& x1, cplx& x2, cplx& x3) {            // W^3, W^6, W^9
    const double p2 = x2.x + x2.y, m2 = x2.y - x2.x;                           // x2 W^6 = (m2, -p2) / sqrt2
    const cplx s0 = cmk(fma(GF3_SQRT1_2, m2, x0.x), fma(-GF3_SQRT1_2, p2, x0.y));
    const cplx s1 = cmk(fma(-GF3_SQRT1_2, m2, x0.x), fma(GF3_SQRT1_2, p2, x0.y));
    const cplx u1 = cmul(x1, cmk(GF3_S16, -GF3_C16));
    const cplx s2 = cfma(x3, cmk(-GF3_C16, GF3_S16), u1), s3 = mul_negi(twice_minus(u1, s2));
    x0 = cadd(s0, s2); x1 = cadd(s1, s3); x2 = csub(s0, s2); x3 = csub(s1, s3);
}

// Radix-16 butterfly in place.  r = r1 + 4 r2, q = q2 + 4 q1:  W16^(r q) = W16^(r1 q2) W4^(r1 q1) W4^(r2 q2), so
//   stage 1: four radix-4 butterflies over r2 (inputs r1, r1+4, r1+8, r1+12), result u[r1][q2] at v[r1 + 4 q2];
//   stage 2: four radix-4 butterflies over r1 on u[.][q2] W16^(r1 q2), result y[q2 + 4 q1] at v[4 q2 + q1].
// The output is therefore TRANSPOSED in the register array: y[q] = v[F16_OUT(q)] (compile-time indices, no moves).
#define F16_OUT(q) (4 * ((q) & 3) + ((q) >> 2))
GF3_DEV void bfly16(cplx (&v)[16]) {
#pragma unroll
    for (int r1 = 0; r1 < 4; ++r1) bfly4(v[r1], v[r1 + 4], v[r1 + 8], v[r1 + 12]);
    bfly4(v[0], v[1], v[2], v[3]);
    bfly4_w16_1(v[4], v[5], v[6], v[7]);
    bfly4_w16_2(v[8], v[9], v[10], v[11]);
    bfly4_w16_3(v[12], v[13], v[14], v[15]);
}
// v[r] *= w^r, then the radix-16 butterfly.  The factor w^r1 common to a stage-1 group is deferred to stage 2, whose
// twiddles become (w W16^q2)^r1; stage 1 needs w^4, w^8, w^12 only.
GF3_DEV void bfly16_tw(cplx (&v)[16], cplx w) {
    const double c2 = w.x + w.x;
    const cplx w2 = cmk(fma(c2, w.x, -1.0), c2 * w.y);
    const cplx w3 = tw_next(c2, w2, w);
    const cplx w4 = cmk(fma(w2.x, w2.x, -(w2.y * w2.y)), (w2.x + w2.x) * w2.y);
    const cplx w8 = cmk(fma(w4.x, w4.x, -(w4.y * w4.y)), (w4.x + w4.x) * w4.y);
    const cplx w12 = cmul(w8, w4);
#pragma unroll
    for (int r1 = 0; r1 < 4; ++r1) bfly4_tw3(v[r1], v[r1 + 4], v[r1 + 8], v[r1 + 12], w4, w8, w12);
    // stage 2, q2 = 0: twiddles w, w^2, w^3
    bfly4_tw3(v[0], v[1], v[2], v[3], w, w2, w3);
    {   // q2 = 1: w W, w^2 W^2, w^3 W^3
        const cplx g1 = cmul(w, cmk(GF3_C16, -GF3_S16));
        const cplx g2 = cmk((w2.x + w2.y) * GF3_SQRT1_2, (w2.y - w2.x) * GF3_SQRT1_2);
        const cplx g3 = cmul(w3, cmk(GF3_S16, -GF3_C16));
        bfly4_tw3(v[4], v[5], v[6], v[7], g1, g2, g3);
    }
    {   // q2 = 2: w W^2, w^2 W^4, w^3 W^6
        const cplx g1 = cmk((w.x + w.y) * GF3_SQRT1_2, (w.y - w.x) * GF3_SQRT1_2);
        const cplx g2 = mul_negi(w2);
        const cplx g3 = cmk((w3.y - w3.x) * GF3_SQRT1_2, -(w3.x + w3.y) * GF3_SQRT1_2);
        bfly4_tw3(v[8], v[9], v[10], v[11], g1, g2, g3);
    }
    {   // q2 = 3: w W^3, w^2 W^6, w^3 W^9
        const cplx g1 = cmul(w, cmk(GF3_S16, -GF3_C16));
        const cplx g2 = cmk((w2.y - w2.x) * GF3_SQRT1_2, -(w2.x + w2.y) * GF3_SQRT1_2);
        const cplx g3 = cmul(w3, cmk(-GF3_C16, GF3_S16));
        bfly4_tw3(v[12], v[13], v[14], v[15], g1, g2, g3);
    }
}

// Per-thread twiddle bases of the 1024-point, 64-thread transform.  Three table entries are kept in registers; the
// other last-pass and split twiddles are those times constants (a handful of operations per transform instead of
// 16 registers held across the caller's loop).
struct F16Tw {
    static constexpr int NC = 1024, T = 64, Q = 256;
    cplx w2;                    // pass 2: exp(-2 pi i (t & 15) / 256)
    cplx wA;                    // pass 3, butterfly j = t: exp(-2 pi i t / 1024)
    cplx wb;                    // split: exp(-2 pi i t / 2048)
    GF3_DEV void init(int t, const cplx* __restrict__ tw, const cplx* __restrict__ twn) {
        w2 = tw[4 * (t & 15)]; wA = tw[t]; wb = twn[t];
    }
    // butterflies 256 - t (thread 0: 128), 64 + t, 192 - t:  -i conj(wA) (thread 0: W16^2),  wA W16,  W16^3 conj(wA)
    GF3_DEV cplx wB(int t) const { return t == 0 ? cmk(GF3_SQRT1_2, -GF3_SQRT1_2) : cmk(-wA.y, -wA.x); }
    GF3_DEV cplx wC() const { return cmul(wA, cmk(GF3_C16, -GF3_S16)); }
    GF3_DEV cplx wD() const { return cmul_conj(cmk(GF3_S16, -GF3_C16), wA); }
    // exp(-2 pi i (64 + t) / 2048) = wb exp(-i pi / 16)
    GF3_DEV cplx wb2() const { return cmul(wb, cmk(0.98078528040323044913, -0.19509032201612826785)); }
    // made opaque once per transform: otherwise LLVM hoists every twiddle power of every pass out of the caller's
    // loop over transforms and keeps them live across it (gf3rx_device.h, FftTw::refresh)
    GF3_DEV void refresh() {
        asm volatile("" : "+v"(w2.x), "+v"(w2.y), "+v"(wA.x), "+v"(wA.y), "+v"(wb.x), "+v"(wb.y));
    }
};

// Passes 1 and 2.  In: v[r] = z[t + 64 r].  Out: the pass-2 result in lds[0 .. 1024) in Stockham order, ready for the
// radix-4 last pass (inputs of butterfly j at j + 256 r).  One wave: no barrier anywhere.
GF3_DEV void f16_passes12(cplx (&v)[16], cplx* lds, const F16Tw& f, int t) {
    bfly16(v);
    {   // exchange 1, XOR-swizzled: logical i = 16 t + q  ->  i ^ ((i >> 4) & 15): conflict-free 16-byte stores and loads
        const int t16 = 16 * t + (t & 15);
#pragma unroll
        for (int q = 0; q < 16; ++q) lds[t16 ^ q] = v[F16_OUT(q)];
        wave_lds_fence();
        const int ts = t ^ (t >> 4);                          // logical t + 64 r  ->  64 r + (ts ^ 4 (r & 3))
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = lds[64 * r + (ts ^ (4 * (r & 3)))];
    }
    bfly16_tw(v, f.w2);
    {
        const int base = 256 * (t >> 4) + (t & 15);
#pragma unroll
        for (int q = 0; q < 16; ++q) lds[base + 16 * q] = v[F16_OUT(q)];
        wave_lds_fence();
    }
}

// Forward complex FFT of 1024 points, result Z[0 .. 1024) in lds in natural order (the inverse transform of the
// correlator: conj in, conj out).
GF3_DEV void f16_fft(cplx (&v)[16], cplx* lds, const F16Tw& f, int t) {
    f16_passes12(v, lds, f, t);
    const cplx w16_2 = cmk(GF3_SQRT1_2, -GF3_SQRT1_2);
#pragma unroll
    for (int b = 0; b < 4; ++b) {                             // butterflies j = t + 64 b: twiddle exp(-2 pi i j / 1024)
        const int j = t + 64 * b;
        cplx x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = lds[j + 256 * r];
        const cplx wbase = (b & 1) ? f.wC() : f.wA;
        const cplx w = (b & 2) ? cmul(wbase, w16_2) : wbase;
        bfly4_tw(x, w);
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[j + 256 * r] = x[r];  // (a butterfly reads and writes the same four places)
    }
    wave_lds_fence();
}

// Slot layout of the real transform's result (see the header comment): slot s = 2 p + h, pair p = 0..7,
// h = 0: bin k_p, h = 1: bin NC - k_p.
struct Spec16 {
    static constexpr int NC = 1024;
    GF3_DEV static int bin(int t, int s) {
        if (s < 8) return Spec<1024>::bin(t, s);
        const int k = 64 + t + 256 * ((s - 8) >> 1);
        return (s & 1) ? NC - k : k;
    }
    GF3_DEV static bool live(int t, int s) { return !(t == 0 && s == 1); }
};

// Real FFT of one packed 2048-sample segment, spectrum delivered in registers in Spec16 slot order, eight slots at a
// time so that a caller that consumes them at once (the correlator's multiply-accumulate) never holds all sixteen:
//   f16_passes12(v, lds, f, t);  rfft16_half<TWICE, 0>(o, ...) -> slots 0..7;  rfft16_half<TWICE, 1>(o, ...) -> slots 8..15
// In: v[r] = z[t + 64 r]; z0 (H = 0, thread 0) = Z[0] (DC and Nyquist packed).  TWICE: slots hold 2 X (real_split).
template <bool TWICE, int H>
GF3_DEV void rfft16_half(cplx (&o)[8], const cplx* lds, const F16Tw& f, int t, cplx& z0) {
    constexpr int Q = 256;
    if constexpr (H == 0) {   // pairs of butterflies t and 256 - t; thread 0's two butterflies (0 and 128) mirror onto themselves
        const int jB = (t == 0) ? Q / 2 : Q - t;
        cplx a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { a[r] = lds[t + Q * r]; b[r] = lds[jB + Q * r]; }
        bfly4_tw(a, f.wA);
        bfly4_tw(b, f.wB(t));
        z0 = a[0];
        const bool t0 = (t == 0);
        auto sel = [&](cplx x, cplx y) { return cmk(t0 ? x.x : y.x, t0 ? x.y : y.y); };
        const cplx A[4] = {sel(a[2], a[0]), a[1], sel(b[0], a[2]), sel(b[1], a[3])};
        const cplx Bm[4] = {sel(a[2], b[3]), sel(a[3], b[2]), sel(b[3], b[1]), sel(b[2], b[0])};
#pragma unroll
        for (int r = 0; r < 4; ++r) real_split<TWICE>(A[r], Bm[r], Spec<1024>::pair_tw(t, r, f.wb), o[2 * r], o[2 * r + 1]);
    } else {                  // butterflies 64 + t and 192 - t
        cplx c[4], d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { c[r] = lds[64 + t + Q * r]; d[r] = lds[192 - t + Q * r]; }
        bfly4_tw(c, f.wC());
        bfly4_tw(d, f.wD());
        const cplx wb2 = f.wb2();
#pragma unroll
        for (int r = 0; r < 4; ++r) real_split<TWICE>(c[r], d[3 - r], Spec<1024>::pair_tw(1, r, wb2), o[2 * r], o[2 * r + 1]);
    }
}
