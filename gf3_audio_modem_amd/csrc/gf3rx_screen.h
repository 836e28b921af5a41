// Stream-mode chirp sync, fast path: single-precision SCREENING of the matched filter + double-precision decisions.
//
// receiver.chirp_method (OFDM.py:356-372) needs, of the whole correlation P = convolve(r, chirp[::-1]):
//   (1) its global maximum M, (2) the lags i with P[i+1]/M > thresh that are local extrema, (3) the suppression walk.
// Only a handful of lags around every chirp ever pass (2).  So P is first evaluated in fp32 for EVERY lag (uniformly
// partitioned overlap-save, forward transforms fused into the accumulation: no spectra in HBM), together with a bound
// E_b on |P32 - P| for each output block.  With Mlo = max(P32 - E_b) <= M:
//   * a lag that could be the maximum has an upper bound P32 + E_b >= Mlo;
//   * a lag that could pass (2) has an upper bound >= thresh M (1 - 1e-6) >= thresh Mlo (1 - 1e-6).
// So every lag that matters has an upper bound >= min(Mlo, thresh Mlo (1 - 1e-6)), a level known before any fp64 value
// is.  The lags above it (in cells of 14, with the two neighbours the extremum test needs) are re-evaluated ONCE as fp64
// dot products with the fp64 replica; M is the largest of those values and the reference's rule is applied literally to
// the same values.
// Most blocks never get as far as P32: before the inverse transform the l1 norm of a block's accumulated spectrum bounds
// every one of its lags, and a block that stays under thresh x (the grid's running lower bound of M) is dropped there.
// No decision is ever taken on an fp32 value: fp32 only proves, with a margin, which lags need NOT be looked at.
// If the screen is not selective (constant streams, pathological thresholds: more cells than the work list holds)
// the caller falls back to the all-fp64 path (spec_kernel + ols_kernel), which is also what serves `d_corr`.
//
// The error bound.  Block b = sum over partitions q of the circular correlation of window b+q with partition q,
// each evaluated as irfft(rfft(x) conj(H_q)) in fp32.  For one term, with u = 2^-24:
//   |y32 - y|_inf <= |y32 - y|_2 <= g |x|_2 max_k|H_q[k]|,   g = (c_f + c_i) log2(N) u + c_m u
// (forward and inverse transform errors are relative in the 2-norm and the inverse transform is a contraction by
// 1/sqrt(N) after the forward one expanded by sqrt(N); products and the rounding of H_q and of fp64 samples to fp32
// are c_m u).  A 16-point butterfly pass with three-deep twiddle products is good for <= 8 u per pass, three passes
// plus the packed-real split per transform: g <~ 60 u.  The kernel uses GF3_SCR_GAMMA = 256 u; tests/ measure the
// realised ratio on random, DC-biased and adversarial streams (it stays below 2 u).
#pragma once
#include "gf3rx_device.h"
#include "gf3rx_screen_defs.h"

GF3_DEV cf cfmk(float a, float b) { return make_float2(a, b); }
GF3_DEV cf cfadd(cf a, cf b) { return cfmk(a.x + b.x, a.y + b.y); }
GF3_DEV cf cfsub(cf a, cf b) { return cfmk(a.x - b.x, a.y - b.y); }
GF3_DEV cf cfmul(cf a, cf b) { return cfmk(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)); }
GF3_DEV cf cfconj(cf a) { return cfmk(a.x, -a.y); }
GF3_DEV cf cf_negi(cf a) { return cfmk(a.y, -a.x); }
GF3_DEV cf cf_posi(cf a) { return cfmk(-a.y, a.x); }
GF3_DEV cf cf_fma_conj(cf a, cf h, cf c) {            // c + a conj(h)
    return cfmk(fmaf(a.x, h.x, fmaf(a.y, h.y, c.x)), fmaf(a.y, h.x, fmaf(-a.x, h.y, c.y)));
}

// Wave-wide sum / maximum of an fp32 value, the same in every lane: four DPP steps inside each row of 16 lanes
// (quad swaps, then the two row mirrors) and the four row results through scalar registers.  The butterfly of
// __shfl_xor it replaces is six ds_bpermute round trips through the LDS pipe, each waited for.
template <bool MAX>
GF3_DEV float scr_wave_reduce(float x) {
    auto step = [](float v, float o) { return MAX ? fmaxf(v, o) : v + o; };
    x = step(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true)));    // quad_perm [1,0,3,2]
    x = step(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true)));    // quad_perm [2,3,0,1]
    x = step(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, true)));   // row_half_mirror
    x = step(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xf, 0xf, true)));   // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
    return step(step(r0, r1), step(r2, r3));
}

// 4-point DFT in place: (a, b, c, d) = x0..x3  ->  X0..X3
GF3_DEV void scr_dft4(cf& a, cf& b, cf& c, cf& d) {
    const cf s0 = cfadd(a, c), s1 = cfsub(a, c), s2 = cfadd(b, d), s3 = cf_negi(cfsub(b, d));
    a = cfadd(s0, s2); c = cfsub(s0, s2); b = cfadd(s1, s3); d = cfsub(s1, s3);
}
// 16-point DFT in place as 4 x 4; output X[m] is left in v[scr_perm(m)]
GF3_DEV constexpr int scr_perm(int m) { return (m >> 2) + 4 * (m & 3); }
GF3_DEV void scr_dft16(cf (&v)[16]) {
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
#pragma unroll
    for (int a = 0; a < 4; ++a) scr_dft4(v[a], v[a + 4], v[a + 8], v[a + 12]);     // v[a + 4b] = T_a[b]
    // T_a[b] *= W16^(a b)
    v[1 + 4] = cfmul(v[1 + 4], cfmk(c1, -s1));  v[1 + 8] = cfmul(v[1 + 8], cfmk(h, -h));    v[1 + 12] = cfmul(v[1 + 12], cfmk(s1, -c1));
    v[2 + 4] = cfmul(v[2 + 4], cfmk(h, -h));    v[2 + 8] = cf_negi(v[2 + 8]);               v[2 + 12] = cfmul(v[2 + 12], cfmk(-h, -h));
    v[3 + 4] = cfmul(v[3 + 4], cfmk(s1, -c1));  v[3 + 8] = cfmul(v[3 + 8], cfmk(-h, -h));   v[3 + 12] = cfmul(v[3 + 12], cfmk(-c1, s1));
#pragma unroll
    for (int b = 0; b < 4; ++b) scr_dft4(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);   // v[c + 4b] = X[b + 4c]
}
// v[r] *= w^r, r = 1..15, products at most four deep
GF3_DEV void scr_twiddle16(cf (&v)[16], cf w) {
    const cf w2 = cfmul(w, w), w3 = cfmul(w2, w), w4 = cfmul(w2, w2);
    v[1] = cfmul(v[1], w); v[2] = cfmul(v[2], w2); v[3] = cfmul(v[3], w3); v[4] = cfmul(v[4], w4);
    const cf w5 = cfmul(w4, w), w6 = cfmul(w4, w2), w7 = cfmul(w4, w3), w8 = cfmul(w4, w4);
    v[5] = cfmul(v[5], w5); v[6] = cfmul(v[6], w6); v[7] = cfmul(v[7], w7); v[8] = cfmul(v[8], w8);
    v[9] = cfmul(v[9], cfmul(w8, w)); v[10] = cfmul(v[10], cfmul(w8, w2)); v[11] = cfmul(v[11], cfmul(w8, w3));
    v[12] = cfmul(v[12], cfmul(w8, w4)); v[13] = cfmul(v[13], cfmul(w8, w5)); v[14] = cfmul(v[14], cfmul(w8, w6));
    v[15] = cfmul(v[15], cfmul(w8, w7));
}

// Forward complex FFT of 4096 points, 256 threads x 16 points, three radix-16 Stockham passes.
// In: v[r] = z[t + 256 r].  Out: Z[t + 256 m] in v[scr_perm(m)].  P, Q: two 4096-point LDS buffers; the caller
// guarantees nobody still reads P when this starts storing into it, and may use P again after the call returns
// (every thread has passed the barrier that follows the stores into Q).
GF3_DEV void scr_fft4096(cf (&v)[16], cf* P, cf* Q, cf tw2, cf tw3, int t) {
    scr_dft16(v);
#pragma unroll
    for (int m = 0; m < 16; ++m) P[t * 16 + (m ^ (t & 15))] = v[scr_perm(m)];       // logical t*16 + m, XOR-swizzled
    lds_barrier();
    {
        const int ts = t ^ ((t >> 4) & 15);
        cf x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = P[ts + 256 * r];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = x[r];
    }
    scr_twiddle16(v, tw2);
    scr_dft16(v);
    {
        const int k = t & 15, base = (t - k) * 16 + k;
#pragma unroll
        for (int m = 0; m < 16; ++m) Q[base + 16 * m] = v[scr_perm(m)];
    }
    lds_barrier();
    {
        cf x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = Q[t + 256 * r];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = x[r];
    }
    scr_twiddle16(v, tw3);
    scr_dft16(v);
}


// one window: samples -> spectrum slots.  X[2r] = X[k_r], X[2r+1] = X[4096 - k_r], k_r = t + 256 r (thread 0, r = 0:
// both slots hold bin 2048); thread 0 also gets DC and Nyquist in z0 = (X[0], X[4096]).  Returns this thread's
// share of the window's energy (sum of squares of the 32 samples it loaded).
// (first part, shared with the band-limited kernel: samples -> complex transform, left in natural order in P; the
//  caller puts a barrier between this and its reads of P)
template <int DT>
GF3_DEV float scr_window_fft(const ScreenArgs& a, int64_t seg, cf (&v)[16], cf* P, cf* Q, cf tw2, cf tw3, int t) {
    typedef typename RawT<DT>::E E;
    // valid part of the window in window-relative sample numbers [lo, hi): 32-bit per-lane arithmetic from here on
    const int lo = seg >= 0 ? 0 : (seg <= -(int64_t)(2 * GF3_SCR_NC) ? 2 * GF3_SCR_NC : (int)(-seg));
    const int64_t rem = a.n_in - seg;
    const int hi = rem >= 2 * GF3_SCR_NC ? 2 * GF3_SCR_NC : (rem <= 0 ? 0 : (int)rem);
    float e2 = 0.0f;
    if (lo == 0 && hi == 2 * GF3_SCR_NC) {              // (uniform) the whole window lies inside the stream
        const E* base = (const E*)a.in + seg;
        const unsigned t2 = 2u * (unsigned)t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            RawPair<DT> raw;
            raw.load_u(base, t2 + 512u * (unsigned)r);
            v[r] = cfmk((float)raw.v.a, (float)raw.v.b);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 2 * (t + 256 * r);
            float x0 = 0.0f, x1 = 0.0f;
            if (j >= lo && j < hi) x0 = (float)((const E*)a.in)[seg + j];
            if (j + 1 >= lo && j + 1 < hi) x1 = (float)((const E*)a.in)[seg + j + 1];
            v[r] = cfmk(x0, x1);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) e2 = fmaf(v[r].x, v[r].x, fmaf(v[r].y, v[r].y, e2));
    scr_fft4096(v, P, Q, tw2, tw3, t);
    // exchange for the packed-real split: natural order into P (free: every thread is past the second barrier)
#pragma unroll
    for (int m = 0; m < 16; ++m) P[t + 256 * m] = v[scr_perm(m)];
    return e2;
}
template <int DT>
GF3_DEV float scr_window_spectrum(const ScreenArgs& a, int64_t seg, cf (&v)[16], cf& z0, cf* P, cf* Q, cf tw2, cf tw3, cf wb, int t) {
    const float e2 = scr_window_fft<DT>(a, seg, v, P, Q, tw2, tw3, t);
    lds_barrier();
    z0 = P[0];
    const float c32[8] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                          0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};
    const float s32[8] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                          0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const bool self = (t == 0 && r == 0);
        const int k = self ? GF3_SCR_NC / 2 : t + 256 * r;
        const cf A = P[k], Bm = P[GF3_SCR_NC - k];
        cf w = cfmul(wb, cfmk(c32[r], -s32[r]));                 // exp(-2 pi i (t + 256 r) / 8192)
        if (self) w = cfmk(0.0f, -1.0f);
        const cf Bc = cfconj(Bm);
        // (the halvings of E and D are dropped: the slots hold 2 X, undone -- exactly, a power of two -- by `inv` below)
        const cf Ee = cfadd(A, Bc), Dd = cfsub(A, Bc);
        const cf Ow = cfmul(cf_negi(Dd), w);
        v[2 * r] = cfadd(Ee, Ow);                                // 2 X[k]
        v[2 * r + 1] = cfconj(cfsub(Ee, Ow));                    // 2 X[4096 - k]
    }
    z0 = cfmk(2.0f * (z0.x + z0.y), 2.0f * (z0.x - z0.y));     // 2 X[0], 2 X[4096]
    return e2;
}

template <int DT>
__global__ __launch_bounds__(GF3_SCR_T, 2) void scr_ols_kernel(ScreenArgs a) {
    extern __shared__ double2 smem[];
    constexpr int NC = GF3_SCR_NC, T = GF3_SCR_T, B = GF3_SCR_B;
    cf* bufA = (cf*)smem;
    cf* bufB = bufA + NC;
    float* nrm = (float*)(bufB + NC);                 // [32][4] per-window, per-wave energy
    float* red = nrm + 128;                           // [B][4] per-block, per-wave maximum
    const int t = threadIdx.x, wave = t >> 6;
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 share one, each with its own L2),
    // and neighbouring groups of blocks share 5 of their 9 windows.  So the XCD that gets blockIdx % 8 == x walks its
    // own contiguous eighth of the stream: the windows a workgroup re-reads were fetched by its predecessor on the SAME
    // L2 a moment ago, instead of by another XCD.  (Placement is a speed matter only; any order is correct.)
    const int64_t b0 = (int64_t)B * xcd_order(blockIdx.x, gridDim.x);  // (the host pads the grid to a multiple of 8)
    if (b0 >= a.nblk) return;                                          // (uniform: padding workgroups)
    cf tw2 = a.tw[(t & 15) * 16], tw3 = a.tw[t], wb = a.twn[t];
    // Made opaque before every transform: otherwise LLVM hoists all 30 twiddle powers of the two passes and the
    // eight split twiddles out of the window loop and keeps ~80 registers of them live across it.
    auto refresh = [&]() {
        asm volatile("" : "+v"(tw2.x), "+v"(tw2.y), "+v"(tw3.x), "+v"(tw3.y), "+v"(wb.x), "+v"(wb.y));
    };
    cf acc[B][16];
    // (DC and Nyquist sums of the B blocks: thread 0's business only, so they live in LDS rather than in eight registers
    //  of every thread -- which the allocator spilled)
    float* dcn = red + 4 * B + (B + 1);               // [B][2]
    if (t < 2 * B) dcn[t] = 0.0f;
#pragma unroll
    for (int g = 0; g < B; ++g) {
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[g][s] = cfmk(0.0f, 0.0f);
    }
    const int nw = a.Q + B - 1;                       // windows b0 .. b0 + nw - 1
    for (int w = 0; w < nw; ++w) {
        cf v[16], z0;
        cf* P = (w & 1) ? bufB : bufA;
        cf* Qb = (w & 1) ? bufA : bufB;
        const int64_t seg = (b0 + w) * (int64_t)a.H - (a.Lc - 1);
        refresh();
        float e2 = scr_window_spectrum<DT>(a, seg, v, z0, P, Qb, tw2, tw3, wb, t);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) e2 += __shfl_xor(e2, d, 64);
        if ((t & 63) == 0) nrm[w * 4 + wave] = e2;
#pragma unroll
        for (int g = 0; g < B; ++g) {
            const int h = w - g;
            if (h >= 0 && h < a.Q) {
                const float4* Hp = a.Hs + ((int64_t)h * 8) * T;            // wave-uniform base + 32-bit lane offset
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float4 hh = Hp[(unsigned)(r * T + t)];
                    acc[g][2 * r] = cf_fma_conj(v[2 * r], cfmk(hh.x, hh.y), acc[g][2 * r]);
                    acc[g][2 * r + 1] = cf_fma_conj(v[2 * r + 1], cfmk(hh.z, hh.w), acc[g][2 * r + 1]);
                }
                if (t == 0) { dcn[2 * g] = fmaf(z0.x, a.H0N[2 * h], dcn[2 * g]); dcn[2 * g + 1] = fmaf(z0.y, a.H0N[2 * h + 1], dcn[2 * g + 1]); }
                asm volatile("" ::: "memory");        // one block's eight spectrum loads in flight at a time (32 registers, not 32 B)
            }
        }
    }
    // ---- per block: error bound, inverse real FFT of the Hermitian spectrum, maximum, store
    float* berr = red + 4 * B;                        // [B] the blocks' error bounds, [B]: the grid's running bound
    float* l1p = dcn + 2 * B;                         // [B][4] per-wave l1 norms of the blocks' accumulated spectra
    // |y[i]| <= (1/N) sum_k |Y_k|, before any inverse transform (see scr_ring_kernel): the slots hold 2 Y_k for the
    // one-sided bins 1 .. 4095 (thread 0's first pair holds bin 2048 twice: an over-estimate), thread 0 adds DC and Nyquist
#pragma unroll
    for (int g = 0; g < B; ++g) {
        float s1 = 0.0f;
#pragma unroll
        for (int s = 0; s < 16; ++s) s1 += sqrtf(fmaf(acc[g][s].x, acc[g][s].x, acc[g][s].y * acc[g][s].y));
        if (t == 0) s1 += fabsf(dcn[2 * g]) + fabsf(dcn[2 * g + 1]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s1 += __shfl_xor(s1, d, 64);
        if ((t & 63) == 0) l1p[g * 4 + wave] = s1;
    }
    const bool may_skip = a.run_lo != nullptr && a.thresh > 0.0f && a.thresh < 1.0f;
    lds_barrier();
    if (t == 0) berr[B] = may_skip ? __int_as_float(__atomic_load_n(a.run_lo, __ATOMIC_RELAXED)) : 0.0f;   // (one lane: see below)
    if (t < B) {
        float e = 0.0f;
        for (int q = 0; q < a.Q; ++q) {
            const float* n4 = nrm + (t + q) * 4;
            const float n2 = (n4[0] + n4[1]) + (n4[2] + n4[3]);
            e = fmaf(a.Hinf[q], sqrtf(n2 + GF3_SCR_UFLOW) * 1.0001f, e);
        }
        berr[t] = e * GF3_SCR_GAMMA * 1.0001f + 1e-37f;
    }
    const float inv = 0.25f / (float)NC;              // 1/NC of the inverse transform, 1/2 of each of the two splits
    const float c32[8] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                          0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};
    const float s32[8] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                          0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f};
#pragma unroll
    for (int g = 0; g < B; ++g) {
        const int64_t m0 = (b0 + g) * (int64_t)a.H;
        if (m0 < a.plen) {                             // (uniform)
            lds_barrier();                             // everyone is done with both buffers (and berr / l1p are in place)
            {   // a block that cannot hold the maximum or a candidate gets no inverse transform at all
                const float l1 = ((l1p[g * 4] + l1p[g * 4 + 1]) + (l1p[g * 4 + 2] + l1p[g * 4 + 3])) * (1.0001f / 8192.0f) + 5e-20f;
                const float be0 = berr[g], run0 = berr[B];
                if (may_skip && run0 > 0.0f && (l1 + be0) < a.thresh * run0 * (1.0f - 1e-6f) * 0.9999f) {   // (uniform)
                    if (t == 0) {
                        a.blk_max[b0 + g] = -INFINITY;
                        a.blk_err[b0 + g] = be0;
                        if (a.bad && !(be0 < INFINITY)) atomicOr(a.bad, 1ull);
                    }
                    continue;
                }
            }
            refresh();
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bool self = (t == 0 && r == 0);
                const int k = self ? NC / 2 : t + 256 * r;
                cf w = cfmul(wb, cfmk(c32[r], -s32[r]));
                if (self) w = cfmk(0.0f, -1.0f);
                const cf A = acc[g][2 * r], Bc = cfconj(acc[g][2 * r + 1]);
                const cf Ee = cfadd(A, Bc), Dd = cfsub(A, Bc);              // (halvings folded into `inv` as well)
                const cf Op = cfmul(Dd, cfconj(w));                         // * exp(+2 pi i k / 8192)
                const cf Zk = cfadd(Ee, cf_posi(Op));
                const cf Zm = cfadd(cfconj(Ee), cf_posi(cfconj(Op)));
                bufA[k] = cfconj(Zk);
                if (!self) bufA[NC - k] = cfconj(Zm);
            }
            if (t == 0) bufA[0] = cfmk(dcn[2 * g] + dcn[2 * g + 1], -(dcn[2 * g] - dcn[2 * g + 1]));   // conj(E + i Op), doubled like the rest
            lds_barrier();
            cf v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = bufA[t + 256 * r];
            scr_fft4096(v, bufB, bufA, tw2, tw3, t);  // (stores into bufA only after its first barrier: every thread has read its inputs by then)
            // z = conj(FFT(conj Z)) / NC ; y[2n] = Re z, y[2n+1] = Im z, n = t + 256 m
            const int64_t left = a.plen - m0;
            const int W = left < a.H ? (int)left : a.H;
            float mx = -INFINITY;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int i = 2 * (t + 256 * m);
                const cf z = v[scr_perm(m)];
                const float y0 = z.x * inv, y1 = -z.y * inv;
                if (i + 1 < W) mx = fmaxf(mx, fmaxf(y0, y1));
                else if (i < W) mx = fmaxf(mx, y0);
            }
            mx = scr_wave_reduce<true>(mx);
            if ((t & 63) == 0) red[g * 4 + wave] = mx;
            // (the shared bound was read by one lane above -- 256 lanes hammering one address would serialise the whole grid)
            lds_barrier();
            const float bmax = fmaxf(fmaxf(red[g * 4], red[g * 4 + 1]), fmaxf(red[g * 4 + 2], red[g * 4 + 3]));
            const float be = berr[g];
            const float run = berr[B];
            // A block whose upper bound stays below thresh x (a lower bound of the maximum already established by
            // any workgroup) can hold neither the maximum nor a candidate, whatever the final maximum turns out to
            // be (it can only be larger): its lags are never read again and need not be written.
            const bool skip = may_skip && run > 0.0f && (bmax + be) < a.thresh * run * (1.0f - 1e-6f) * 0.9999f;
            if (!skip) {
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    const int i = 2 * (t + 256 * m);
                    const cf z = v[scr_perm(m)];
                    const float y0 = z.x * inv, y1 = -z.y * inv;
                    if (i + 1 < W) *(float2*)(a.P32 + m0 + i) = make_float2(y0, y1);
                    else if (i < W) a.P32[m0 + i] = y0;
                }
            }
            if (t == 0) {
                a.blk_max[b0 + g] = bmax;
                a.blk_err[b0 + g] = be;
                const float lo = bmax - be;
                if (a.run_lo && lo > 0.0f && lo > run) atomicMax(a.run_lo, __float_as_int(lo));   // positive floats order like their bits; rare once a chirp has been seen
                if (a.bad && !(be < INFINITY)) atomicOr(a.bad, 1ull);
            }
        }
    }
}

// ---------------------------------------------------------------- band-limited screening kernel
// The reference's chirp sweeps 0 .. 8 kHz at fs = 48 kHz (OFDM.py:106-109): above bin 8192 * 8000 / 48000 = 1365 of an
// 8192-sample window the partitions' spectra hold about 1 % of their energy (the tails of the segment edges).
// Leaving the bins |k| >= 256 KS = 1536 out of the products changes a lag of one (window, partition) term by at most
//     |y_drop|_inf <= (1/N) sum_dropped |X[k]| |H_q[k]| <= |x_out|_2 |h_q,out|_2,   |.._out|_2^2 = (1/N) sum_dropped |..[k]|^2
// (Cauchy-Schwarz; the host evaluates |h_q,out|_2 in fp64 and rounds up; |x_out|_2, the 2-norm of the window's own
// dropped part, is summed in the kernel from the transform it has just computed -- the computed spectrum is within
// GF3_SCR_GAMMA |x|_2 of the true one, which is added), and this joins the rounding term in the block's error bound:
//     E_b = sum_q [ GAMMA (max|H_q| + |h_q,out|_2) |x_{b+q}|_2  +  |h_q,out|_2 |x_{b+q},out|_2 ].
// Next to a chirp the window is nearly all in-band and the second term is as small as the first.  Nothing else in the
// method changes, the bound stays rigorous and the decisions stay fp64.  What it buys: a block's accumulator is KS = 6 complex registers per thread instead of 16, so a
// RING of up to RQ = 8 unfinished blocks fits in registers and a workgroup walks R consecutive blocks with ONE forward
// transform per window -- 2 + (Q - 1) / R transforms per block where scr_ols_kernel, which has to re-transform the
// Q + B - 1 windows under its B = 4 blocks, needs 3.25 (Q = 6) -- and 6 instead of 16 multiply-adds per partition.
// Used when Q <= RQ and the dropped share is small (build_screen_plan); otherwise scr_ols_kernel.
template <int DT>
__global__ __launch_bounds__(GF3_SCR_T, 2) void scr_ring_kernel(ScreenArgs a) {
    extern __shared__ double2 smem[];
    constexpr int NC = GF3_SCR_NC, T = GF3_SCR_T, KS = GF3_SCR_KS, RQ = GF3_SCR_RQ;
    cf* bufA = (cf*)smem;
    cf* bufB = bufA + NC;
    float* nrm = (float*)(bufB + NC);                 // [16][4] energy of window j in row j & 15, per wave
    float* nro = nrm + 64;                            // [16][4] the same for the part of its spectrum that is dropped
    float* red = nro + 64;                            // [4] per-wave maximum of the finished block, [4] per-wave l1 norm of its spectrum
    float* bc = red + 8;                              // [2] the finished block's error bound; the grid's running bound as read for it; [8] + [8] the partitions' error coefficients
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);   // (in a scalar register for the once-per-block code)
    // (XCD-aware order as in scr_ols_kernel: each XCD walks its own contiguous eighth of the stream)
    const int64_t b0 = (int64_t)a.R * xcd_order(blockIdx.x, gridDim.x);
    if (b0 >= a.nblk) return;                                          // (uniform: padding workgroups)
    const int64_t b1 = b0 + a.R < a.nblk ? b0 + a.R : a.nblk;
    cf tw2 = a.tw[(t & 15) * 16], tw3 = a.tw[t], wb = a.twn[t];
    auto refresh = [&]() {                            // (see scr_ols_kernel: keeps the twiddle powers out of the loop-invariant set)
        asm volatile("" : "+v"(tw2.x), "+v"(tw2.y), "+v"(tw3.x), "+v"(tw3.y), "+v"(wb.x), "+v"(wb.y));
    };
    cf acc[RQ][KS];                                   // acc[i]: block j - (Q - 1) + i while window j is being added
#pragma unroll
    for (int i = 0; i < RQ; ++i)
#pragma unroll
        for (int r = 0; r < KS; ++r) acc[i][r] = cfmk(0.0f, 0.0f);
    const float inv = 0.25f / (float)NC;              // 1/NC of the inverse transform, 1/2 of each of the two splits
    const float c32[8] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                          0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};
    const float s32[8] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                          0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f};
    const bool may_skip = a.run_lo != nullptr && a.thresh > 0.0f && a.thresh < 1.0f;
    // (the partitions' error coefficients wait in LDS, and the few per-lane values of the once-per-block code below are
    //  recomputed there: kept in registers across the transforms they are what the allocator spills)
    if (t < 2 * GF3_SCR_RQ) bc[2 + t] = (t & 7) < a.Q ? a.ecoef[(t >> 3) * a.Q + (t & 7)] : 0.0f;
    for (int64_t j = b0; j < b1 + (a.Q - 1); ++j) {    // windows: block b is the sum over h of window b + h with partition h
        cf v[16];
        const int64_t seg = j * (int64_t)a.H - (a.Lc - 1);
        refresh();
        // (the grid's running bound is fetched here and used after the transform: any earlier value of a lower bound is
        //  a lower bound, and a load issued where it is needed would hold every wave at the barrier for a round trip to
        //  memory.  One lane reads it -- 256 lanes hammering one address would serialise the whole grid)
        float run_pre = 0.0f;
        if (may_skip && t == 0) run_pre = __int_as_float(__atomic_load_n(a.run_lo, __ATOMIC_RELAXED));
        float e2 = scr_window_fft<DT>(a, seg, v, bufA, bufB, tw2, tw3, t);
        // Energy of the DROPPED part of this window's spectrum (what the truncation term of the bound multiplies: next
        // to a chirp the window is nearly all in-band and this is 1e-3 of its energy).  One-sided bins 1536 .. 4096:
        // the pairs (k, 4096 - k), 1536 <= k <= 2048, as |Z_k|^2 + |Z_{4096-k}|^2 straight from the packed transform
        // (indices 1536 .. 2560: this thread's outputs 6 .. 9, and output 10 of thread 0) ...
        float eo = 0.0f;
#pragma unroll
        for (int m = 6; m < 10; ++m) { const cf z = v[scr_perm(m)]; eo = fmaf(z.x, z.x, fmaf(z.y, z.y, eo)); }
        { const cf z = v[scr_perm(10)]; const float g = t == 0 ? 1.0f : 0.0f; eo = fmaf(g * z.x, z.x, fmaf(g * z.y, z.y, eo)); }
        lds_barrier();
        // The partitions' spectra for this window's multiply-adds are asked for in two batches of four, into registers
        // the transform has just vacated: the first arrives under the split below, the second under the first batch's
        // multiply-adds.  Fetched one partition at a time next to its own multiply-adds they cost six L2 round trips in
        // a row: a third of the step (s_memtime stamps: 6.3 of 18.2 k cycles).
        auto fetch_h = [&](float4 (&hb)[RQ / 2][KS / 2], int i0) {
#pragma unroll
            for (int ii = 0; ii < RQ / 2; ++ii) {
                int h = a.Q - 1 - (i0 + ii);           // ring slot i holds block j - h
                asm volatile("" : "+s"(h));            // (kept scalar and out of the loop-invariant set: hoisted, the partitions'
                                                       //  per-lane 64-bit addresses are spilled around the transforms)
                const float4* Hp = a.Hb + (int64_t)(h < 0 ? 0 : h) * (KS / 2) * T;    // (slots from Q on: fetched, never used)
#pragma unroll
                for (int p = 0; p < KS / 2; ++p) hb[ii][p] = Hp[(unsigned)(p * T + t)];
            }
        };
        float4 hb0[RQ / 2][KS / 2], hb1[RQ / 2][KS / 2];
        fetch_h(hb0, 0);
        // packed-real split, kept bins only: X[r] = 2 X[t + 256 r]  (k = 0 pairs with itself: its slot is 2 X[0], real)
        cf X[KS];
#pragma unroll
        for (int r = 0; r < KS; ++r) {
            const int k = t + 256 * r;
            const cf A = bufA[k], Bc = cfconj(bufA[(NC - k) & (NC - 1)]);
            const cf w = cfmul(wb, cfmk(c32[r], -s32[r]));                // exp(-2 pi i k / 8192)
            const cf Ee = cfadd(A, Bc), Dd = cfsub(A, Bc);
            const cf Ow = cfmul(cf_negi(Dd), w);
            X[r] = cfadd(Ee, Ow);
            const cf Pn = cfsub(Ee, Ow);               // ... and the partners 4096 - k of the kept bins (k = 0: the Nyquist bin), doubled like X
            eo = fmaf(0.25f * Pn.x, Pn.x, fmaf(0.25f * Pn.y, Pn.y, eo));
        }
        asm volatile("" ::: "memory");                 // (the second batch is not to be hoisted over the split)
        fetch_h(hb1, RQ / 2);
        e2 = scr_wave_reduce<false>(e2);
        eo = scr_wave_reduce<false>(eo);
        if (lane == 0) { nrm[(int)(j & 15) * 4 + wave] = e2; nro[(int)(j & 15) * 4 + wave] = eo; }
        auto mac = [&](const float4 (&hb)[RQ / 2][KS / 2], int i0) {
#pragma unroll
            for (int ii = 0; ii < RQ / 2; ++ii)
                if (i0 + ii < a.Q) {                   // (uniform)
#pragma unroll
                    for (int p = 0; p < KS / 2; ++p) {
                        acc[i0 + ii][2 * p] = cf_fma_conj(X[2 * p], cfmk(hb[ii][p].x, hb[ii][p].y), acc[i0 + ii][2 * p]);
                        acc[i0 + ii][2 * p + 1] = cf_fma_conj(X[2 * p + 1], cfmk(hb[ii][p].z, hb[ii][p].w), acc[i0 + ii][2 * p + 1]);
                    }
                }
        };
        mac(hb0, 0);
        mac(hb1, RQ / 2);
        const int64_t b = j - (a.Q - 1);               // the block this window completes
        if (b >= b0) {                                 // (uniform; the first Q - 1 windows only fill the ring)
            // ---- Can the block matter at all?  |y[i]| <= (1/N) sum_k |Y_k| over the two-sided spectrum: the l1 norm of the
            // kept half, before any inverse transform.  A chirp partition lives in its own sixth of the band, so the
            // products of a window with the partitions it is NOT aligned with are small in every bin: off the two or
            // three blocks around a chirp's peak this bound is 4-7 % of the peak, well under the threshold, and the
            // inverse transform (40 % of the kernel's arithmetic) is not run for 17 of 20 blocks of a config-3 packet.
            {
                float s1 = 0.0f;
#pragma unroll
                for (int r = 0; r < KS; ++r) s1 += sqrtf(fmaf(acc[0][r].x, acc[0][r].x, acc[0][r].y * acc[0][r].y));
                s1 = scr_wave_reduce<false>(s1);
                if (lane == 0) red[4 + wave_s] = s1;
            }
            if (t == 0) bc[1] = run_pre;
            lds_barrier();                             // (also: every thread is done with the split's reads of bufA)
            // Error bound of block b from the energies of windows b .. b + Q - 1.  The rows of THIS step's window were
            // written by all four waves just above, so they are read after the barrier only -- and by every wave for
            // itself (lane & 7 takes window b + (lane & 7); three DPP steps add the eight terms), which needs no second
            // barrier to hand the result round.
            float be;
            {
                const int ql = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & 7;   // (= lane & 7, from the exec mask)
                const float* n4 = nrm + (((int)(b & 15) + ql) & 15) * 4;
                const float* o4 = n4 + 64;
                // |x|_2 of the window, and of its dropped part: |x_out|_2^2 <= (2 / 8192) x the one-sided sum
                const float nx = sqrtf((n4[0] + n4[1]) + (n4[2] + n4[3]) + GF3_SCR_UFLOW) * 1.0001f;
                const float no = sqrtf(((o4[0] + o4[1]) + (o4[2] + o4[3])) * (1.0f / 4096.0f) + GF3_SCR_UFLOW) * 1.0001f;
                // (lanes from Q on look at rows that may never have been written: their term is dropped, not multiplied by 0)
                float e = ql < a.Q ? fmaf(bc[2 + ql], nx, bc[10 + ql] * no) : 0.0f;
                e += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e), 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
                e += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e), 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
                e += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e), 0x141, 0xf, 0xf, true));    // row_half_mirror
                be = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(e))) * 1.0001f + 1e-37f;
            }
            const float run = bc[1];
            // acc holds 2 Y_k (the split's doubling), k >= 0 only: (1/8192) (|Y_0| + 2 sum_{k>0} |Y_k|) <= (1/8192) sum |acc_k|
            const float l1 = ((red[4] + red[5]) + (red[6] + red[7])) * (1.0001f / 8192.0f) + 5e-20f;   // (+: moduli whose squares underflowed, 1536 x 1.1e-19 at most)
            // (same rule as for the stores below: under thresh x an established lower bound of the maximum a block can hold
            //  neither the maximum nor a candidate, whatever the final maximum turns out to be)
            if (may_skip && run > 0.0f && (l1 + be) < a.thresh * run * (1.0f - 1e-6f) * 0.9999f) {     // (uniform)
                if (t == 0) {
                    a.blk_max[b] = -INFINITY;           // never listed, no part in the lower bound of the maximum
                    a.blk_err[b] = be;
                    if (a.bad && !(be < INFINITY)) atomicOr(a.bad, 1ull);
                }
            } else {
            // ---- inverse real FFT of the Hermitian spectrum whose kept half is acc[0]; every other bin is zero
            refresh();
            const int ti = launder(t);                 // (LDS addresses of this block-rate code are recomputed, not kept across the transforms)
#pragma unroll
            for (int r = 0; r < KS; ++r) {
                const int k = ti + 256 * r;
                const bool dc = (r == 0 && ti == 0);
                const cf w = cfmul(wb, cfmk(c32[r], -s32[r]));
                const cf A = acc[0][r];
                const cf Op = cfmul(A, cfconj(w));                          // * exp(+2 pi i k / 8192)
                const cf Zk = cfadd(A, cf_posi(Op));
                const cf Zm = cfadd(cfconj(A), cf_posi(cfconj(Op)));
                bufB[k] = cfconj(Zk);
                // (bin 4096 - k; k = 0 has no partner, and thread 0 uses the turn to clear the one bin of the dropped
                //  range [1536, 2560] that is read from LDS below rather than known to be zero)
                bufB[dc ? 2560 : NC - k] = dc ? cfmk(0.0f, 0.0f) : cfconj(Zm);
            }
            lds_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = (r >= KS && r < 16 - KS) ? cfmk(0.0f, 0.0f) : bufB[t + 256 * r];
            scr_fft4096(v, bufA, bufB, tw2, tw3, t);
            // z = conj(FFT(conj Z)) / NC ; y[2n] = Re z, y[2n+1] = Im z, n = t + 256 m; the block's lags are n < H / 2 <= 2048
            const int64_t m0 = b * (int64_t)a.H;
            const int64_t left = a.plen - m0;
            const int W = left < a.H ? (int)left : a.H;
            float mx = -INFINITY;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int i = 2 * (t + 256 * m);
                const cf z = v[scr_perm(m)];
                const float y0 = z.x * inv, y1 = -z.y * inv;
                if (i + 1 < W) mx = fmaxf(mx, fmaxf(y0, y1));
                else if (i < W) mx = fmaxf(mx, y0);
            }
            mx = scr_wave_reduce<true>(mx);
            if (lane == 0) red[wave_s] = mx;
            lds_barrier();
            const float bmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            // (same rule as scr_ols_kernel: a block whose upper bound stays below thresh x an established lower bound
            //  of the maximum can hold neither the maximum nor a candidate; its lags are never read again)
            const bool skip = may_skip && run > 0.0f && (bmax + be) < a.thresh * run * (1.0f - 1e-6f) * 0.9999f;
            if (!skip) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int i = 2 * (t + 256 * m);
                    const cf z = v[scr_perm(m)];
                    const float y0 = z.x * inv, y1 = -z.y * inv;
                    if (i + 1 < W) *(float2*)(a.P32 + m0 + i) = make_float2(y0, y1);
                    else if (i < W) a.P32[m0 + i] = y0;
                }
            }
            if (t == 0) {
                a.blk_max[b] = bmax;
                a.blk_err[b] = be;
                const float lo = bmax - be;
                if (a.run_lo && lo > 0.0f && lo > run) atomicMax(a.run_lo, __float_as_int(lo));
                if (a.bad && !(be < INFINITY)) atomicOr(a.bad, 1ull);
            }
            }
        } else {
            lds_barrier();                             // (the next window's first stores go to bufA, which the split above reads)
        }
#pragma unroll
        for (int i = 0; i + 1 < RQ; ++i)
#pragma unroll
            for (int r = 0; r < KS; ++r) acc[i][r] = acc[i + 1][r];
#pragma unroll
        for (int r = 0; r < KS; ++r) acc[RQ - 1][r] = cfmk(0.0f, 0.0f);
    }
}

// fp64 re-evaluation of the 16 lags of one cell: P[m] = sum_k r[m - Lc + 1 + k] c[k], m = 14 c + j, j = 0..15.
// Each thread takes a contiguous range of taps and slides a 31-sample window over them, 16 taps at a time
// (256 fma per 31 sample loads + 16 tap loads); the 16 partial sums are then reduced over the workgroup.
// Every WAVE works on its own: it walks its cells' taps 1024 at a time and owns a private slice of LDS for the samples, so
// there is no workgroup barrier anywhere.  What the first version of this kernel (rounds 1-2: samples AND taps staged
// through LDS as doubles, 57 LDS instructions per step; cells drawn from a work counter) was bound by was measured in
// round 3 with s_memtime stamps per phase, timing-only ablations and SQ_LDS_* counters: neither
// the LDS array nor the HBM latency, but (i) the NUMBER of LDS and vector-memory instructions a wave has to get through
// beside its 256 fma per step -- at two waves per SIMD an LDS instruction of any width costs a wave ~30 cycles -- and
// (ii) two atomic additions per wave on one address at the start of the kernel: 4 096 of them, served one after the
// other, were 40 of its 210 us.  This version
//   * stages the samples RAW (4 bytes each; 8 for an f64 stream): a lane loads 16 consecutive bytes of the stream
//     (coalesced, 1 KB per wave-instruction), stores them with one ds_write_b128 and reads its own 31-sample window back
//     with 8 ds_read_b128 (16 for f64), widening to double in registers.  A lane's 16 elements are followed by one
//     16-byte pad, so a lane's run is 80 (144) bytes: runs stay 16-byte aligned and the reads are conflict-free under
//     the b128 lane groups;
//   * keeps the taps out of LDS: they are the same for every cell, so the context holds them tiled the way the lanes
//     consume them (RefineArgs::chirp_t) and a lane's 16 taps of a step are 8 coalesced 16-byte loads into registers;
//   * deals the cells to the waves by a fixed stride (every cell costs the same) and keeps everything about a wave's
//     position in scalar registers, a cell's number fetched by a scalar load;
//   * reduces the 16 partial sums over the wave by halving: each level exchanges only the half a lane does not keep
//     (8 + 4 + 2 + 1 + 1 + 1 = 17 exchanges instead of 96).  The pairing of every addition is that of the butterfly it
//     replaces and the order of a lane's own fma is unchanged, so the values are bit-for-bit those of the first version.
// 13 (25) LDS instructions per step instead of 57, 42 per cell's reduction instead of 192; 211 -> 150 us on the config-3
// stream (5 267 cells), of which ~35 us are the 13 KB a wave pulls through the vector-memory pipe per step.
template <int DT>
__global__ __launch_bounds__(SCR_REF_THREADS, 2) void scr_refine_kernel(RefineArgs a) {
    constexpr int WT = SCR_REF_WT, NW = SCR_REF_THREADS / 64;
    typedef typename RawT<DT>::E E;
    typedef typename ScrStage<DT>::S S;                                   // what LDS holds (exact for every sample type)
    constexpr int PER = 16 / (int)sizeof(S);                              // elements per 16-byte access: 4 (2)
    constexpr int GPR = 16 / PER;                                         // 16-byte groups per lane run: 4 (8)
    constexpr int NG = (WT + 16) / PER;                                   // groups staged per step: 260 (520)
    constexpr int NQ = (NG + 63) / 64;                                    // stores per lane and step: 5 (9)
    constexpr int NR = (31 + PER - 1) / PER;                              // reads per lane and step: 8 (16)
    typedef S SV __attribute__((ext_vector_type(PER)));
    typedef double D2 __attribute__((ext_vector_type(2)));
    struct __attribute__((packed, aligned(sizeof(E)))) RawV { E v[PER]; };  // PER consecutive samples, aligned like one
    constexpr int SPARE = (WT + 16) / 16 * (GPR + 1);                    // 64 slots behind the runs for the lanes the last store has no group for
    __shared__ SV xs_all[NW][SPARE + 64];
    if (a.misc->status & 1) return;
    const int ncell = __builtin_amdgcn_readfirstlane((int)a.misc->ncell);   // (<= the capacity of the list, far below 2^31)
    const int lane = threadIdx.x & 63;
    SV* xs = xs_all[threadIdx.x >> 6];
    const int nst = (a.Lc + WT - 1) / WT;             // steps per cell
    // A wave's work is a stream of items (cell, step); three of them are in flight: `c` is being computed out of LDS,
    // `w` has its samples in registers and its taps on the way, `f` is being fetched.  All three are wave-uniform.
    // The cells are dealt to the waves by a fixed stride -- every cell costs the same.  (Drawing them from a counter, as
    // the first version did, opens the kernel with two atomic additions per wave on one address: 4 096 of them, served
    // one after the other, were 40 of the kernel's 190 us.)  With two workgroups per CU the stride also leaves the
    // cells of the last, partial round to the first workgroup of every CU, one wave per SIMD.
    // Everything about an item lives in scalar registers, and a cell's number is fetched with a scalar load: as a vector
    // load it made the compiler wait for ALL the wave's vector loads (vmcnt(0)) -- the prefetches of the next steps -- at
    // every step.
    struct Item { int cur; int64_t cell; int st; };
    const int nwaves = (int)gridDim.x * NW;
    auto cell_at = [&](int i) -> int64_t {             // a.cells[i], i wave-uniform
        long long v;
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(a.cells + i) : "memory");
        return v;
    };
    auto advance = [&](const Item& it) -> Item {       // the item after `it` (it.cur < ncell)
        if (it.st + 1 < nst) return Item{it.cur, it.cell, it.st + 1};
        const int nc = it.cur + nwaves;
        return Item{nc, nc < ncell ? cell_at(nc) : 0, 0};
    };
    RawV xr[NQ];
    D2 tn[8];
    const unsigned lane16 = 16u * lane;                // (a wave-uniform base plus this: the scalar-base form of global_load)
    auto fetch = [&](const Item& it) {                 // the item's samples -> xr
        const int64_t i0 = GF3_SCR_CELL * it.cell - (a.Lc - 1) + (int64_t)it.st * WT;
        if (i0 >= 0 && i0 + WT + 16 <= a.n_in) {        // (uniform) the whole step lies inside the stream
            const E* base = (const E*)a.in + i0 + PER * lane;   // (one address per lane, the rest are immediate offsets)
#pragma unroll
            for (int q = 0; q < NQ - 1; ++q) xr[q] = *(const RawV*)(base + 64 * PER * q);
            const int lt = launder(lane);
            xr[NQ - 1] = *(const RawV*)((const E*)a.in + i0 + PER * (64 * (NQ - 1) + (lt < NG - 64 * (NQ - 1) ? lt : 0)));
        } else {
            const int64_t last_i = a.n_in - 1;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int t = 0; t < PER; ++t) {
                    const int64_t i = i0 + PER * (lane + 64 * q) + t;
                    const int64_t ic = i < 0 ? 0 : (i > last_i ? last_i : i);
                    const E val = ((const E*)a.in)[ic];
                    xr[q].v[t] = (i == ic) ? val : (E)0;
                }
            }
        }
    };
    auto fetch_taps = [&](int st) {
#pragma unroll
        for (int q = 0; q < 8; ++q) tn[q] = *(const D2*)((const char*)a.chirp_t + (size_t)st * (8 * 1024) + lane16 + 1024 * q);
    };
    auto stage = [&]() {                               // xr -> this wave's LDS (LDS serves a wave's instructions in order:
#pragma unroll                                         //  the reads of the step before were issued ahead of these stores)
        for (int q = 0; q < NQ; ++q) {
            const int g = lane + 64 * q;
            SV v;
#pragma unroll
            for (int t = 0; t < PER; ++t) v[t] = (S)xr[q].v[t];
            xs[(q < NQ - 1 || g < NG) ? g + g / GPR : SPARE + lane] = v;   // (a store by every lane: no branch in the step)
        }
    };
    Item c, w, f;
    {
        const int first = (int)blockIdx.x * NW + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (first >= ncell) return;                    // (per wave: nothing below synchronises across waves)
        c = Item{first, cell_at(first), 0};
        fetch(c); fetch_taps(0);
        w = advance(c);
        stage();
        if (w.cur < ncell) fetch(w);
        f = w.cur < ncell ? advance(w) : w;
    }
    double acc[16];
#ifdef GF3_STAMPS
    unsigned long long tk0 = 0, tk3 = 0, tk4 = 0, t_first = 0, s_body = 0, s_tail = 0, s_next = 0, n_steps = 0;
    SCR_TICK(t_first);
    unsigned long long rt0, rt1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0) :: "memory");
#endif
    while (true) {
#ifdef GF3_STAMPS
        SCR_TICK(tk0);
#endif
        if (c.st == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.0;
        }
        // this step's window out of LDS, its taps out of the prefetch registers; then the next item's samples to LDS, its
        // taps and the samples of the item after it requested -- none of that is waited for in this step.  (Spreading these
        // 5 stores and 13 loads over the fma with sched_group_barrier, as one basic block, changed nothing: 150.9 vs 151.7 us.)
        double x[NR * PER];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const SV v = xs[(GPR + 1) * lane + r + r / GPR];
#pragma unroll
            for (int t = 0; t < PER; ++t) x[PER * r + t] = (double)v[t];
        }
        D2 tc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) tc[q] = tn[q];
        if (w.cur < ncell) { stage(); fetch_taps(w.st); }
        if (f.cur < ncell) fetch(f);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const double ck = tc[kk >> 1][kk & 1];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = fma(ck, x[kk + j], acc[j]);
        }
#ifdef GF3_STAMPS
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(acc[j]));
        SCR_TICK(tk3);
#endif
        if (c.st == nst - 1) {
            // ---- the cell is complete: sum the 16 partial sums over the wave.  Level d leaves a lane with the half of its
            // values selected by its bit d and adds the partner's copy of that half; after d = 32, 16, 8, 4 a lane holds
            // the one sum j = (lane >> 2) & 15, which d = 2, 1 complete.
            const int64_t m0 = GF3_SCR_CELL * c.cell;
            double s;
            {
                double v8[8], v4[4], v2[2];
                const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
                for (int t = 0; t < 8; ++t) { const double give = b5 ? acc[t] : acc[t + 8], keep = b5 ? acc[t + 8] : acc[t]; v8[t] = keep + __shfl_xor(give, 32, 64); }
#pragma unroll
                for (int t = 0; t < 4; ++t) { const double give = b4 ? v8[t] : v8[t + 4], keep = b4 ? v8[t + 4] : v8[t]; v4[t] = keep + __shfl_xor(give, 16, 64); }
#pragma unroll
                for (int t = 0; t < 2; ++t) { const double give = b3 ? v4[t] : v4[t + 2], keep = b3 ? v4[t + 2] : v4[t]; v2[t] = keep + __shfl_xor(give, 8, 64); }
                { const double give = b2 ? v2[0] : v2[1], keep = b2 ? v2[1] : v2[0]; s = keep + __shfl_xor(give, 4, 64); }
                s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 1, 64);
            }
            const int j = (lane >> 2) & 15;
            const bool counts = m0 + j < a.plen;
            if ((lane & 3) == 0) a.cell_val[(int64_t)c.cur * 16 + j] = s;
            const bool nan = __ballot(counts && !(s == s)) != 0ull;
            double mx = (counts && s == s) ? s : -INFINITY;
#pragma unroll
            for (int d = 32; d >= 4; d >>= 1) mx = fmax(mx, __shfl_xor(mx, d, 64));
            if (lane == 0) {
                if (nan) atomicOr(&a.misc->m_nan, 1u);
                else if (mx > -INFINITY) atomicMax(&a.misc->m_key, scr_key(mx));
            }
        }
#ifdef GF3_STAMPS
        SCR_TICK(tk4);
        s_body += tk3 - tk0; s_tail += tk4 - tk3; ++n_steps;
#endif
        if (w.cur >= ncell) break;                     // (uniform)
        c = w; w = f;
        if (f.cur < ncell) f = advance(f);
#ifdef GF3_STAMPS
        SCR_TICK(tk0);
        s_next += tk0 - tk4;
#endif
    }
#ifdef GF3_STAMPS
    if (a.stamps && lane == 0) {
        unsigned long long* o = a.stamps + ((size_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 8;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1) :: "memory");
        o[0] = tk4 - t_first; o[1] = rt0; o[2] = rt1; o[3] = s_body; o[4] = s_tail; o[5] = s_next; o[6] = 0; o[7] = n_steps;
    }
#endif
}

// The same re-evaluation on the fp64 matrix cores.  With taps k = 16 a + b and s = j + b, the 16 lags of a cell are the
// anti-diagonal sums  P_j = sum_b G[j + b][b]  of  G[s][b] = sum_a x[16 a + s] c[16 a + b]  (s < 31), a 31 x 16 x (Lc / 16)
// matrix product: twice the multiply-adds of the dot products, but v_mfma_f64_16x16x4 takes one coalesced operand per lane
// per 1024 of them (lane l: x[64 kb + l] resp. x[64 kb + 16 + l] for the two row tiles, c[64 kb + l] for both), so there is
// no staging through LDS, no per-lane sliding window and nothing to wait for but the loads.  One wave per cell; the sums
// of a product are accumulated in a fixed order, so the values are reproducible.
#ifndef GF3_REFINE_MFMA
#define GF3_REFINE_MFMA 0
#endif

typedef double scr_d4 __attribute__((ext_vector_type(4)));
template <int DT>
__global__ __launch_bounds__(SCR_REF_THREADS) void scr_refine_mfma_kernel(RefineArgs a) {
    __shared__ double Gs[SCR_REF_THREADS / 64][32 * 17];
    typedef typename RawT<DT>::E E;
    if (a.misc->status & 1) return;
    // The cells are dealt to the waves by a fixed stride (every cell costs the same), so the loop is a counted loop over
    // wave-uniform scalars.  The first version drew cells from the work counter inside a `while (true)` and compared the
    // drawn number with an `ncell` the compiler kept in a vector register: to the compiler the loop exit was a per-lane
    // decision, it built the loop out of EXEC masks, and once lane 0 had been masked off `readfirstlane` returned the
    // `drawn` of a lane that never drew -- zero -- so the wave re-evaluated cell 0 for ever.  That is the run that did
    // not return at the end of round 2 (and again, under a 120 s limit, in round 3 before this change).
    auto first_lane64 = [](unsigned long long v) -> long long {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (long long)(((unsigned long long)hi << 32) | lo);
    };
    const long long ncell = first_lane64((unsigned long long)a.misc->ncell);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* G = Gs[wave];
    const int nkb = (a.Lc + 63) / 64;                 // blocks of 64 taps
    const E* xin = (const E*)a.in;
    const long long nwaves = (long long)gridDim.x * (SCR_REF_THREADS / 64);
    for (long long cur = (long long)blockIdx.x * (SCR_REF_THREADS / 64) + wave; cur < ncell; cur += nwaves) {
        const int64_t c = (int64_t)first_lane64((unsigned long long)a.cells[cur]);
        const int64_t i0 = GF3_SCR_CELL * c - (a.Lc - 1);
        scr_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        const bool inside = i0 >= 0 && i0 + (int64_t)64 * nkb + 16 <= a.n_in;      // (uniform)
        if (inside) {
            const E* xl = xin + i0 + lane;
#pragma unroll 4
            for (int kb = 0; kb < nkb; ++kb) {
                const int k = 64 * kb + lane;
                const double xa = (double)xl[64 * kb], xb = (double)xl[64 * kb + 16];
                const double cv = a.chirp[k < a.Lc ? k : a.Lc - 1];
                const double cb = k < a.Lc ? cv : 0.0;
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, cb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xb, cb, acc1, 0, 0, 0);
            }
        } else {
            const int64_t last_i = a.n_in - 1;
            for (int kb = 0; kb < nkb; ++kb) {
                const int k = 64 * kb + lane;
                const int64_t ia = i0 + k, ib = ia + 16;
                const int64_t ca = ia < 0 ? 0 : (ia > last_i ? last_i : ia), cb2 = ib < 0 ? 0 : (ib > last_i ? last_i : ib);
                const double va = (double)xin[ca], vb = (double)xin[cb2];
                const double xa = ia == ca ? va : 0.0, xb = ib == cb2 ? vb : 0.0;
                const double cv = a.chirp[k < a.Lc ? k : a.Lc - 1];
                const double cb = k < a.Lc ? cv : 0.0;
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, cb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xb, cb, acc1, 0, 0, 0);
            }
        }
        // D[row][col]: lane holds rows (lane >> 4) + 4 r, r = 0..3, of column lane & 15
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) + 4 * r, col = lane & 15;
            G[row * 17 + col] = acc0[r];
            G[(16 + row) * 17 + col] = acc1[r];
        }
        asm volatile("" ::: "memory");                 // (one wave: LDS serves its instructions in order)
        double Pj = 0.0;
        if (lane < 16) {
#pragma unroll
            for (int b = 0; b < 16; ++b) Pj += G[(lane + b) * 17 + b];
            a.cell_val[cur * 16 + lane] = Pj;
        }
        asm volatile("" ::: "memory");
        const int64_t m0 = GF3_SCR_CELL * c;
        const bool valid = lane < 16 && m0 + lane < a.plen;
        const unsigned long long nanb = __ballot(valid && !(Pj == Pj));
        double mx = valid ? Pj : -INFINITY;
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) mx = fmax(mx, __shfl_xor(mx, d, 64));
        if (lane == 0) {
            if (nanb) atomicOr(&a.misc->m_nan, 1u);
            else if (mx > -INFINITY) atomicMax(&a.misc->m_key, scr_key(mx));
        }
    }
}

