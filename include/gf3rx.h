/*
 * gf3rx.h -- C ABI of the MI355X-native OFDM receive-path engine (libgf3rx.so).
 *
 * This is the drop-in boundary for the demodulation path of the GF3 audio
 * modem.  The reference has no FFI of its own (it is one Python file); the
 * boundary is the set of methods of class `receiver` in /root/reference/OFDM.py
 * that `receiver.receive` (OFDM.py:581-657) strings together.  Each entry
 * point below names the reference method(s) it replaces.  INTEGRATION.md shows
 * the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no exceptions; every call returns 0 (GF3_OK) or a negative
 *     gf3_status; gf3_last_error(ctx) holds a message for the last failure.
 *   - every `d_*` pointer is DEVICE memory owned by the caller (e.g. a
 *     torch tensor's data_ptr()); the library owns only the context and its
 *     look-up tables.  `stream` is a hipStream_t passed as void* (NULL = the
 *     default stream).  All work is enqueued asynchronously on that stream;
 *     only gf3_sync_stream synchronises (it returns a count to the host).
 *   - a context is immutable after creation: no call writes into it, so
 *     concurrent calls on one context from different host threads / on
 *     different streams are safe (every call brings its own workspace and
 *     outputs; error text and diagnostics are kept per calling THREAD, not in
 *     the context, and so are the 256 bytes of pinned host memory the calls
 *     that return counts read them back into).  The one exception is the legacy convenience
 *     gf3_sync_stream_mode, which stores a default mode for the legacy entry
 *     point gf3_sync_stream; gf3_sync_stream_ex takes the mode per call.
 *   - complex128 arrays are interleaved (re, im) doubles, as NumPy stores them.
 */
#ifndef GF3RX_H
#define GF3RX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GF3RX_VERSION "0.1.0"

typedef enum {
    GF3_OK = 0,
    GF3_EINVAL = -1,      /* bad argument / unsupported geometry            */
    GF3_EHIP = -2,        /* a HIP runtime call failed                      */
    GF3_ENOMEM = -3,
    GF3_ERANGE = -4,      /* capacity of an output buffer exceeded          */
    GF3_ENODETECT = -5    /* sync found fewer than two chirps (the reference
                             dies in np.vstack([]) here, OFDM.py:400)       */
} gf3_status;

/* storage type of the input sample stream (samples are widened to fp64 in
 * registers; all arithmetic is fp64) */
typedef enum { GF3_F64 = 0, GF3_F32 = 1, GF3_I16 = 2, GF3_U8 = 3 } gf3_dtype;

/*
 * Engine configuration == the attributes CamG.__init__ sets (OFDM.py:18-101),
 * generalised: any N in {1024,2048,4096,8192}, any CP, any constellation.
 */
typedef struct {
    int32_t N;                  /* ofdm_symbol_size            OFDM.py:27   */
    int32_t CP;                 /* cp_length                   OFDM.py:42   */
    int32_t P;                  /* no_pilots (each side), >=1  OFDM.py:51   */
    int32_t D;                  /* packet_length               OFDM.py:50   */
    int32_t Lc;                 /* chirp_length; 0 => 5*(N+CP) OFDM.py:64   */
    double  fs, f0, f1;         /* 48000, 0, 8000              OFDM.py:24,62-63 */
    double  thresh;             /* 0.4                         OFDM.py:361  */
    int32_t fit_lo, fit_hi;     /* 500, 1000 (python slice)    OFDM.py:462  */
    int32_t mu;                 /* bits per constellation point             */
    int32_t M;                  /* constellation size (<= 64)               */
    const double  *const_re;    /* [M] points in mapping_table insertion    */
    const double  *const_im;    /*     order                   OFDM.py:72-77 */
    const uint8_t *const_bits;  /* [M*mu] bit labels, tuple order           */
    const double  *known_re;    /* [K] map(known_sequence[:K*mu]) OFDM.py:429 */
    const double  *known_im;    /*     K = N/2-1                            */
    const int32_t *data_bins;   /* [C] data_carriers (FFT bin numbers 1..K),
                                   output bit order follows this array  OFDM.py:47,603 */
    int32_t C;
    int32_t in_dtype;           /* gf3_dtype of the sample stream           */
    int32_t max_window;         /* largest search window (lags) gf3_sync_frames
                                   will be asked for; 0 => 512               */
} gf3_config;

typedef struct gf3_ctx gf3_ctx;

const char *gf3_version(void);
/* SHA-256 (hex) of the sources + compiler flags this binary was built from, "unknown" for a build made by hand */
const char *gf3_source_hash(void);

/* replaces CamG.__init__ + sync_chirp (OFDM.py:18-109): builds twiddles, the
 * chirp replica and its partition spectra, demap tables, on the current device */
int gf3_ctx_create(const gf3_config *cfg, gf3_ctx **out);
void gf3_ctx_destroy(gf3_ctx *ctx);
/* message of the CALLING THREAD's last failed call (the argument is ignored: nothing is stored in a context) */
const char *gf3_last_error(const gf3_ctx *ctx);
/* Returns and CLEARS the HIP runtime's sticky last error of the calling thread (0 = none).  For callers that made a HIP
 * call of their own that may fail by design -- e.g. pinning a read-only mapping with hipHostRegister before handing it
 * to the ingest path -- so that the refusal is not reported by the next unrelated launch check. */
int gf3_clear_runtime_error(void);

/* diagnostics: in a library built with -DGF3_STAMPS, gf3_demod_frames writes eight s_memtime
 * stamps per frame into this device buffer ([F][8] uint64); a no-op in the product build */
void gf3_debug_set_stamps(gf3_ctx *ctx, void *d_buf_u64);

/* derived sizes a caller needs to allocate outputs */
int32_t gf3_bytes_per_frame(const gf3_ctx *ctx);     /* ceil(D*C*mu/8)            */
int32_t gf3_sync_max_window(const gf3_ctx *ctx);     /* lags one gf3_sync_frames block resolves */
int64_t gf3_sync_stream_workspace_bytes(const gf3_ctx *ctx, int64_t n);

/* chirp replica, Lc doubles, copied to HOST memory (sync_chirp, OFDM.py:106-109) */
int gf3_chirp_replica(const gf3_ctx *ctx, double *h_out);

/*
 * remove_cp + np.fft.fft (OFDM.py:407-408, 593) on n_sym independent symbols.
 * d_offsets[i] = index (in samples, into d_in) of the first of the N samples of
 * symbol i (i.e. already past its cyclic prefix).  Output: [n_sym, N/2+1]
 * complex128, bins 0..N/2 of the unnormalised forward DFT.
 */
int gf3_rfft_batch(gf3_ctx *ctx, const void *d_in, int64_t n_in,
                   const int64_t *d_offsets, int64_t n_sym,
                   void *d_out_c128, void *stream);

/*
 * get_symbols + remove_cp + fft + get_data + equalise + data-carrier select +
 * demap + PS (OFDM.py:391-505, 593, 603) fused, one packet ("frame") per
 * workgroup.  d_frame_offsets[f] = sample index of the first pilot symbol's
 * cyclic prefix (what get_symbols computes as peak+2, OFDM.py:393).
 *
 *   d_bits_packed  [F, gf3_bytes_per_frame] uint8, MSB-first (np.packbits order),
 *                  bit order packet -> symbol -> data carrier -> bit (OFDM.py:505)
 *   d_eq           optional [F*D, C] complex128: equalised data-carrier symbols
 *   d_Hs, d_He     optional [F, K]   complex128: Hest_start / Hest_end (OFDM.py:450-451)
 *   d_slope        optional [F]      float64   : polyfit slope p (OFDM.py:462)
 *   d_Hest         optional [F, D, K] complex128: channel model (OFDM.py:471-475)
 * A frame whose samples fall outside [0, n_in) decodes to zero bits and sets
 * bit 0 of *d_status (optional int32 on the device).
 */
int gf3_demod_frames(gf3_ctx *ctx, const void *d_in, int64_t n_in,
                     const int64_t *d_frame_offsets, int64_t F,
                     uint8_t *d_bits_packed, void *d_eq, void *d_Hs, void *d_He,
                     double *d_slope, void *d_Hest, int32_t *d_status, void *stream);

/*
 * The same call with a workspace, which opens the TWO-PHASE form for LONG packets, FEW at a time -- the reference's
 * own geometry (no_pilots = 20, packet_length = 180, OFDM.py:18; three packets in Final System Test.ipynb cell 7), where
 * one packet per workgroup would leave the chip idle.  The triple loop of equalise (OFDM.py:466-478) is independent over
 * (symbol, carrier) once Hest_start, Hest_end and the slope exist (OFDM.py:443-462), so: the pilot symbols of each side
 * are summed in the time domain over F x 2 x N/512 workgroups, the channel estimate runs once per packet, and the
 * data symbols of a packet are spread over ceil(D / Dc) workgroups that each pack their own word-aligned bit range.
 * Outputs are those of gf3_demod_frames: Hs / He / slope bit for bit, equalised symbols to ~1e-13 (the phasor of a
 * chunk's first symbol is computed directly instead of by recurrence), bits identical.
 *   d_work   gf3_demod_workspace_bytes(ctx, F) bytes of device memory, or NULL (then always the one-launch kernel)
 *   mode     0: the library chooses by F and D (two-phase when F <= 2 x CUs and a packet cuts into >= 2 chunks of the
 *               length that fills the chip once -- the measured crossover, tools/ab/time_split.py)
 *            1: always the one-launch kernel      2: two-phase whenever d_work is given
 */
int64_t gf3_demod_workspace_bytes(const gf3_ctx *ctx, int64_t F);
/* what gf3_demod_frames_ex(F, mode) does when it is given a workspace: returns 1 for the two-phase form (0: one launch)
 * and stores the chunk length Dc (data symbols per workgroup) and the chunks per packet it would use */
int gf3_demod_split_plan(const gf3_ctx *ctx, int64_t F, int32_t mode, int32_t *h_Dc, int32_t *h_nchunk);
int gf3_demod_frames_ex(gf3_ctx *ctx, const void *d_in, int64_t n_in,
                        const int64_t *d_frame_offsets, int64_t F,
                        uint8_t *d_bits_packed, void *d_eq, void *d_Hs, void *d_He,
                        double *d_slope, void *d_Hest, int32_t *d_status,
                        void *d_work, int32_t mode, void *stream);

/*
 * receiver.equalise as a stand-alone stage (OFDM.py:422-480): the same kernel
 * as gf3_demod_frames, fed with frequency-domain symbols instead of samples.
 *   d_data [F, D, K], d_start [F, P, K], d_end [F, P, K] complex128 (get_data's outputs)
 *   d_eq_all [F*D, K] complex128 (all carriers, as the reference returns)
 *   d_Hs, d_He, d_slope, d_Hest: as above, optional;  d_bits: packed decisions
 *   on the data carriers (required scratch/out, [F, gf3_bytes_per_frame]).
 */
int gf3_equalise(gf3_ctx *ctx, const void *d_data, const void *d_start, const void *d_end,
                 int64_t F, void *d_eq_all, void *d_Hs, void *d_He, double *d_slope,
                 void *d_Hest, uint8_t *d_bits, void *stream);

/*
 * chirp_method + the '+2' of get_symbols (OFDM.py:356-372, 393) for a batch of
 * independent frame buffers: frame f is searched for a chirp START in sample
 * range [f*stride + win_lo, f*stride + win_hi).  Peak rule as the reference:
 * correlation normalised by the window maximum, first local extremum above
 * `thresh`.  d_starts[f] = sample index of the first pilot symbol (chirp start +
 * Lc), ready to be passed to gf3_demod_frames; -1 where nothing qualifies.
 */
int gf3_sync_frames(gf3_ctx *ctx, const void *d_in, int64_t n_in,
                    int64_t F, int64_t stride, int32_t win_lo, int32_t win_hi,
                    int64_t *d_starts, double *d_peak_or_null, void *stream);

/*
 * The same search with an fp32 SCREEN in front (gf3rx_fscreen.h), opt-in: mode 1 evaluates every window in single
 * precision with a proven bound on |fp32 lag - exact lag| (2048-sample transforms held by one wave each, the chirp's
 * partition spectra multiplied in, one inverse transform) and takes the decision -- the index of the first extremum above
 * thresh x the window's maximum -- only where the bound decides it: every lag is certainly out, certainly in, or undecided,
 * and a window is resolved when no undecided lag precedes the first certain one.  Unresolved windows (noise at the threshold,
 * flat tops, non-finite samples, a maximum the bound cannot tell from zero) are listed on the device and the all-fp64 kernel
 * runs on exactly those.  No decision rests on an fp32 value the bound does not back: d_starts is what mode 0 writes.
 *   mode    0: all fp64 (== gf3_sync_frames)   1: screened
 *   d_work  gf3_sync_frames_workspace_bytes(ctx, F) bytes of device memory (mode 1; NULL selects mode 0); after the call its
 *           first int32 holds the number of windows that went to the fp64 kernel
 * Falls back to mode 0 when d_peak is asked for (an fp64 VALUE), when the window is wider than the context's max_window or
 * than 1024 lags, and when F exceeds 2^31.
 */
int64_t gf3_sync_frames_workspace_bytes(const gf3_ctx *ctx, int64_t F);
int gf3_sync_frames_ex(gf3_ctx *ctx, const void *d_in, int64_t n_in, int64_t F, int64_t stride,
                       int32_t win_lo, int32_t win_hi, int64_t *d_starts, double *d_peak_or_null,
                       int32_t mode, void *d_work, void *stream);
/* tests: the screening pass alone.  d_y32 [F][W] fp32 lags, d_err [F] the bound of each window, d_cls [F] 0 resolved with a
 * detection / 1 resolved without / 2 unresolved (d_starts[f] is then left alone); d_work as above */
int gf3_debug_frames_screen(gf3_ctx *ctx, const void *d_in, int64_t n_in, int64_t F, int64_t stride,
                            int32_t win_lo, int32_t win_hi, int64_t *d_starts, float *d_y32, float *d_err,
                            int32_t *d_cls, void *d_work, void *stream);

/*
 * chirp_method with full reference semantics on one contiguous stream
 * (OFDM.py:356-372): full-coverage matched filter, normalisation by the GLOBAL
 * maximum, extremum-and-threshold candidates, sequential non-max suppression
 * over Lc samples, and the except-branch that drops every detection when a
 * chirp ends within the last two samples.  Writes the indices i with
 * zeros[i]==True (ascending) to d_peaks (capacity cap) and their count to
 * *n_peaks (host).  d_work: gf3_sync_stream_workspace_bytes(ctx, n) bytes.
 * d_corr (optional, n+Lc-1 doubles) receives the raw correlation P (OFDM.py:358).
 */
int gf3_sync_stream(gf3_ctx *ctx, const void *d_r, int64_t n,
                    int64_t *d_peaks, int64_t cap, int64_t *n_peaks,
                    void *d_work, double *d_corr_or_null, void *stream);

/*
 * How the stream-mode sync evaluates the matched filter.  Screened: every lag is first evaluated in fp32 with a proven
 * error bound; only the lags that the bound cannot exclude (a few around every chirp) are re-evaluated as fp64 dot
 * products, and the reference's rule (global maximum, threshold, extremum test) is applied to those fp64 values -- no
 * decision rests on an fp32 number; streams on which the screen is not selective take the all-fp64 overlap-save.
 * mode 0 (default): screened for streams of 2^23 samples or more (below that the all-fp64 path is the faster one);
 * mode 1: always all-fp64; mode 2: screened at any length; mode 3: as 2, with the general screening kernel even
 * where the band-limited one applies (a chirp whose spectrum lives below 3/16 of the sample rate, as the reference's
 * 0-8 kHz sweep at 48 kHz does: the bins above are left out of the fp32 products and their norm joins the error
 * bound).  Calls that ask for d_corr are always all-fp64.
 *
 * gf3_sync_stream_ex is gf3_sync_stream with the mode as an argument and the diagnostics as an output; it reads the
 * context only, so any number of threads may call it on one context at once.
 *   h_info4 (host, optional): path taken (0 screened, 1 fp64 after a non-selective screen, 2 fp64), cells (of 14 lags)
 *   re-evaluated in fp64, cells among them that hold a candidate, candidates found.
 */
int gf3_sync_stream_ex(const gf3_ctx *ctx, const void *d_r, int64_t n,
                       int64_t *d_peaks, int64_t cap, int64_t *n_peaks,
                       void *d_work, double *d_corr_or_null,
                       int32_t mode, int64_t *h_info4_or_null, void *stream);
/* legacy conveniences: the default mode gf3_sync_stream uses (the one field of a context that a call writes; an atomic
 * int), and the h_info4 of the CALLING THREAD's last gf3_sync_stream / gf3_sync_stream_ex */
int gf3_sync_stream_mode(gf3_ctx *ctx, int32_t mode);
int gf3_sync_stream_info(const gf3_ctx *ctx, int64_t *h_out4);
/*
 * chirp_method (OFDM.py:356-372) on a stream that arrives piece by piece -- host ingest through pinned buffers, streams
 * longer than HBM -- with the reference's EXACT global rule.  The 0.4 threshold is relative to the maximum of the whole
 * stream (OFDM.py:359), known only at the end; so each piece keeps the running maximum and the few lags that could
 * still pass whatever the final maximum is, with the raw fp64 values the rule looks at, and the rule itself is applied
 * afterwards (or provisionally, with the maximum so far) by gf3_sync_decide.  Engine.receive_host (engine.py) drives
 * the two calls; INTEGRATION.md shows the loop.
 *
 * gf3_sync_chunk: all-fp64 matched filter of d_buf[0..n) -- a piece of the stream with at least Lc + 1 samples of the
 * previous piece in front of it (none at the stream's start) -- restricted to the lags [lag_lo, lag_hi) of the buffer's
 * own full convolution P_buf[m] = sum_k buf[m-Lc+1+k] chirp[k] (1 <= lag_lo <= lag_hi <= n+Lc-2; the caller chooses
 * them so that every lag of the stream is owned by exactly one piece and its taps and both neighbours are complete).
 *   d_run_max (device, TWO doubles): [0] in  = maximum of the earlier pieces (-inf before the first), NumPy's NaN rule;
 *                                        out = maximum including this piece's lags;   [1] out = this piece's own maximum
 *   *h_piece_max (host, optional): that own maximum -- a piece whose list overflowed needs no second look if its own
 *                        maximum stays below thresh * (final maximum) * (1 - 1e-6): none of its lags can pass
 *   listed: every owned lag g with P[g] >= thresh * run_max * (1 - 1e-6) (every owned lag while run_max is not a
 *           positive finite number): d_idx[k] = g - 1 + lag_offset (the zeros-index of OFDM.py:360 in the caller's
 *           global numbering), d_val3[3k..3k+2] = P[g-1], P[g], P[g+1]; ascending; *n_listed (host) = how many.
 *   GF3_ERANGE with *n_listed = the number wanted when cap is too small (nothing is written): look at the piece again
 *   with a larger list, or once the final maximum is known (preset *d_run_max).
 *   d_work: gf3_sync_chunk_workspace_bytes(ctx, n) bytes.  Synchronises the stream (it returns a count).
 */
int64_t gf3_sync_chunk_workspace_bytes(const gf3_ctx *ctx, int64_t n);
int gf3_sync_chunk(const gf3_ctx *ctx, const void *d_buf, int64_t n,
                   int64_t lag_lo, int64_t lag_hi, int64_t lag_offset,
                   double *d_run_max, int64_t *d_idx, double *d_val3, int64_t cap,
                   int64_t *n_listed, double *h_piece_max_or_null, void *d_work, void *stream);
/*
 * gf3_sync_decide: the reference's rule on the listed raw values -- p = P / *d_max first, candidate <=>
 * (p1-p0)(p2-p1) <= 0 and p1 > thresh (OFDM.py:359-361) -- then the suppression walk over Lc samples with the
 * except-branch (OFDM.py:364-370) for a stream whose zeros array has nz_total = n_total + Lc - 3 entries (pass
 * INT64_MAX / 2 for a provisional decision on a stream that has not ended).  Peaks as gf3_sync_stream returns them.
 *   d_work: gf3_sync_decide_workspace_bytes(ctx, n_listed) bytes.  Synchronises the stream.
 */
int64_t gf3_sync_decide_workspace_bytes(const gf3_ctx *ctx, int64_t n_listed);
int gf3_sync_decide(const gf3_ctx *ctx, const int64_t *d_idx, const double *d_val3, int64_t n_listed,
                    const double *d_max, int64_t nz_total,
                    int64_t *d_peaks, int64_t cap, int64_t *n_peaks, void *d_work, void *stream);

/* tests: the fp32 screening pass alone.  d_p32 [n+Lc-1] float; d_blk [2*nblk] float: per block of *h_hop lags its
 * maximum, then the bound on |P32 - P| of its lags (nblk = ceil((n+Lc-1) / hop)) */
int gf3_debug_stream_screen(gf3_ctx *ctx, const void *d_r, int64_t n, float *d_p32, float *d_blk,
                            int32_t *h_hop, void *stream);

/*
 * receiver.schmidlcox_method (OFDM.py:376-387; unused by receive()): running-sum
 * autocorrelation metric P[d+1] = P[d] + r[d+L] r[d+2L] - r[d] r[d+L] (L = K+1) over search_len lags;
 * *d_index (device int64) = first index of max |P| + N - 1.  Needs n >= search_len - 1 + 2L samples.
 */
int gf3_schmidl_cox(gf3_ctx *ctx, const void *d_r, int64_t n, int64_t search_len,
                    int64_t *d_index, void *stream);

/*
 * demap (OFDM.py:484-500) standalone: hard decisions for n symbols.
 * d_bits_u8: [n*mu] one byte per bit (0/1); d_idx_u8 (optional): [n] index of
 * the chosen constellation point (hardDecision = constellation[idx]).
 */
int gf3_demap_hard(gf3_ctx *ctx, const void *d_sym_c128, int64_t n,
                   uint8_t *d_bits_u8, uint8_t *d_idx_u8, void *stream);

/*
 * Transmit-side synthesiser: transmitter.map + build_OFDM_symbol + ifft + add_cp +
 * send_to_stream (OFDM.py:196-259) for F packets, one row of `stride` samples per packet:
 *   [gap_f zeros | chirp | P known symbols | D data symbols | P known symbols | zeros], symbols x2.
 *   d_bits_packed [F, gf3_bytes_per_frame]: payload in the format gf3_demod_frames writes
 *   d_filler_c128 [K]: value of every carrier that is not a data carrier (random_qpsk, OFDM.py:201-215),
 *                      indexed by carrier (entries of data carriers are ignored)
 *   d_gaps        [F] int64 leading zeros per row, or NULL
 *   out_dtype     GF3_F32 or GF3_F64
 */
int gf3_tx_frames(gf3_ctx *ctx, const uint8_t *d_bits_packed, const void *d_filler_c128,
                  const int64_t *d_gaps, int64_t F, void *d_out, int64_t stride,
                  int32_t out_dtype, void *stream);

/*
 * Known-channel zero forcing, the reference's older receive flow (`Weekend Challenge.ipynb` cells 9-19:
 * H = np.fft.fft(h, N); symbols = FFT(rx_no_cp) / H; demap of the data carriers): an optional extra mode with
 * sample-exact timing supplied by the caller, not part of receive() (OFDM.py:581-657 estimates the channel from
 * pilots instead, gf3_demod_frames).  The function it used (`equalise(OFDM_demod, H)`) no longer exists in OFDM.py, so
 * parity is pinned by the formula only.
 *   d_offsets [n_sym] int64: first of the N samples of each symbol (past its prefix);  d_h [n_taps] float64 channel taps
 *   d_eq [n_sym, C] complex128 equalised data-carrier symbols;  d_bits [n_sym*C*mu] one byte per bit;  d_idx optional
 *   d_work: gf3_known_h_workspace_bytes(ctx, n_sym) bytes
 */
int64_t gf3_known_h_workspace_bytes(const gf3_ctx *ctx, int64_t n_sym);
int gf3_equalise_known_h(gf3_ctx *ctx, const void *d_in, int64_t n_in, const int64_t *d_offsets, int64_t n_sym,
                         const double *d_h, int32_t n_taps, void *d_eq_c128, uint8_t *d_bits_u8,
                         uint8_t *d_idx_u8_or_null, void *d_work, void *stream);

/*
 * PS + decode (OFDM.py:504-505, 541-544) on the packed decisions of gf3_demod_frames: one int64 0/1 per bit, in the
 * reference's order (packet -> symbol -> data carrier -> bit), XORed with the whitening mask when one is given --
 * `bits ^ tile(known_sequence[:C*mu])[:len]` for encoding "XOR", d_mask_u8 = NULL for encoding "None".
 *   d_bits_packed [F, gf3_bytes_per_frame] uint8     d_mask_u8 [n_mask] one byte per bit (0/1), or NULL
 *   out_i64       [F * D*C*mu] int64: DEVICE memory, or PINNED host memory (hipHostMalloc / hipHostRegister -- e.g. a
 *                 torch tensor with pin_memory=True): the kernel then writes the array the reference returns straight
 *                 into host memory, 16 bytes per lane, and no separate copy is needed.  Pageable host memory is refused.
 */
int gf3_unpack_bits(gf3_ctx *ctx, const uint8_t *d_bits_packed, int64_t F, const uint8_t *d_mask_u8, int32_t n_mask,
                    void *out_i64, void *stream);

/* max-log soft demapping (not in the reference; LLR > 0 <=> bit 0). [n*mu] f32 */
int gf3_soft_demap(gf3_ctx *ctx, const void *d_sym_c128, int64_t n,
                   double noise_var, float *d_llr_f32, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GF3RX_H */
