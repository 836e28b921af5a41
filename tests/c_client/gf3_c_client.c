/*
 * A plain C99 client of libgf3rx: the receive path of OFDM.py (receiver.receive, OFDM.py:581-657) driven through
 * include/gf3rx.h alone -- no Python, no C++, no torch: device memory comes from the HIP runtime's C API.
 *
 *   gf3_c_client <case.bin> <bits_out.bin>
 *
 * case.bin (written by tests/test_c_client.py, little endian):
 *   int32  N, CP, P, D, Lc, mu, M, C, in_dtype, n_samples_lo, n_samples_hi(=0), encoding_xor
 *   double fs, f0, f1, thresh;  int32 fit_lo, fit_hi
 *   double const_re[M], const_im[M]; uint8 const_bits[M*mu]; double known_re[K], known_im[K]; int32 data_bins[C]
 *   uint8  mask[C*mu] (the whitening sequence, used when encoding_xor); samples (n x element size of in_dtype)
 * bits_out.bin: int64 n_bits, then n_bits int64 0/1 values -- what receive() returns as `bits`.
 * Exit code 0 on success; 2 + the gf3_status on a library error (message on stderr).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "gf3rx.h"

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define GF3OK(x) do { int rc_ = (x); if (rc_ != GF3_OK) { fprintf(stderr, "%s: %d %s\n", #x, rc_, gf3_last_error(NULL)); return 2 - rc_; } } while (0)

static int rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n ? 0 : 1; }

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s case.bin bits_out.bin\n", argv[0]); return 64; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 65; }
    int32_t h[12];
    double fl[4];
    int32_t fit[2];
    if (rd(f, h, sizeof h) || rd(f, fl, sizeof fl) || rd(f, fit, sizeof fit)) return 66;
    const int N = h[0], mu = h[5], M = h[6], C = h[7], K = N / 2 - 1;
    const int64_t n = (int64_t)(uint32_t)h[9] | ((int64_t)h[10] << 32);
    const int esz = h[8] == GF3_F64 ? 8 : (h[8] == GF3_F32 ? 4 : (h[8] == GF3_I16 ? 2 : 1));
    double* cre = malloc(sizeof(double) * M); double* cim = malloc(sizeof(double) * M);
    uint8_t* cbits = malloc((size_t)M * mu);
    double* kre = malloc(sizeof(double) * K); double* kim = malloc(sizeof(double) * K);
    int32_t* bins = malloc(sizeof(int32_t) * C);
    uint8_t* mask = malloc((size_t)C * mu);
    void* samples = malloc((size_t)n * esz);
    if (rd(f, cre, sizeof(double) * M) || rd(f, cim, sizeof(double) * M) || rd(f, cbits, (size_t)M * mu) || rd(f, kre, sizeof(double) * K) ||
        rd(f, kim, sizeof(double) * K) || rd(f, bins, sizeof(int32_t) * C) || rd(f, mask, (size_t)C * mu) || rd(f, samples, (size_t)n * esz)) return 67;
    fclose(f);

    gf3_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.N = N; cfg.CP = h[1]; cfg.P = h[2]; cfg.D = h[3]; cfg.Lc = h[4]; cfg.mu = mu; cfg.M = M; cfg.C = C; cfg.in_dtype = h[8];
    cfg.fs = fl[0]; cfg.f0 = fl[1]; cfg.f1 = fl[2]; cfg.thresh = fl[3]; cfg.fit_lo = fit[0]; cfg.fit_hi = fit[1];
    cfg.const_re = cre; cfg.const_im = cim; cfg.const_bits = cbits; cfg.known_re = kre; cfg.known_im = kim; cfg.data_bins = bins;
    cfg.max_window = 512;
    gf3_ctx* ctx = NULL;
    GF3OK(gf3_ctx_create(&cfg, &ctx));                                   /* CamG.__init__ + sync_chirp */

    void *d_r = NULL, *d_work = NULL, *d_dwork = NULL, *d_bits = NULL, *d_out = NULL, *d_mask = NULL;
    int64_t *d_peaks = NULL, *d_starts = NULL;
    const int64_t Lc = cfg.Lc > 0 ? cfg.Lc : 5 * (int64_t)(N + cfg.CP), cap = n / Lc + 8;
    HIPOK(hipMalloc(&d_r, (size_t)n * esz));
    HIPOK(hipMemcpy(d_r, samples, (size_t)n * esz, hipMemcpyHostToDevice));
    HIPOK(hipMalloc((void**)&d_peaks, sizeof(int64_t) * cap));
    HIPOK(hipMalloc(&d_work, (size_t)gf3_sync_stream_workspace_bytes(ctx, n)));
    int64_t n_peaks = 0;
    GF3OK(gf3_sync_stream(ctx, d_r, n, d_peaks, cap, &n_peaks, d_work, NULL, NULL));   /* chirp_method */
    if (n_peaks < 2) { fprintf(stderr, "fewer than two chirps\n"); return 2 - GF3_ENODETECT; }
    const int64_t F = n_peaks - 1;                                      /* get_symbols: peaks + 2, the last one dropped */
    int64_t* hp = malloc(sizeof(int64_t) * n_peaks);
    HIPOK(hipMemcpy(hp, d_peaks, sizeof(int64_t) * n_peaks, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < F; ++i) hp[i] += 2;
    HIPOK(hipMalloc((void**)&d_starts, sizeof(int64_t) * F));
    HIPOK(hipMemcpy(d_starts, hp, sizeof(int64_t) * F, hipMemcpyHostToDevice));
    const int32_t row = gf3_bytes_per_frame(ctx);
    HIPOK(hipMalloc(&d_bits, (size_t)F * row));
    HIPOK(hipMalloc(&d_dwork, (size_t)gf3_demod_workspace_bytes(ctx, F)));
    int32_t Dc = 0, nchunk = 0;
    const int two_phase = gf3_demod_split_plan(ctx, F, 0, &Dc, &nchunk);
    GF3OK(gf3_demod_frames_ex(ctx, d_r, n, d_starts, F, d_bits, NULL, NULL, NULL, NULL, NULL, NULL, d_dwork, 0, NULL));   /* remove_cp .. demap */
    const int64_t n_bits = F * (int64_t)cfg.D * C * mu;
    int64_t* bits = NULL;
    HIPOK(hipHostMalloc((void**)&bits, sizeof(int64_t) * (size_t)n_bits, 0));       /* pinned: the kernel writes it over PCIe */
    if (h[11]) { HIPOK(hipMalloc(&d_mask, (size_t)C * mu)); HIPOK(hipMemcpy(d_mask, mask, (size_t)C * mu, hipMemcpyHostToDevice)); }
    GF3OK(gf3_unpack_bits(ctx, d_bits, F, d_mask, C * mu, bits, NULL));  /* PS + decode */
    HIPOK(hipDeviceSynchronize());
    (void)d_out;

    FILE* g = fopen(argv[2], "wb");
    if (!g) { perror(argv[2]); return 68; }
    fwrite(&n_bits, sizeof n_bits, 1, g);
    fwrite(bits, sizeof(int64_t), (size_t)n_bits, g);
    fclose(g);
    printf("gf3_c_client: library %s, %lld samples, %lld chirps, %lld packets, demodulation %s (Dc %d, %d chunks), %lld bits\n", gf3_version(),
           (long long)n, (long long)n_peaks, (long long)F, two_phase ? "two-phase" : "one launch", (int)Dc, (int)nchunk, (long long)n_bits);
    gf3_ctx_destroy(ctx);
    return 0;
}
