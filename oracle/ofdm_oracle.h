/*
 * ofdm_oracle.h -- CPU restatement (f64) of the jkelleyrtp/ofdm TX/RX hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it, and
 * only as the checker.  The product path (ofdm_amd/, libofdm_hip.so) never links or imports it.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * checkout).  Rows marked EXT are extensions the north_star asks for that the reference does
 * not contain; for those this file IS the definition and
 * parity is "unpinned by the reference".
 *
 * Pinning status: the reference is nightly Rust with un-vendored dependencies and cannot be
 * built here, so the oracle is pinned by the reference's own known-answer tests
 * only: src/lib.rs:37-51, src/signals/mod.rs:385-394, :420-441, src/utils.rs:281-327,
 * src/receiver.rs:253-256, src/channel.rs:99-177, src/transmitter.rs:63-69 (tests/test_oracle_kat.py).
 * rustfft / rand::StdRng / bincode arithmetic is restated from their published contracts
 * (textbook DFT; fixint little-endian u128); StdRng pilot values are NOT reproduced -- pilot
 * tables are inputs, defaults come from the documented SplitMix64 generator below.
 */
#ifndef OFDM_ORACLE_H
#define OFDM_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double re, im; } oc64; /* num::Complex64, #[repr(C)] */

/* modulation = bits per constellation point */
enum { ORC_BPSK = 1, ORC_QPSK = 2, ORC_QAM16 = 4, ORC_QAM64 = 6, ORC_QAM256 = 8 };

/* ---- PRNG used for default pilot tables and the seeded channel (documented, not rand::StdRng) */
uint64_t orc_splitmix64(uint64_t *state);
double orc_uniform_pm1(uint64_t *state); /* U(-1,1): ((z>>11) * 2^-53) * 2 - 1 */
double orc_uniform_01(uint64_t *state);  /* U(0,1):  (z>>11) * 2^-53 */

/* ---- signals/mod.rs primitives */
/* 0 (default): twiddles from libm on every transform, like the reference's planner-per-call; 1: per-thread cached tables.
 * Bit-identical results; only the CPU-baseline timing differs. */
void orc_set_fft_cache(int on);
void orc_fft(oc64 *x, int n, int inverse);          /* mod.rs:27-58 (inverse scales by 1/n) */
void orc_fft_shift(oc64 *x, int n);                 /* mod.rs:61-77  */
void orc_ifft_shift(oc64 *x, int n);                /* mod.rs:80-95  */
int orc_xcorr_fft(const oc64 *a, int na, const oc64 *b, int nb, oc64 *out /* 2*na-1 */); /* mod.rs:186-217 */
void orc_convolve(const oc64 *a, int na, const oc64 *b, int nb, oc64 *out /* na+nb-1 */); /* mod.rs:219-237 */
oc64 orc_mean(const oc64 *x, int n);                /* mod.rs:251-259 */
oc64 orc_variance(const oc64 *x, int n);            /* mod.rs:239-249 (complex pseudo-variance) */
double orc_angle(oc64 z);                           /* receiver.rs:242-246 */

/* ---- utils.rs helpers */
void orc_to_bools(uint8_t b, uint8_t out[8]);       /* utils.rs:21-27 */
uint8_t orc_bools_to_u8(const uint8_t in[8]);       /* utils.rs:30-36 */
void orc_analysis(const uint8_t *l, const uint8_t *r, size_t n, uint32_t *num_errs,
                  uint32_t *num_block_errs, double *err_rate); /* utils.rs:45-68 */
void orc_sig_to_fc32(const oc64 *x, size_t n, float *out);     /* utils.rs:228-236 */
void orc_fc32_to_sig(const float *in, size_t n, oc64 *out);    /* utils.rs:238-254 */

/* ---- carrier map: reference classes for 64 bins (transmitter.rs:150-161), tiled k = n_fft/64 (EXT-4)
 * returns 0 = data, 1 = null, 2 = pilot */
int orc_carrier_class(int bin, int n_fft, int guard);
int orc_data_carriers(int n_fft, int guard);

/* ---- transmitter.rs */
void orc_locking_signal(int len, oc64 *out);                    /* transmitter.rs:60-72 */
void orc_default_preamble(int len, oc64 *out);                  /* transmitter.rs:75-84, SplitMix64 seed 100 */
void orc_default_training(int len, oc64 *out);                  /* transmitter.rs:88-96, SplitMix64 seed 50 */
void orc_stdrng_preamble(int len, oc64 *out);                   /* transmitter.rs:75-84 with rand 0.8 StdRng restated (unverified) */
void orc_stdrng_training(int len, oc64 *out);                   /* transmitter.rs:88-96 with rand 0.8 StdRng restated (unverified) */
int orc_rs_encode(const uint8_t *msg, int n, int nsym, uint8_t *out);   /* reed-solomon 0.2.1 construction (utils.rs:108) */
int orc_rs_correct(uint8_t *cw, int len, int nsym);
long orc_create_transmission_bytes(const uint8_t *data, long n, uint8_t *out);   /* utils.rs:97-136 */
long orc_decipher_transmission_bytes(const uint8_t *code, long n, uint8_t *out); /* utils.rs:150-180 */
void orc_chacha_keystream_block(const uint32_t in[16], int rounds, uint32_t out[16]); /* pinned by RFC 7539 2.3.2 */
size_t orc_modulate_count(size_t nbytes, int modulation);
size_t orc_modulate(const uint8_t *bytes, size_t nbytes, int modulation, oc64 *out); /* transmitter.rs:108-140 + EXT-1 */
void orc_encode_block(const oc64 *stream, size_t avail, size_t *consumed, int n_fft, int guard,
                      oc64 *out /* n_fft */);                   /* transmitter.rs:144-165 */
void orc_prefix_block(const oc64 *freq, int n_fft, int cp, oc64 *out /* n_fft+cp */); /* transmitter.rs:168-181 */
void orc_normalize(oc64 *x, size_t n);                          /* transmitter.rs:183-194 */
size_t orc_frame_len(size_t payload_bytes, int n_fft, int cp, int guard, int modulation);
size_t orc_encode(const uint8_t *data, size_t nbytes, int n_fft, int cp, int guard, int modulation,
                  const oc64 *preamble /* n_fft+cp */, const oc64 *training /* n_fft */,
                  oc64 *out);                                   /* transmitter.rs:11-58 */

/* ---- receiver.rs */
void orc_unprefix_block(const oc64 *in, int n_fft, int cp, oc64 *out);          /* receiver.rs:99-104 */
size_t orc_decode_block(const oc64 *in, int n_fft, int guard, oc64 *out);       /* receiver.rs:106-145 */
size_t orc_demodulate(const oc64 *sym, size_t nsym, int modulation, uint8_t *out); /* receiver.rs:147-190 + EXT-1 */
void orc_demap_indices(const oc64 *sym, size_t nsym, int modulation, uint8_t *idx);/* EXT-1 hard-decision index */
void orc_estimate_channel(const oc64 *blocks /* 5*(n_fft+cp) */, int n_fft, int cp,
                          const oc64 *training, oc64 *hk);                      /* receiver.rs:212-229 */
double orc_frequency_correction(const oc64 *left, const oc64 *right, int len);  /* receiver.rs:231-240 */
void orc_cfo_rotate(oc64 *x, size_t n, double f_delta, size_t first_index);     /* receiver.rs:44-50 */

/* EXT-3: sliding Schmidl-Cox metric over the repeated preamble.
 *   P(d) = sum_{m<W} conj(r[d+m]) r[d+m+L],  E(d) = sum_{m<W} |r[d+m]|^2,  R(d) = sum_{m<W} |r[d+m+L]|^2
 *   M(d) = |P|^2 / (E R)   (0 where E R == 0),   W = window_reps * L
 * over lags d in [0, n_lags) with d + W + L <= n.  Detection is "threshold, then peak":
 *   d1 = first d with M(d) >= threshold;  d_hat = first maximum of M over [d1, d1 + W].
 * (The frame holds two period-L regions -- 4 preamble repetitions and 5 identical training symbols,
 * transmitter.rs:27-34 -- so a global argmax is ambiguous; the first crossing selects the preamble.)
 * Returns d_hat (or -1 when no lag reaches the threshold), and fills P(d_hat), the peak metric and
 * f_delta = arg P(d_hat) / L (signed). */
int orc_sc_sync(const oc64 *r, size_t n, int L, int window_reps, long n_lags, double threshold, oc64 *p_hat,
                double *metric, double *f_delta);
/* all-lag metric dump for tests (metric[d] for d < n_lags, P[d]) */
void orc_sc_metric(const oc64 *r, size_t n, int L, int window_reps, long n_lags, double *metric, oc64 *p);

/* EXT-2: Hamming(7,4), 4 data bytes <-> 7 code bytes */
size_t orc_hamming74_encoded_len(size_t nbytes);
size_t orc_hamming74_encode(const uint8_t *data, size_t nbytes, uint8_t *out);
size_t orc_hamming74_decode(const uint8_t *code, size_t nbytes, uint8_t *out, uint32_t *corrected);

/* RX chain results */
typedef struct {
    int status;          /* 0 ok, -1 "Input not long enough, bailing early" (receiver.rs:27-29), -2 no sync */
    long offset;         /* samples trimmed from the front */
    double f_delta;      /* rad/sample */
    double metric;       /* sync peak (SC) or |xcorr| peak */
    size_t n_bytes;      /* decoded bytes after header truncate */
    size_t n_symbols;    /* data OFDM symbols demodulated */
} orc_rx_info;

/* Reference-faithful decode: xcorr_fft timing (offset = lag-1), mean-of-angles |CFO| (receiver.rs:9-96).
 * soft (optional) receives the equalised, phase-corrected data symbols. */
orc_rx_info orc_decode_ref(const oc64 *samples, size_t n, int n_fft, int cp, int guard, int modulation,
                           const oc64 *training, uint8_t *out, size_t out_cap, oc64 *soft, size_t soft_cap);

/* North-star decode: Schmidl-Cox timing + signed (cfo_abs=0) or |.| (cfo_abs=1) CFO, then the same chain.
 * sync_lags <= 0 searches every valid lag; offset = max(d_hat - L - backoff, 0).
 * max_symbols > 0 caps the number of data symbols (batch geometry); 0 = as many as the capture holds. */
orc_rx_info orc_decode_sc(const oc64 *samples, size_t n, int n_fft, int cp, int guard, int modulation,
                          const oc64 *training, int window_reps, long sync_lags, double threshold, int backoff,
                          int cfo_abs, int max_symbols, uint8_t *out, size_t out_cap, oc64 *soft, size_t soft_cap);

/* Chain with timing and CFO supplied (stage-level parity): */
orc_rx_info orc_decode_given(const oc64 *samples, size_t n, long offset, double f_delta, int n_fft, int cp,
                             int guard, int modulation, const oc64 *training, int max_symbols, uint8_t *out,
                             size_t out_cap, oc64 *soft, size_t soft_cap);

/* RX demod only (config 2): per symbol CP strip + FFT + optional /hk + decode_block + demodulate.
 * hk may be NULL (H == 1). Returns bytes written. */
size_t orc_rx_demod(const oc64 *samples, size_t n_symbols, int n_fft, int cp, int guard, int modulation,
                    const oc64 *hk, uint8_t *out, oc64 *soft);

/* ---- channel.rs:26-74 with a seeded PRNG (reference uses thread_rng) */
extern const double ORC_CHANNEL[64];
size_t orc_channel(const oc64 *tx, size_t n, double snr_db, int timing_error, uint64_t seed, oc64 *out /* n+63 */,
                   double *f_delta_used);

#ifdef __cplusplus
}
#endif
#endif
