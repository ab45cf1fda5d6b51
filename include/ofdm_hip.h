/*
 * ofdm_hip.h -- C ABI of libofdm_hip.so: the MI355X (gfx950) OFDM modulate / demodulate hot path.
 *
 * This is the drop-in boundary for the DSP bodies of jkelleyrtp/ofdm's src/transmitter.rs and
 * src/receiver.rs.  The reference has no FFI of its own (its hot path is plain `pub fn`s re-exported
 * from src/lib.rs:8-21); each entry point below names the reference function (file:line) whose body it
 * replaces.  The Rust host keeps its signatures and calls these through `extern "C"`
 * (bindings/ofdm_hip.rs).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types; every call returns an `int` status
 *     (0 = OFDM_OK, negative = error) and never throws or aborts across the boundary.
 *   - sample buffers are interleaved IQ `fc32` = {float re, float im} (8 B), the reference's wire format
 *     (src/utils.rs:228-254).  All `dev` pointers are DEVICE pointers on the context's GPU; buffers are
 *     caller-owned.  Kernels are enqueued on the context's HIP stream and NOT synchronised: call
 *     ofdm_synchronize() (or synchronise the stream you passed in) before reading results on the host.
 *   - batch first: the reference's one-frame functions are the n_frames = 1 case.
 *   - one context per (thread, GPU); a context is not thread-safe, different contexts are independent.
 *   - CFO values are f64 rad/sample (frequency_correction returns f64, src/receiver.rs:231): an f32 CFO would
 *     cost ~1e-5 rad of phase by the end of a 2000-sample frame.
 *   - pilot tables (preamble, training) are INPUTS: the reference draws them from rand 0.8 StdRng
 *     (src/transmitter.rs:75-96), which cannot be verified offline; ofdm_default_pilots() supplies the
 *     documented SplitMix64 defaults and ofdm_stdrng_pilots() a restatement of the StdRng tables.
 *
 * Extensions named by the north star that the reference lacks (64/256-QAM, Hamming(7,4), Schmidl-Cox,
 * N != 64) are defined in DESIGN.md section 3 and restated on the CPU in oracle/ofdm_oracle.c.
 */
#ifndef OFDM_HIP_H
#define OFDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFDM_HIP_ABI_VERSION 1

typedef struct ofdm_ctx ofdm_ctx;
typedef struct { float re, im; } ofdm_fc32;

/* call status */
enum {
    OFDM_OK = 0,
    OFDM_ERR_INVALID = -1,     /* bad argument / parameter combination */
    OFDM_ERR_UNSUPPORTED = -2, /* valid request this build does not implement */
    OFDM_ERR_NO_DEVICE = -3,   /* no HIP device / wrong architecture */
    OFDM_ERR_HIP = -4,         /* a HIP runtime call failed (ofdm_last_hip_error) */
    OFDM_ERR_NOMEM = -5,
    OFDM_ERR_UNCORRECTABLE = -6 /* outer RS block with more than 16 byte errors (the reference returns None) */
};

/* per-frame status written by ofdm_rx_decode_batch */
enum {
    OFDM_FRAME_OK = 0,
    OFDM_FRAME_SHORT = -1,  /* "Input not long enough, bailing early" (src/receiver.rs:27-29) */
    OFDM_FRAME_NOSYNC = -2, /* no lag reached the Schmidl-Cox threshold */
    OFDM_FRAME_BADTIMING = -3, /* OFDM_SYNC_REFERENCE: offset = lag - 1 outside the capture (the reference panics in split_off, receiver.rs:25) */
    OFDM_FRAME_HEADER = -4  /* fewer than 16 decoded bytes (reference panics in drain, receiver.rs:88) */
};

/* ModulationScheme (src/transmitter.rs:98-104) as bits per constellation point */
enum { OFDM_MOD_BPSK = 1, OFDM_MOD_QPSK = 2, OFDM_MOD_QAM16 = 4, OFDM_MOD_QAM64 = 6, OFDM_MOD_QAM256 = 8 };
enum { OFDM_ECC_NONE = 0, OFDM_ECC_HAMMING74 = 1 };
enum { OFDM_CFO_OFF = 0, OFDM_CFO_SIGNED = 1, OFDM_CFO_ABS = 2 }; /* ABS = reference's abs() (receiver.rs:239) */
/* timing / CFO detector of decode: the north star's Schmidl-Cox (default), or the reference's own pair -- cross-correlation
 * with the locking signal (xcorr_fft, src/signals/mod.rs:186-217; offset = idx_max - N = lag - 1, src/receiver.rs:20-25) and
 * frequency_correction on preamble repetitions 3 and 4 (src/receiver.rs:39, 231-240; always |.|) */
enum { OFDM_SYNC_SCHMIDL_COX = 0, OFDM_SYNC_REFERENCE = 1 };
/* ofdm_params.rx_path: rounds 2-4 offered a second N = 64 receive chain here (ONE_PASS = 2: timing and receive body in one kernel
 * from one LDS image).  It never beat the staged chain (7.4 against 4.5 ms per 1 M config-3 frames in round 5) and was removed;
 * the slot stays so that the struct layout does not change, and takes AUTO or STAGED (the same chain). */
enum { OFDM_RX_AUTO = 0, OFDM_RX_STAGED = 1 };

typedef struct {
    int32_t n_fft;            /* sub-carriers: 64 (reference) .. 4096, power of two */
    int32_t cp_len;           /* cyclic prefix, must be n_fft / 4 (reference: 16) */
    int32_t modulation;       /* OFDM_MOD_* (reference default Bpsk, transmitter.rs:17) */
    int32_t guard_bands;      /* 0 / 1 (reference default false, transmitter.rs:16) */
    int32_t ecc;              /* OFDM_ECC_* applied to the payload around encode / decode */
    int32_t sync_window_reps; /* Schmidl-Cox window W = reps * (n_fft + cp_len); 1..3, default 3 */
    int32_t sync_backoff;     /* frame start = d_hat - L - backoff, default 4 */
    int32_t cfo_mode;         /* OFDM_CFO_* , default SIGNED */
    float sync_threshold;     /* packet-detect threshold on M(d), default 0.5 */
    int32_t sync_mode;        /* OFDM_SYNC_*, default SCHMIDL_COX (this slot was reserved[0] == 0: same layout, same default) */
    int32_t rx_path;          /* OFDM_RX_AUTO or OFDM_RX_STAGED: one chain exists (see above) */
    int32_t reserved[5];      /* must be zero */
} ofdm_params;

/* ------------------------------------------------------------------ library / context */
int ofdm_abi_version(void);
const char *ofdm_strerror(int status);
int ofdm_device_count(int *count);
/* reference defaults: 64 carriers, CP 16, BPSK, no guard bands, no ECC */
int ofdm_default_params(ofdm_params *p);
/* SplitMix64 pilot tables (interleaved re,im doubles): preamble[2*(n_fft+cp)] = 0.25*U(-1,1) seed 100
 * (src/transmitter.rs:75-84), training[2*n_fft] = U(-1,1) seed 50 (src/transmitter.rs:88-96). Host call. */
int ofdm_default_pilots(int32_t n_fft, int32_t cp_len, double *preamble, double *training);
/* The reference's own tables: rand 0.8 StdRng (ChaCha12, seed_from_u64(100) / (50), gen_range(-1.0..1.0), re before im;
 * src/transmitter.rs:75-96) restated from the published algorithm.  Same layout as ofdm_default_pilots.  Needed only to
 * decode captures produced by the real Rust transmitter; UNVERIFIED against a running `rand` (none exists offline).
 * Host call. */
int ofdm_stdrng_pilots(int32_t n_fft, int32_t cp_len, double *preamble, double *training);
/* One ChaCha block (16 words out) from key[8] and state words 12..15, `rounds` = 8 / 12 / 20: exported so that the
 * published test vectors can pin the generator core. Host call. */
int ofdm_chacha_block(const uint32_t *key8, const uint32_t *words12_15, int32_t rounds, uint32_t *out16);
/* preamble / training: host pointers (interleaved doubles) or NULL for the defaults.
 * device: HIP device ordinal.  stream: a hipStream_t (as void*); NULL = the device's default (null) stream,
 * which is what torch uses as its current stream unless told otherwise. */
int ofdm_create(const ofdm_params *p, const double *preamble, const double *training, int device, void *stream,
                ofdm_ctx **out);
int ofdm_destroy(ofdm_ctx *ctx);
int ofdm_set_stream(ofdm_ctx *ctx, void *stream);
/* Give the context a non-blocking stream of its own (destroyed with it): what a host that runs several contexts side by side
 * -- one thread per context, on one device or on several -- wants instead of the shared default stream ("one host
 * thread + one stream per device"; include/ofdm_host.hpp ShardedContext). */
int ofdm_use_own_stream(ofdm_ctx *ctx);
int ofdm_synchronize(ofdm_ctx *ctx);
int ofdm_last_hip_error(const ofdm_ctx *ctx); /* raw hipError_t of the last failing HIP call */
/* Which kernels served the LAST stage-level / pipeline entry point on this context: the names of the kernels its launchers
 * really enqueued, in order, joined by '+' (e.g. "k_sc_stream<regs>+k_rx_prepare+k_rxframe1024<finish>").
 * Every shape-specialised launcher has a generic fallback (k_sym<...>) for requests outside its envelope -- output rows that are not
 * 4- / 16-byte aligned, soft outputs -- with the same results (captures need only their natural 8-byte alignment, any row stride); this call is how a caller (and the parity tests) tell which
 * one ran.  Returns the full length of the string (like snprintf), buf receives at most n - 1 characters. Host call. */
int ofdm_last_dispatch(const ofdm_ctx *ctx, char *buf, size_t n);
/* Per-context knobs and counters.  The library reads NO environment variable.  Keys a host legitimately needs:
 *   "grid_cap"             set/get, > 0 caps every persistent grid (0 = sized from the device); the parity tests use it to make small
 *                          batches walk many pipeline steps per workgroup
 *   "profile_build"        get: 1 in libofdm_hip_profile.so (-DOFDM_PROFILE_BUILD=1), 0 in the product build
 *   "stat_sc_slow_frames"  get: frames of the context's LAST N = 64 Schmidl-Cox search that were redone by the all-f64 kernel
 *   "stat_sc_redo_frames"  get: frames the first launch of the two-launch filter search left to the whole search
 *                          (both synchronise the stream; -1 = that search kept no such list)
 * Every other key is a laboratory switch between kernel variants (A/B measurements, profiling exits): listed and described in
 * ofdm_amd/csrc/ofdm_hip_tuning.h, NOT part of the drop-in contract, free to change between releases.  Unknown key: OFDM_ERR_INVALID;
 * a profiling key in the product build: OFDM_ERR_UNSUPPORTED.  Host calls. */
int ofdm_set_tuning(ofdm_ctx *ctx, const char *key, int64_t value);
int ofdm_get_tuning(const ofdm_ctx *ctx, const char *key, int64_t *value);

/* device-memory helpers so a host without its own HIP binding (the Rust crate) can stage buffers */
int ofdm_dev_alloc(ofdm_ctx *ctx, size_t bytes, void **dev);
int ofdm_dev_free(ofdm_ctx *ctx, void *dev);
int ofdm_memcpy_h2d(ofdm_ctx *ctx, void *dev, const void *host, size_t bytes); /* async on the ctx stream */
int ofdm_memcpy_d2h(ofdm_ctx *ctx, void *host, const void *dev, size_t bytes); /* synchronises the stream */
int ofdm_memset(ofdm_ctx *ctx, void *dev, int value, size_t bytes);

/* frame geometry for this context's parameters */
int ofdm_symbol_len(const ofdm_ctx *ctx);            /* S = n_fft + cp_len */
int ofdm_data_carriers(const ofdm_ctx *ctx);         /* 64k (no guard) or 48k */
int ofdm_bytes_per_symbol(const ofdm_ctx *ctx);      /* data_carriers * modulation / 8 */
int64_t ofdm_coded_len(const ofdm_ctx *ctx, int64_t payload_bytes);  /* after ECC (== payload when ECC off) */
int64_t ofdm_data_symbols(const ofdm_ctx *ctx, int64_t payload_bytes); /* D = ceil((16+coded)*8/bps / carriers) */
int64_t ofdm_frame_samples(const ofdm_ctx *ctx, int64_t payload_bytes); /* 10*S + D*S (transmitter.rs:22-54) */

/* ------------------------------------------------------------------ stage-level entry points
 * (mirror the reference's helper functions one to one; used by the parity tests and by hosts that
 * want to keep part of the chain on the CPU) */

/* SignalMut::fft / ifft (src/signals/mod.rs:27-58): n_vec transforms of length n_fft, unnormalised forward,
 * inverse scaled by 1/n_fft.  in == out allowed. */
int ofdm_fft_batch(ofdm_ctx *ctx, const ofdm_fc32 *in_dev, ofdm_fc32 *out_dev, int64_t n_vec, int inverse);
/* prefix_block (src/transmitter.rs:168-181): n_sym blocks of n_fft bins -> n_sym blocks of n_fft+cp samples */
int ofdm_ifft_cp_batch(ofdm_ctx *ctx, const ofdm_fc32 *freq_dev, ofdm_fc32 *out_dev, int64_t n_sym);
/* modulate + encode_block + prefix_block in one pass (src/transmitter.rs:40-53 without the frame header and without
 * normalize): a continuous byte stream -> n_sym symbols of n_fft+cp samples; n_sym * bytes_per_symbol >= n_bytes, bins
 * past the end of the stream carry 0 (transmitter.rs:158-160).  Same samples as the three staged calls. */
int ofdm_tx_symbols_batch(ofdm_ctx *ctx, const uint8_t *bytes_dev, int64_t n_bytes, ofdm_fc32 *out_dev, int64_t n_sym);
/* unprefix_block (src/receiver.rs:99-104): n_sym blocks of n_fft+cp samples -> n_sym blocks of n_fft bins */
int ofdm_unprefix_batch(ofdm_ctx *ctx, const ofdm_fc32 *in_dev, ofdm_fc32 *out_dev, int64_t n_sym);
/* modulate (src/transmitter.rs:108-140) + 16/64/256-QAM: n_bytes -> ceil(8 n_bytes / bps) points */
int ofdm_qam_map_batch(ofdm_ctx *ctx, const uint8_t *bytes_dev, int64_t n_bytes, ofdm_fc32 *out_dev);
/* demodulate (src/receiver.rs:147-190): n_sym points (multiple of 8) -> n_sym*bps/8 bytes;
 * idx_dev (optional) receives the per-point hard-decision index (bps-bit integer) */
int ofdm_qam_demap_batch(ofdm_ctx *ctx, const ofdm_fc32 *sym_dev, int64_t n_sym, uint8_t *bytes_dev,
                         uint8_t *idx_dev);
/* encode_block (src/transmitter.rs:144-165): per symbol, data_carriers points -> n_fft bins with nulls/pilots */
int ofdm_encode_block_batch(ofdm_ctx *ctx, const ofdm_fc32 *data_dev, ofdm_fc32 *bins_dev, int64_t n_sym);
/* normalize (src/transmitter.rs:183-194): per frame, divide by max(0, max re, max im) */
int ofdm_normalize_batch(ofdm_ctx *ctx, ofdm_fc32 *x_dev, int64_t n_frames, int64_t frame_stride, int64_t frame_len);
/* Hamming(7,4): 4 data bytes <-> 7 code bytes (north-star extension).
 * encode: n_bytes (zero-padded to a multiple of 4) -> ceil(n/4)*7; decode: floor(n/7)*4 bytes,
 * corrected_dev (optional uint32) accumulates the number of corrected codewords. */
int ofdm_hamming74_encode(ofdm_ctx *ctx, const uint8_t *in_dev, int64_t n_bytes, uint8_t *out_dev);
int ofdm_hamming74_decode(ofdm_ctx *ctx, const uint8_t *in_dev, int64_t n_bytes, uint8_t *out_dev,
                          uint32_t *corrected_dev);

/* Outer Reed-Solomon(255,223) framing of the demos (create_transmission_bytes / decipher_transmission_bytes,
 * src/utils.rs:97-180; reed-solomon 0.2.1: GF(2^8) 0x11d, generator 2, roots 2^0..2^31, parity after the data).
 * HOST calls on host buffers: byte-level work off the roofline, applied outside encode/decode as the reference does.
 * encode: n_bytes -> 255*(n/223 + 1) bytes (a final zero-padded block is always emitted, utils.rs:123-131);
 * decode: n_code -> 223*(n/255 + 1) bytes (the zero-padded remainder is decoded too, utils.rs:172-176), corrects up to
 * 16 bytes per block, OFDM_ERR_UNCORRECTABLE otherwise; *corrected (optional) = corrected bytes in total. */
int64_t ofdm_rs255_encoded_len(int64_t n_bytes);
int64_t ofdm_rs255_decoded_len(int64_t n_code);
int ofdm_rs255_encode(const uint8_t *data, int64_t n_bytes, uint8_t *out);
int ofdm_rs255_decode(const uint8_t *code, int64_t n_code, uint8_t *out, int32_t *corrected);

/* Schmidl-Cox sliding autocorrelation (north-star extension replacing xcorr_fft timing, receiver.rs:20-25).
 * Frame f occupies in_dev[f*frame_stride .. +frame_len).  Lags d in [0, n_lags) (n_lags <= 0: every lag with
 * d + W + L <= frame_len).  d_hat = first max of M over [d1, d1+W], d1 = first lag with M >= threshold;
 * d_hat[f] = -1 if none.  f_delta[f] = arg P(d_hat)/L (signed, rad/sample), metric[f] = M(d_hat).
 * n_lags bounds the peak window too: it is [d1, min(d1 + W, n_lags - 1)], so a bounded search returns an earlier lag than the
 * full one whenever the true peak lies beyond n_lags - 1 (same rule in the oracle; tests: test_bounded_search_clips_the_peak_window).
 * Any n_fft is served: L = 80 by the one-tile f32-filter / f64-decision kernel, L >= 160 by streaming chunk sums and a bounded
 * exact search whose LDS footprint does not depend on L (N = 4096 included). */
int ofdm_sc_correlate_batch(ofdm_ctx *ctx, const ofdm_fc32 *in_dev, int64_t n_frames, int64_t frame_stride,
                            int64_t frame_len, int64_t n_lags, int32_t *d_hat_dev, double *f_delta_dev,
                            float *metric_dev);
/* xcorr_fft (src/signals/mod.rs:186-217) for every capture a[f] (a_len samples) against b (nb samples, device): the full
 * cross-correlation out[i] = sum_n a[n + i - (a_len - 1)] conj(b[n]) over the 2 a_len - 1 fft_shifted indices (zero lag at
 * a_len - 1), idx_max[f] = the FIRST index of the largest |out|^2 (0 when everything is zero, as the reference's loop), peak[f]
 * (optional) = |out[idx_max]|, out_dev (optional, out_stride >= 2 a_len - 1) = the whole output.  Evaluated lag by lag in f64
 * instead of through three odd-length FFTs: the same numbers to ~1e-15.  decode's offset is idx_max - a_len (receiver.rs:21). */
int ofdm_xcorr_batch(ofdm_ctx *ctx, const ofdm_fc32 *a_dev, int64_t n_frames, int64_t a_stride, int64_t a_len,
                     const ofdm_fc32 *b_dev, int32_t nb, int32_t *idx_max_dev, float *peak_dev, ofdm_fc32 *out_dev,
                     int64_t out_stride);
/* frequency_correction (src/receiver.rs:231-240): |mean_m angle(right[m]/left[m])| / L over n_pairs blocks of
 * L = n_fft+cp samples; pair p reads left = in[p*stride ..], right = in[p*stride + right_offset ..] */
int ofdm_frequency_correction_batch(ofdm_ctx *ctx, const ofdm_fc32 *in_dev, int64_t n_pairs, int64_t stride,
                                    int64_t right_offset, double *f_delta_dev);
/* CFO derotation (src/receiver.rs:44-50): x[f][n] *= exp(-j f_delta[f] (first_index[f] + n)), in place.
 * first_index_dev may be NULL (0). */
int ofdm_cfo_rotate_batch(ofdm_ctx *ctx, ofdm_fc32 *x_dev, int64_t n_frames, int64_t frame_stride,
                          int64_t frame_len, const double *f_delta_dev, const int32_t *first_index_dev);
/* estimate_channel (src/receiver.rs:212-229): H[k] = mean_b FFT(block_b minus CP)[k] / training[k] over the five
 * training blocks starting at sample offset_dev[f] + 5*L of frame f (offset_dev NULL: 0).
 * f_delta_dev (optional): blocks are derotated with sample_id counted from offset_dev[f]. hk_dev: n_frames*n_fft */
int ofdm_estimate_channel_batch(ofdm_ctx *ctx, const ofdm_fc32 *in_dev, int64_t n_frames, int64_t frame_stride,
                                int64_t frame_len, const int32_t *offset_dev, const double *f_delta_dev,
                                ofdm_fc32 *hk_dev);
/* RX demod = unprefix_block + equalise + decode_block + demodulate (src/receiver.rs:64-83) over
 * syms_per_frame OFDM symbols per frame.  Symbol k of frame f starts (at its cyclic prefix) at sample
 * offset_dev[f] + first_symbol*L + k*L of the frame (offset_dev NULL: 0); samples at or beyond frame_len read
 * as zero (pad_chunk, receiver.rs:203-210).  f_delta_dev (optional) derotates with sample_id counted from
 * offset_dev[f].  hk_dev: per-frame channel (hk_stride = n_fft), shared (hk_stride = 0) or NULL (H == 1).
 * out_dev[f*out_stride ..]: syms_per_frame * bytes_per_symbol bytes.  soft_dev (optional): equalised,
 * phase-corrected data points, syms_per_frame*data_carriers per frame. */
int ofdm_rx_demod_batch(ofdm_ctx *ctx, const ofdm_fc32 *in_dev, int64_t n_frames, int64_t frame_stride,
                        int64_t frame_len, int32_t first_symbol, int32_t syms_per_frame,
                        const int32_t *offset_dev, const double *f_delta_dev, const ofdm_fc32 *hk_dev,
                        int64_t hk_stride, uint8_t *out_dev, int64_t out_stride, ofdm_fc32 *soft_dev);

/* ------------------------------------------------------------------ pipelines */

/* encode (src/transmitter.rs:11-58) for a batch: frame f = [lock][preamble x4][training x5][data symbols],
 * normalised per frame.  payload f = payload_dev[f*payload_stride .. + len_f), len_f = payload_len_dev[f] or
 * payload_bytes when NULL; a len_f outside [0, payload_bytes] is clamped to that range.  EVERY row, the last one included, must be READABLE for payload_bytes bytes whatever its len_f (the
 * kernels prefetch whole rows and mask afterwards): payload_stride >= payload_bytes (OFDM_ERR_INVALID otherwise when n_frames > 1),
 * and the buffer ends no earlier than the last row's payload_bytes.  Every frame is laid out for payload_bytes (D = ofdm_data_symbols(payload_bytes))
 * and written to out_dev[f*out_stride ..] (out_stride >= ofdm_frame_samples(payload_bytes)).
 * With ECC the payload is Hamming(7,4)-encoded first and the header carries the coded length. */
int ofdm_tx_encode_batch(ofdm_ctx *ctx, const uint8_t *payload_dev, int64_t n_frames, int64_t payload_stride,
                         const int32_t *payload_len_dev, int32_t payload_bytes, ofdm_fc32 *out_dev,
                         int64_t out_stride);

/* decode (src/receiver.rs:9-96) for a batch, with Schmidl-Cox timing/CFO in place of xcorr_fft:
 * sync over n_lags lags -> offset = max(d_hat - L - backoff, 0) -> CFO derotation -> channel estimate ->
 * per symbol FFT / equalise / pilot phase / demap -> 16-byte length header -> truncate [-> Hamming decode].
 * At most max_symbols data symbols per frame are demodulated (fewer if the frame is shorter).
 * out_dev[f*out_stride ..] receives out_len_dev[f] bytes (out_stride >= max_symbols*bytes_per_symbol).
 * status/offset/f_delta/metric are per-frame outputs; any of offset/f_delta/metric may be NULL. */
int ofdm_rx_decode_batch(ofdm_ctx *ctx, const ofdm_fc32 *in_dev, int64_t n_frames, int64_t frame_stride,
                         int64_t frame_len, int64_t n_lags, int32_t max_symbols, uint8_t *out_dev,
                         int64_t out_stride, int32_t *out_len_dev, int32_t *status_dev, int32_t *offset_dev,
                         double *f_delta_dev, float *metric_dev);

/* ------------------------------------------------------------------ host buffers
 * The reference's encode returns a host Vec<Complex64> and decode consumes one (src/transmitter.rs:11-15, src/receiver.rs:9-13).
 * These entry points take HOST pointers and do the staging inside the library: the batch is cut into chunks of chunk_frames
 * frames (<= 0: about 48 MB of samples) that move through three slots -- H2D of chunk k + 1 on a copy stream, the kernels of chunk k
 * on the context's stream, D2H of chunk k - 1 on a second copy stream.  Page-locked caller memory (ofdm_host_alloc,
 * ofdm_host_register) is DMA-ed in place; pageable memory is staged through pinned bounce buffers by the calling thread (slower:
 * one extra host copy).  The calls are synchronous: results are in the caller's arrays when they return.  Same results, byte for
 * byte, as the device-buffer entry points they wrap. */
int ofdm_host_alloc(size_t bytes, void **host);   /* page-locked host memory (hipHostMalloc) */
int ofdm_host_free(void *host);
int ofdm_host_register(void *host, size_t bytes); /* page-lock memory the caller owns (a Rust Vec's buffer) for the time being */
int ofdm_host_unregister(void *host);
int ofdm_host_is_pinned(const void *host, size_t bytes); /* 1 when [host, host + bytes) can be DMA-ed in place */
/* ofdm_rx_decode_batch on host buffers: frame f = in_host[f*frame_stride .. +frame_len) (frame_stride >= frame_len), outputs as there
 * but in host arrays; offset / f_delta / metric may be NULL. */
int ofdm_rx_decode_host(ofdm_ctx *ctx, const ofdm_fc32 *in_host, int64_t n_frames, int64_t frame_stride, int64_t frame_len,
                        int64_t n_lags, int32_t max_symbols, uint8_t *out_host, int64_t out_stride, int32_t *out_len_host,
                        int32_t *status_host, int32_t *offset_host, double *f_delta_host, float *metric_host, int64_t chunk_frames);
/* ofdm_rx_demod_batch on host buffers for regular streams (no offsets, no CFO, H == 1): the BASELINE metric's path. */
int ofdm_rx_demod_host(ofdm_ctx *ctx, const ofdm_fc32 *in_host, int64_t n_frames, int64_t frame_stride, int64_t frame_len,
                       int32_t first_symbol, int32_t syms_per_frame, uint8_t *out_host, int64_t out_stride, int64_t chunk_frames);
/* ofdm_tx_encode_batch on host buffers (payload_len_host may be NULL; row f is read for its own length only).  A length above
 * payload_bytes is clamped to payload_bytes, as the device entry point clamps it; the samples are those of ofdm_tx_encode_batch. */
int ofdm_tx_encode_host(ofdm_ctx *ctx, const uint8_t *payload_host, int64_t n_frames, int64_t payload_stride,
                        const int32_t *payload_len_host, int32_t payload_bytes, ofdm_fc32 *out_host, int64_t out_stride,
                        int64_t chunk_frames);

/* ------------------------------------------------------------------ one long capture
 * The reference's receiver hands ONE long buffer to each decode! (examples/jetson_rx.rs:15-17,48-49,84-86: 2 000 000 samples) and
 * looks for the packet anywhere in it (src/receiver.rs:20-25).  A batch of one such frame would occupy one workgroup; here the
 * search runs as a batch of overlapping SLICES of the capture (slice_lags own lags each, <= 0: chosen by the library, plus a
 * read-only halo of 2W + L samples).
 *
 * ofdm_sc_correlate_long: the Schmidl-Cox detection whose first threshold crossing d1 lies in [lag_lo, lag_hi) (lag_hi <= 0: the
 * capture's last lag), with its whole peak window [d1, d1 + W] -- which may reach past lag_hi -- exactly as ONE search over the
 * whole capture evaluates it.  *d_hat = the lag in the capture, -1 if none; f_delta / metric optional.  A host that splits the
 * lags of one capture over several contexts / GPUs (ofdm_amd.dist.halo_ranges) takes the answer of the lowest range that has one.
 * Synchronous; d_hat, f_delta, metric are HOST pointers. */
int ofdm_sc_correlate_long(ofdm_ctx *ctx, const ofdm_fc32 *in_dev, int64_t n_samples, int64_t lag_lo, int64_t lag_hi,
                           int64_t slice_lags, int64_t *d_hat, double *f_delta, float *metric);
/* decode (src/receiver.rs:9-96) of ONE long capture: the search above over [lag_lo, lag_hi) (skipped when d_hat_known >= 0: a
 * detection merged from several contexts), then the receive chain from that frame on.  With the whole lag range it is identical to
 * ofdm_rx_decode_batch with n_frames = 1 on the whole capture: status, offset (into the capture), CFO, metric, bytes -- except that
 * a capture without a detection (OFDM_FRAME_NOSYNC) reports f_delta = metric = 0 without running the chain.  With lag_lo > 0 the
 * detection is the first crossing at or after lag_lo (lags in front of it are never looked at, although the frame's own first
 * samples may lie there), i.e. the result of ofdm_rx_decode_batch on the capture from lag_lo on, re-based to the whole capture.  out_dev[0 .. out_cap) is a DEVICE
 * buffer (out_cap as out_stride there); out_len, status, offset, f_delta, metric are HOST pointers (the last three optional).
 * With sync_mode = OFDM_SYNC_REFERENCE the lag range must be the whole capture. */
int ofdm_rx_decode_long(ofdm_ctx *ctx, const ofdm_fc32 *in_dev, int64_t n_samples, int64_t lag_lo, int64_t lag_hi,
                        int64_t d_hat_known, int32_t max_symbols, uint8_t *out_dev, int64_t out_cap, int32_t *out_len,
                        int32_t *status, int64_t *offset, double *f_delta, float *metric);
/* the same from and to host memory: what `decode!(samples)` of a 2 M-sample Vec is */
int ofdm_rx_decode_long_host(ofdm_ctx *ctx, const ofdm_fc32 *in_host, int64_t n_samples, int32_t max_symbols, uint8_t *out_host,
                             int64_t out_cap, int32_t *out_len, int32_t *status, int64_t *offset, double *f_delta, float *metric);

/* ------------------------------------------------------------------ loop-back test bench
 * channel (src/channel.rs:33-74), per frame:
 *     y = convolve(tx, CHANNEL)                          tx_len + 63 samples (CHANNEL: the 64 taps of channel.rs:26-31)
 *     timing_error:  y[i] *= exp(+j f (i + 1)),  f = pi U(0,1) / 80                                   (channel.rs:48-62)
 *     y[i] += sqrt(0.5 var / snr) (U(-1,1) + j U(-1,1)),  var = the complex pseudo-variance of y, snr = 10^(snr_db/10)
 * The reference draws from an unseeded thread_rng; here frame f draws from SplitMix64(seed + f) in the reference's order
 * (f first when timing_error, then re, im per sample) -- the stream oracle/ofdm_oracle.c:orc_channel uses, so the two can be
 * compared sample by sample.  Test-bench extensions, NULL = the reference's behaviour: delay_dev[f] >= 0 places the channel
 * output at out + delay[f] (what does not fit out_len is dropped; every one of the out_len slot samples receives noise, the
 * samples outside the channel output nothing else), f_delta_in_dev[f] replaces the drawn CFO (any sign).
 * out_len >= tx_len + 63, out_stride >= out_len.  f_delta_out_dev (optional): the CFO applied to each frame. */
int ofdm_channel_batch(ofdm_ctx *ctx, const ofdm_fc32 *tx_dev, int64_t n_frames, int64_t tx_stride, int64_t tx_len,
                       double snr_db, int32_t timing_error, uint64_t seed, const int32_t *delay_dev,
                       const double *f_delta_in_dev, ofdm_fc32 *out_dev, int64_t out_stride, int64_t out_len,
                       double *f_delta_out_dev);
int ofdm_channel_taps(double *taps64); /* CHANNEL (src/channel.rs:26-31). Host call. */

/* ------------------------------------------------------------------ measurement helpers
 * HIP events on the context's stream, so a host without HIP bindings can time kernels (bench.py). */
/* Read-only stream over n_symbols 80-sample (640-byte) symbols at in_dev, nothing but the loads: the practical HBM read
 * ceiling of an access pattern, to read the demod kernel's rate against.  pattern 0 = the N = 64 demod kernel's access
 * (8-byte loads, the 128-byte cyclic prefix of every symbol never touched), 1 = the same loads over whole symbols,
 * 2 = unit-stride 16-byte loads.  Enqueued on the context's stream; time it with ofdm_timer_start / _stop_ms. */
int ofdm_hbm_read_probe(ofdm_ctx *ctx, const ofdm_fc32 *in_dev, int64_t n_symbols, int32_t pattern);
int ofdm_timer_start(ofdm_ctx *ctx);
int ofdm_timer_stop_ms(ofdm_ctx *ctx, float *elapsed_ms); /* records, synchronises, returns elapsed */

#ifdef __cplusplus
}
#endif
#endif /* OFDM_HIP_H */
