// kernels.hpp -- host-visible launch interface of the HIP kernels (internal to libofdm_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef OFDM_PROFILE_BUILD
#define OFDM_PROFILE_BUILD 0 // 1 (ofdm_amd/build.py, profile=True): libofdm_hip_profile.so with the ablation / section-timing branches
#endif

namespace ofdm {

// The `debug` branches of the kernels (early exits, s_memtime section ticks) are compiled only into the profile build.
constexpr bool kProfile = OFDM_PROFILE_BUILD != 0;

// Per-context tuning (ofdm_set_tuning): A/B switches between kernel families, grid shapes and the tests' grid cap.  The
// library reads no environment variable; a host that wants one maps it onto these keys itself (tools/tune_env.py).
struct Tuning {
    int no_sc80 = 0;               // L = 80, W = 240: the round-4 f32 filter pair (k_sc_cf<128,first> + k_sc_cf<256,list>) instead of the exact streaming detector k_sc80 (A/B)
    int sc80_depth = 2;            // k_sc80: steps between the last read of a ring piece and its refill (1: 9 - 10 KiB in flight per wavefront, 2: 7)
    int no_sc_stream = 0;          // L = 160 .. 1280: k_scb_chunks + k_scb_fine / k_sc_tile instead of the streaming detector
    int no_sc_big = 0;             // long-period Schmidl-Cox through k_sc_tile instead of k_scb_chunks + k_scb_fine
    int no_fast64 = 0, no_demod4096 = 0, no_mid_kernels = 0, no_rxframe1024 = 0, no_txframe64 = 0; // take the generic k_sym
    int no_rx1024_finish = 0;      // k_rxframe1024 writes raw bytes and k_rx_finish runs as its own launch
    int no_rxframe64_split = 0;    // k_rxframe64 as ONE kernel with both frame bodies (three waves per SIMD) instead of the common-body / cut-body pair (A/B)
    long long grid_cap = 0;        // > 0: caps every persistent grid (test hook: many steps per workgroup on a small batch)
    int tx_waves = 16;             // k_txframe64: wavefronts per CU
    int txframe_keep_steps = 1;    // k_txframe_mid: 1 = frames whose data symbols fit ONE workgroup step (<= 32 / R symbols) are built once, their points kept in registers until the maximum is known; 0 = always twice
    int no_txframe_optimistic = 0; // k_txframe_mid / k_txframe4096, frames of more than one step: 1 = the round-4 scheme (every symbol built twice: once for the frame maximum, once to be stored); 0 = symbols leave divided by the header maximum while the frame maximum forms, a round is rebuilt only if a frame exceeds it
    int txframe_rewrite = 0;       // k_txframe_mid, frames of more than one step: 1 = build every symbol ONCE -- unnormalised samples out, then a rescale sweep over what
                                   // was just written (L2 / memory-side cache) -- instead of building every symbol twice
    int sc_wg_per_cu = 7;          // k_sc_cf: persistent workgroups per CU
    int sc128_one_wave = 1;        // k_sc_cf over <= 960 lags: one wavefront per frame (two chunks per lane, 15 frames per CU, no wavefront idling through the fine pass) instead of two (0: A/B); measured 1.46 -> 1.31 ms per 262 144 config-3 frames over all lags, 1.43 -> 1.25 bounded
    int sc_first_lags = 576;       // k_sc_cf, searches of >= twice as many lags: the lags the first launch looks at (0 = one launch over every lag).
                                   // 576 = a crossing up to lag 335 with its whole window of W = 240 lags: a packet that starts within ~250 samples of its slot.
                                   // Round 3 used 384 (config 3's delay <= 64); round 4's sweep (round-4 script tools/first_lags_sweep.py, since removed, 262 144 frames, one box): the first
                                   // launch costs the same up to 576 lags (early packets: 0.48 ms at 384, 0.49 at 576, 0.71 from 592 on), while on late
                                   // packets (delay 1..401, 10 % empty slots) every lag more decides more frames: search 1.42 -> 0.93 ms, chain 2.05 -> 1.59 ms
    int demod64_wg_per_cu = 0;     // k_demod64: persistent workgroups per CU (0 = from the occupancy API)
    int demod64_burst = 16;        // k_demod64: groups per store burst (16 / 8 / 4; 1 = no bursts)
    int demod64_narrow_stores = 0; // k_demod64: 4-byte instead of 16-byte image stores
    int demod64_store_policy = 0;  // k_demod64: cache-policy bits on the 16-byte image stores (0 default, 1 nt, 2 sc1, 3 sc0 sc1): the output-buffer populations probe
    int scb_two_segments = 0;      // k_scb_chunks: two-segment staging also for L <= 1280
    int scb_big_tiles = 0;         // k_scb_fine: 1280-lag tiles / 128 threads
    int debug_demod64 = 0, debug_sc = 0, debug_tx = 0; // profile build only (kProfile)
};
inline const Tuning &tuning_or_default(const Tuning *t) { static const Tuning d; return t ? *t : d; }

// Which kernels served the last entry point (ofdm_last_dispatch): every launcher appends the kernel it really launched.
struct Trace {
    char buf[256];
    int len = 0;
    void reset() { len = 0; buf[0] = 0; }
    void add(const char *name) {
        if (len && len < (int)sizeof(buf) - 1) buf[len++] = '+';
        while (*name && len < (int)sizeof(buf) - 1) buf[len++] = *name++;
        buf[len] = 0;
    }
};
inline void trace_add(Trace *t, const char *name) { if (t) t->add(name); }

// Counters of the last Schmidl-Cox search on a context (ofdm_get_tuning "stat_sc_slow_frames" / "stat_sc_redo_frames"): how many
// frames went to the all-f64 kernel, and how many the first launch of the two-launch search left to the whole search.  `dev` is a
// small device buffer OWNED BY THE CONTEXT ([0] slow, [1] redo): the search copies its counters there on its stream, so that the
// figures survive the search's workspace being regrown or reused by another entry point.
struct ScStats {
    int32_t *dev = nullptr;
    bool has_slow = false, has_redo = false;
};

struct SymParams {
    const Tuning *tune = nullptr; // nullptr = defaults
    Trace *trace = nullptr;
    // input samples / bins
    const float2 *in = nullptr;
    long long n_frames = 0;
    long long frame_stride = 0; // samples between frame bases
    long long frame_len = 0;    // valid samples per frame; reads at or beyond it return 0 (pad_chunk)
    int syms_per_frame = 1;
    int first_symbol = 0;       // symbol index (in units of in_sym_stride) of symbol 0
    int in_sym_stride = 0;      // samples between consecutive symbols in the input
    int in_skip = 0;            // samples skipped at the start of each symbol (cyclic prefix) on input
    int sym_len = 0;            // S = N + CP
    const int32_t *offset = nullptr;     // per-frame sample offset of the trimmed frame start
    const double *f_delta = nullptr;     // per-frame CFO (rad/sample); sample ids count from offset[f]
    const int32_t *nsym_frame = nullptr; // per-frame number of live symbols (demod)
    // channel / tables
    const float2 *hk = nullptr;
    long long hk_stride = 0;
    const float2 *tw = nullptr;           // exp(-2 pi i m / N)
    const float2 *inv_training = nullptr; // 1 / training[k]
    // outputs
    float2 *out = nullptr;       // FFT / IFFT(+CP) / channel estimate / TX frames
    long long out_stride_s = 0;  // TX: samples between output frames
    unsigned char *out_bytes = nullptr;
    long long out_stride = 0;    // bytes between per-frame outputs
    float2 *soft = nullptr;
    // modulation / TX
    int bps = 1;
    int guard = 0;
    const uint8_t *payload = nullptr;
    long long payload_stride = 0;
    const int32_t *payload_len = nullptr;
    int payload_bytes = 0;
    unsigned *frame_max = nullptr; // per-frame max(0, re, im) as float bits
    long long tx_raw_total = -1;   // >= 0: TX of a continuous symbol stream (modulate + encode_block + prefix_block only):
                                   // payload is tx_raw_total plain bytes, one symbol per "frame", no length header, no frame
                                   // header blocks, no normalise
};

// N = 64 RX-demod fast path (kernels_fast.hip)
struct Fast64Params {
    const float2 *in = nullptr;
    long long frame_stride = 0;
    int first_symbol = 0;
    const float2 *hk = nullptr; // shared channel or nullptr
    const float2 *tw = nullptr;
    unsigned char *out = nullptr;
    long long out_stride = 0;
    int groups_per_frame = 1;   // 8-symbol groups per frame
    long long n_groups = 0, stride_groups = 0;
    long long f0 = 0, blk_df = 0, wave_df = 0, step_df = 0; // frame index bookkeeping without division
    int k0 = 0, blk_dk = 0, wave_dk = 0, step_dk = 0;
    int wide_stores = 0;        // the packed image leaves with 16-byte stores (aligned output)
    int store_policy = 0;       // lab key demod64_store_policy: cache-policy bits on the image stores (0 default, 1 nt, 2 sc1, 3 sc0 sc1)
    int debug = 0;              // profiling aid (OFDM_DEMOD64_DEBUG): 1 no stores, 2 no packing either, 3 loads + first butterfly only
};
hipError_t run_demod64_fast(const SymParams &p, hipStream_t st, int num_cu);
// N = 4096 RX demod as 64 x 64 (regular streams: no offset / CFO / per-frame symbol counts / soft output)
hipError_t run_demod4096(const SymParams &p, hipStream_t st, int num_cu);
// fused estimate_channel + per-symbol demod [+ finish] for N = 1024 frames (16 x 64 FFT, kernels_rx1024.hip), the N = 1024
// analogue of run_rxframe64.  final_* (optional): length header, truncate and Hamming decode done by the same kernel when the
// frame's packed bytes fit its LDS image; *fused_out reports whether they were (else the caller runs k_rx_finish).  A frame that
// must not be decoded is one with nsym_frame[f] == 0 (the prepare kernels guarantee it for every status != 0).
hipError_t run_rxframe1024(const SymParams &p, float2 *hk_out, hipStream_t st, int num_cu, unsigned char *final_out = nullptr,
                           long long final_stride = 0, int32_t *final_len = nullptr, int ecc = 0, bool *fused_out = nullptr);
// N = 4096 continuous-stream TX (map + IFFT + CP) as 64 x 64; needs tx_raw_total >= 0
hipError_t run_tx4096(const SymParams &p, hipStream_t st, int num_cu);
hipError_t run_txframe4096(const SymParams &p, const float2 *header, float header_max, hipStream_t st, int num_cu);
// N = 128 .. 2048 symbol-stream fast paths (kernels_mid.hip); hipErrorNotSupported => generic k_sym
hipError_t run_demod_mid(int n_fft, const SymParams &p, hipStream_t st, int num_cu);
hipError_t run_tx_mid(int n_fft, const SymParams &p, hipStream_t st, int num_cu);
hipError_t run_txframe_mid(int n_fft, const SymParams &p, const float2 *header, float header_max, hipStream_t st, int num_cu);
// fused estimate_channel + per-symbol demod for N = 64 frames with per-frame offset / CFO / live-symbol count
// final_out / final_stride / final_len (optional, 4-byte aligned): also do the length-header parse + truncate and write the
// payload bytes to their final place (no outer code), so that no separate finish kernel is needed
// frame_list / frame_count (device, optional): only the listed frames are processed
// cut_ws (device, optional, n_frames + 4 ints): the split form -- one launch with only the common frame body (four waves per SIMD) over
// every frame, one with only the capture-cut body over the frames the first one left on the list in cut_ws
hipError_t run_rxframe64(const SymParams &p, float2 *hk_out, hipStream_t st, int num_cu, unsigned char *final_out = nullptr,
                         long long final_stride = 0, int32_t *final_len = nullptr, const int32_t *frame_list = nullptr,
                         const int32_t *frame_count = nullptr, int32_t *cut_ws = nullptr);
// fused encode for N = 64 (map + IFFT + CP + header + normalise, one HBM pass); hipErrorNotSupported outside its envelope
hipError_t run_txframe64(const SymParams &p, const float2 *header, float header_max, hipStream_t st, int num_cu);

hipError_t run_fft(int n, const SymParams &p, bool inverse, hipStream_t st, int num_cu);
hipError_t run_ifft_cp(int n, const SymParams &p, hipStream_t st, int num_cu);
hipError_t run_demod(int n, const SymParams &p, hipStream_t st, int num_cu);
hipError_t run_chest(int n, const SymParams &p, hipStream_t st, int num_cu);
hipError_t run_tx_symbols(int n, const SymParams &p, hipStream_t st, int num_cu);

// ---- Schmidl-Cox (kernels_sync.hip)
struct ScParams {
    const Tuning *tune = nullptr;
    Trace *trace = nullptr;
    const float2 *in = nullptr;
    long long n_frames = 0, frame_stride = 0, frame_len = 0;
    long long n_lags = 0;        // lags searched per frame (already clipped to the valid range)
    int L = 0, W = 0;            // period and window
    double threshold = 0.5;
    bool tail_mapped = false;    // the sample behind every frame's last one is mapped memory (rows with slack, or not the batch's last row)
    int tiles_per_frame = 1;
    int mode = 0;                // 0 fused (single tile), 1 first-crossing only, 2 peak search from lag_base
    const int32_t *lag_base = nullptr; // mode 2: per-frame first lag (d1), -1 = skip
    long long *cross = nullptr;        // mode 1: per-tile first crossing (or LLONG_MAX)
    int32_t *d_hat = nullptr;
    double *f_delta = nullptr;
    float *metric = nullptr;
    const int32_t *slow_list = nullptr;  // k_sc_tile list mode: frames to (re)do, count in *slow_count
    const int32_t *slow_count = nullptr;
    ScStats *stats = nullptr;            // optional: where the search leaves the addresses of its list counters
};
struct ScExact { double pr, pi, num, den; }; // exact sums at the chosen lag; k_sc_post turns them into CFO and metric
size_t sc_lds_bytes(const ScParams &p);
// L = 80, W = 240 (N = 64): every lag on f64 prefix differences, one streaming pass that stops when the peak window has closed
// (kernels_sc80.hip).  Leaves d_hat and the exact sums of the chosen lag (k_sc_post's input); frames it does not trust go to slow_list.
bool sc80_ok(const ScParams &p);
bool sc80_wanted(const ScParams &p);   // sc80_ok and enough frames (or short enough slots) for one row per frame to pay
hipError_t launch_sc80(const ScParams &p, ScExact *exact, int32_t *slow_list, int32_t *slow_count, int num_cu, hipStream_t st);
hipError_t run_sc(const ScParams &p, hipStream_t st);
// fast path for one-tile frames with a short period (f32 filter + exact f64 decisions; kernels_sync.hip)
bool sc_fast_ok(const ScParams &p);
size_t sc_fast_workspace_bytes(long long n_frames, long long n_lags);
hipError_t run_sc_fast(const ScParams &p, void *workspace, int num_cu, hipStream_t st);
// periods L = 160 .. 1280 (N = 128 .. 1024): one streaming pass per frame that stops once the decision is determined
// (kernels_scstream.hip)
bool sc_stream_ok(const ScParams &p);
hipError_t run_sc_stream(const ScParams &p, int num_cu, hipStream_t st);
// long periods (N >= 128): chunk sums + bounded exact search (kernels_scbig.hip)
struct ScBigParams {
    const float2 *in = nullptr;
    long long n_frames = 0, frame_stride = 0, frame_len = 0, n_lags = 0;
    int L = 0, W = 0, C = 0;       // period, window, chunk = L / 8
    int nch = 0, nch_pad = 0;      // chunks per frame; row length of the workspace (>= nch + 1)
    int tiles_per_frame = 0;
    double threshold = 0.5;
    double *ws = nullptr;          // [n_frames][3][nch_pad]: chunk totals, then exclusive prefixes, of q.re, q.im, e
    int32_t *d_hat = nullptr;
    double *f_delta = nullptr;
    float *metric = nullptr;
};
bool sc_big_ok(const ScParams &p);
size_t sc_big_workspace_bytes(const ScParams &p);
hipError_t run_sc_big(const ScParams &p, void *workspace, int num_cu, hipStream_t st);
hipError_t run_sc_min_cross(const long long *cross, int tiles_per_frame, long long n_frames, int32_t *d1, hipStream_t st);
// base (optional): per-pair sample offset added to the pair's start; status (optional): pairs with status != 0 get 0
hipError_t run_freq_correction(const float2 *in, long long n_pairs, long long stride, long long right_offset, int L,
                               double *f_delta, hipStream_t st, const int32_t *base = nullptr, const int32_t *status = nullptr);
// xcorr_fft (src/signals/mod.rs:186-217) of every capture a[f] (N samples) with b (nb samples): idx_max into the fft_shifted
// 2N - 1 output, |.| at it, optionally the whole output
size_t xcorr_workspace_bytes(long long n_frames, long long N, int nb);
hipError_t run_xcorr(const float2 *a, long long n_frames, long long stride, long long N, const float2 *b, int nb, void *workspace,
                     int32_t *idx_max, float *peak, float2 *out, long long out_stride, int num_cu, hipStream_t st);
// reference timing -> trimmed offset (src/receiver.rs:21-36): offset = idx_max - N.  offset_report (optional) receives that
// value as it is (negative for quirk Q1), offset_kernels the copy the receive kernels index with (0 unless status == 0)
hipError_t run_rx_prepare_ref(long long n_frames, const int32_t *idx_max, long long frame_len, int L, int max_symbols,
                              int bytes_per_symbol, int32_t *status, int32_t *offset_report, int32_t *offset_kernels, int32_t *nsym,
                              hipStream_t st);
hipError_t run_cfo_rotate(float2 *x, long long n_frames, long long frame_stride, long long frame_len,
                          const double *f_delta, const int32_t *first_index, hipStream_t st);

// ---- byte / elementwise kernels (kernels_bytes.hip)
hipError_t run_qam_map(const uint8_t *bytes, long long n_bytes, int bps, float2 *out, hipStream_t st);
hipError_t run_qam_demap(const float2 *sym, long long n_sym, int bps, uint8_t *bytes, uint8_t *idx, hipStream_t st);
hipError_t run_encode_block(const float2 *data, float2 *bins, long long n_sym, int n_fft, int guard, hipStream_t st);
hipError_t run_frame_max(const float2 *x, long long n_frames, long long frame_stride, long long frame_len,
                         unsigned *frame_max, hipStream_t st);
hipError_t run_frame_scale(float2 *x, long long n_frames, long long frame_stride, long long frame_len,
                           const unsigned *frame_max, hipStream_t st);
hipError_t run_ham_encode(const uint8_t *in, long long n_frames, long long in_stride, const int32_t *in_len,
                          long long n_bytes, uint8_t *out, long long out_stride, int32_t *out_len, hipStream_t st);
hipError_t run_ham_decode(const uint8_t *in, long long n_bytes, uint8_t *out, uint32_t *corrected, hipStream_t st);
// TX finish: write the 10 header blocks and divide the frame by its max (transmitter.rs:183-194)
hipError_t run_tx_finish(float2 *out, long long n_frames, long long out_stride, int header_len, long long frame_len,
                         const float2 *header, float header_max, const unsigned *frame_max, hipStream_t st);
// RX prepare: d_hat -> status / offset / live symbols / CFO (receiver.rs:21-39)
hipError_t run_rx_prepare(long long n_frames, const int32_t *d_hat, double *f_delta, long long frame_len, int L,
                          int backoff, int cfo_mode, int max_symbols, int bytes_per_symbol, int32_t *status,
                          int32_t *offset, int32_t *nsym, hipStream_t st, const int32_t *frame_list = nullptr,
                          const int32_t *frame_count = nullptr);
// one capture's timing written where the search would have left it (ofdm_abi_rx_decode_known)
hipError_t run_set_sync(int32_t *d_hat, double *f_delta, float *metric, int32_t d, double fd, float m, hipStream_t st);
// RX finish: header parse + truncate [+ Hamming decode] (receiver.rs:85-95)
hipError_t run_rx_finish(const uint8_t *raw, long long raw_stride, long long n_frames, const int32_t *status,
                         const int32_t *nsym, int bytes_per_symbol, int ecc, uint8_t *out, long long out_stride,
                         int32_t *out_len, hipStream_t st, const int32_t *frame_list = nullptr,
                         const int32_t *frame_count = nullptr);

// channel (src/channel.rs:33-74) on the GPU (kernels_bytes.hip)
struct ChannelParams {
    const float2 *tx = nullptr;
    long long n_frames = 0, tx_stride = 0, tx_len = 0;
    double snr_lin = 1000.0;
    int timing_error = 0;
    unsigned long long seed = 0;
    const int32_t *delay = nullptr;      // optional test-bench extension: per-frame placement inside the slot
    const double *f_delta_in = nullptr;  // optional test-bench extension: per-frame CFO instead of the drawn one
    float2 *out = nullptr;
    long long out_stride = 0, out_len = 0;
    double *f_delta_out = nullptr;
    int n_taps = 0;                      // the non-zero taps of CHANNEL
    int tap_idx[16] = {0};
    float tap_val[16] = {0};
};
hipError_t run_channel(const ChannelParams &p, int num_cu, hipStream_t st);

// measurement helper (ofdm_hbm_read_probe): read-only stream in k_demod64's access pattern (0), over whole symbols (1), or
// with unit-stride 16-byte loads (2)
hipError_t run_read_probe(const float2 *in, long long n_sym, int pattern, unsigned *sink, int num_cu, hipStream_t st);

} // namespace ofdm
