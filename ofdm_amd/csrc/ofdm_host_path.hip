// ofdm_host_path.hip -- the host-buffer side of the C ABI (include/ofdm_hip.h, "host buffers" and "one long capture").
//
// The reference's encode / decode own HOST vectors (encode returns a Vec<Complex64>, src/transmitter.rs:11-15; decode consumes one,
// src/receiver.rs:9-13) and its receiver example hands ONE 2 M-sample buffer to each decode! (examples/jetson_rx.rs:15-17,48-49,
// 84-86).  The device-buffer entry points of ofdm_abi.hip leave staging to the caller; these do it inside the library:
//   * ofdm_rx_decode_host / ofdm_rx_demod_host / ofdm_tx_encode_host: the batch is cut into chunks that travel through three slots --
//     H2D(k+1) on a copy stream, the kernels of chunk k on the context's stream, D2H(k-1) on a second copy stream, ordered by events.
//     Pinned caller memory (ofdm_host_alloc / ofdm_host_register) is DMA-ed in place; pageable memory goes through pinned bounce
//     buffers that the calling thread fills while the previous chunk's DMA runs.
//   * ofdm_sc_correlate_long / ofdm_rx_decode_long[_host]: one long capture is searched as a BATCH of overlapping slices (own lags +
//     a read-only halo of 2W + L samples, the single-GPU form of SURVEY 8(e)'s halo split), so that a 2 M-sample buffer fills the
//     chip instead of one workgroup; the lowest slice with a threshold crossing in its OWN lags is then searched again with its whole
//     peak window, which makes the answer the one a single search over the whole capture gives.
#include "ofdm_ctx.hpp"

#include <algorithm>
#include <cstring>
#include <new>

using namespace ofdm;

namespace {
struct Buf {
    void *p = nullptr;
    size_t cap = 0;
};
} // namespace

struct HostPipe {
    static constexpr int kSlots = 3;
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t in_done[kSlots] = {}, k_done[kSlots] = {}, out_done[kSlots] = {};
    Buf d_in[kSlots], d_out[kSlots]; // device slots
    Buf h_in[kSlots], h_out[kSlots]; // pinned bounce slots
    Buf d_long;                      // device copy of one long capture (ofdm_rx_decode_long_host)
    Buf d_scal, h_scal;              // per-slice detector outputs / per-frame scalars (device, pinned)
};

namespace {

int sync_all(ofdm_ctx *c, HostPipe *hp) {
    HIP_TRY(c, hipStreamSynchronize(hp->s_in));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipStreamSynchronize(hp->s_out));
    return OFDM_OK;
}

int dev_grow(ofdm_ctx *c, HostPipe *hp, Buf &b, size_t bytes) {
    if (bytes <= b.cap) return OFDM_OK;
    if (b.p) { int rc = sync_all(c, hp); if (rc) return rc; HIP_TRY(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    const size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) { c->last_hip = (int)e; b.p = nullptr; return OFDM_ERR_NOMEM; }
    b.cap = want;
    return OFDM_OK;
}
int pin_grow(ofdm_ctx *c, HostPipe *hp, Buf &b, size_t bytes) {
    if (bytes <= b.cap) return OFDM_OK;
    if (b.p) { int rc = sync_all(c, hp); if (rc) return rc; HIP_TRY(c, hipHostFree(b.p)); b.p = nullptr; b.cap = 0; }
    const size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipHostMalloc(&b.p, want, hipHostMallocDefault);
    if (e != hipSuccess) { c->last_hip = (int)e; b.p = nullptr; return OFDM_ERR_NOMEM; }
    b.cap = want;
    return OFDM_OK;
}

int pipe_get(ofdm_ctx *c, HostPipe **out) {
    if (c->pipe) { *out = c->pipe; return OFDM_OK; }
    HostPipe *hp = new (std::nothrow) HostPipe();
    if (!hp) return OFDM_ERR_NOMEM;
    // the pipe is published only when every stream and event exists: a failed creation must not leave a half-built pipe that later
    // host calls would pick up with null streams; ofdm_host_pipe_destroy tolerates the partial one
    auto build = [&]() -> int {
        HIP_TRY(c, hipStreamCreateWithFlags(&hp->s_in, hipStreamNonBlocking));
        HIP_TRY(c, hipStreamCreateWithFlags(&hp->s_out, hipStreamNonBlocking));
        for (int s = 0; s < HostPipe::kSlots; s++) {
            HIP_TRY(c, hipEventCreateWithFlags(&hp->in_done[s], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&hp->k_done[s], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&hp->out_done[s], hipEventDisableTiming));
        }
        return OFDM_OK;
    };
    const int rc = build();
    c->pipe = hp;
    if (rc) { ofdm_host_pipe_destroy(c); return rc; } // frees what was created and clears c->pipe
    *out = hp;
    return OFDM_OK;
}

// Is [p, p + bytes) page-locked memory the DMA engines can read in place (hipHostMalloc / hipHostRegister)?
bool host_pinned(const void *p, size_t bytes) {
    if (!p || !bytes) return true;
    auto one = [](const void *q) {
        hipPointerAttribute_t a;
        std::memset(&a, 0, sizeof(a));
        if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; } // pageable: not an error of ours
        return a.type == hipMemoryTypeHost;
    };
    return one(p) && one(static_cast<const char *>(p) + bytes - 1);
}

// One pipelined pass over the chunks of a job.
struct PipeJob {
    int64_t n_chunks = 0;
    virtual ~PipeJob() {}
    virtual size_t in_bytes(int64_t k) const = 0;
    virtual size_t out_bytes(int64_t k) const = 0;
    virtual const void *in_direct(int64_t) const { return nullptr; } // non-null: pinned source, DMA-ed in place
    virtual void fill_in(int64_t, void *) const {}                   // else: pack the chunk's input into the pinned bounce slot
    virtual void *out_direct(int64_t) const { return nullptr; }      // non-null: pinned destination of the whole output block
    virtual void drain_out(int64_t, const void *) const {}           // else: scatter the bounce slot into the caller's arrays
    virtual int launch(int64_t k, void *d_in, void *d_out) = 0;      // enqueue chunk k's kernels on the context's stream
};

int run_pipe(ofdm_ctx *c, PipeJob &job) {
    if (job.n_chunks <= 0) return OFDM_OK;
    HostPipe *hp;
    int rc = pipe_get(c, &hp);
    if (rc) return rc;
    size_t max_in = 0, max_out = 0;
    bool bounce_in = false, bounce_out = false;
    for (int64_t k : {(int64_t)0, job.n_chunks - 1}) { // every chunk but the last has chunk 0's shape
        max_in = std::max(max_in, job.in_bytes(k));
        max_out = std::max(max_out, job.out_bytes(k));
        bounce_in = bounce_in || !job.in_direct(k);
        bounce_out = bounce_out || !job.out_direct(k);
    }
    const int slots = (int)std::min<int64_t>(HostPipe::kSlots, job.n_chunks);
    for (int s = 0; s < slots; s++) {
        if ((rc = dev_grow(c, hp, hp->d_in[s], max_in + 64))) return rc; // + 64: the frame kernels' branch-free prefetches may read a little past a row
        if ((rc = dev_grow(c, hp, hp->d_out[s], max_out + 64))) return rc;
        if (bounce_in && (rc = pin_grow(c, hp, hp->h_in[s], max_in))) return rc;
        if (bounce_out && (rc = pin_grow(c, hp, hp->h_out[s], max_out))) return rc;
    }
    int64_t drained = 0;
    auto drain = [&](int64_t k) -> int { // chunk k's outputs are on the host: hand them to the caller (this also frees slot k % kSlots)
        const int s = (int)(k % HostPipe::kSlots);
        HIP_TRY(c, hipEventSynchronize(hp->out_done[s]));
        if (!job.out_direct(k)) job.drain_out(k, hp->h_out[s].p);
        return OFDM_OK;
    };
    auto body = [&]() -> int {
        for (int64_t k = 0; k < job.n_chunks; k++) {
            const int s = (int)(k % HostPipe::kSlots);
            if (k >= HostPipe::kSlots) { // the slot's previous chunk must be completely through (its D2H is the last user of the slot)
                int r = drain(k - HostPipe::kSlots);
                if (r) return r;
                drained = k - HostPipe::kSlots + 1;
            }
            const void *src = job.in_direct(k);
            if (!src) { job.fill_in(k, hp->h_in[s].p); src = hp->h_in[s].p; } // overlaps the DMA of chunk k - 1
            HIP_TRY(c, hipMemcpyAsync(hp->d_in[s].p, src, job.in_bytes(k), hipMemcpyHostToDevice, hp->s_in));
            HIP_TRY(c, hipEventRecord(hp->in_done[s], hp->s_in));
            HIP_TRY(c, hipStreamWaitEvent(c->stream, hp->in_done[s], 0));
            int r = job.launch(k, hp->d_in[s].p, hp->d_out[s].p);
            if (r) return r;
            HIP_TRY(c, hipEventRecord(hp->k_done[s], c->stream));
            HIP_TRY(c, hipStreamWaitEvent(hp->s_out, hp->k_done[s], 0));
            void *dst = job.out_direct(k);
            if (!dst) dst = hp->h_out[s].p;
            HIP_TRY(c, hipMemcpyAsync(dst, hp->d_out[s].p, job.out_bytes(k), hipMemcpyDeviceToHost, hp->s_out));
            HIP_TRY(c, hipEventRecord(hp->out_done[s], hp->s_out));
        }
        for (int64_t k = drained; k < job.n_chunks; k++) {
            int r = drain(k);
            if (r) return r;
        }
        return OFDM_OK;
    };
    rc = body();
    if (rc) sync_all(c, hp); // nothing may still be reading or writing the caller's memory when the error is returned
    return rc;
}

int64_t auto_chunk(int64_t n_frames, size_t bytes_per_frame, int64_t chunk_frames) {
    if (chunk_frames <= 0) { // ~48 MB of the large side per chunk: a millisecond of PCIe time against tens of microseconds of launches
        chunk_frames = (int64_t)((size_t)48 << 20) / (int64_t)std::max<size_t>(bytes_per_frame, 1);
        if (chunk_frames > 4) chunk_frames &= ~(int64_t)3;
    }
    return std::min(std::max<int64_t>(chunk_frames, 1), std::max<int64_t>(n_frames, 1));
}

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------- decode(host) = ofdm_rx_decode_batch per chunk
struct DecodeJob : PipeJob {
    ofdm_ctx *c;
    const ofdm_fc32 *in;
    int64_t n_frames, stride, frame_len, n_lags, chunk;
    int32_t max_symbols;
    uint8_t *out; int64_t out_stride;
    int32_t *out_len, *status, *offset; double *fd; float *metric;
    size_t ob; // device row bytes
    bool pinned_in;
    int64_t f0(int64_t k) const { return k * chunk; }
    int64_t m(int64_t k) const { return std::min(chunk, n_frames - k * chunk); }
    size_t in_bytes(int64_t k) const override {
        const int64_t last = f0(k) + m(k) == n_frames; // only the batch's last frame may end before its stride does
        return (size_t)((m(k) - (last ? 1 : 0)) * stride + (last ? frame_len : 0)) * sizeof(ofdm_fc32);
    }
    size_t out_bytes(int64_t k) const override { return (size_t)m(k) * (24 + ob); }
    const void *in_direct(int64_t k) const override { return pinned_in ? in + f0(k) * stride : nullptr; }
    void fill_in(int64_t k, void *p) const override { std::memcpy(p, in + f0(k) * stride, in_bytes(k)); }
    int launch(int64_t k, void *d_in, void *d_out) override {
        const int64_t n = m(k);
        char *o = static_cast<char *>(d_out); // [f_delta f64 | len | status | offset i32 | metric f32 | rows]
        return ofdm_rx_decode_batch(c, static_cast<const ofdm_fc32 *>(d_in), n, stride, frame_len, n_lags, max_symbols,
                                    reinterpret_cast<uint8_t *>(o + 24 * n), (int64_t)ob, reinterpret_cast<int32_t *>(o + 8 * n),
                                    reinterpret_cast<int32_t *>(o + 12 * n), reinterpret_cast<int32_t *>(o + 16 * n),
                                    reinterpret_cast<double *>(o), reinterpret_cast<float *>(o + 20 * n));
    }
    void drain_out(int64_t k, const void *p) const override {
        const int64_t n = m(k), a = f0(k);
        const char *o = static_cast<const char *>(p);
        const int32_t *len = reinterpret_cast<const int32_t *>(o + 8 * n);
        if (fd) std::memcpy(fd + a, o, 8 * (size_t)n);
        std::memcpy(out_len + a, len, 4 * (size_t)n);
        std::memcpy(status + a, o + 12 * n, 4 * (size_t)n);
        if (offset) std::memcpy(offset + a, o + 16 * n, 4 * (size_t)n);
        if (metric) std::memcpy(metric + a, o + 20 * n, 4 * (size_t)n);
        const uint8_t *rows = reinterpret_cast<const uint8_t *>(o + 24 * n);
        if ((size_t)out_stride == ob) std::memcpy(out + a * out_stride, rows, (size_t)n * ob);
        else
            for (int64_t f = 0; f < n; f++) // a row holds out_len[f] meaningful bytes
                std::memcpy(out + (a + f) * out_stride, rows + (size_t)f * ob, (size_t)std::min<int64_t>(std::max(len[f], 0), out_stride));
    }
};

// ---------------------------------------------------------------- rx_demod(host) = ofdm_rx_demod_batch per chunk (regular streams)
struct DemodJob : PipeJob {
    ofdm_ctx *c;
    const ofdm_fc32 *in;
    int64_t n_frames, stride, frame_len, chunk;
    int32_t first_symbol, syms;
    uint8_t *out; int64_t out_stride;
    size_t nb;
    bool pinned_in, pinned_out;
    int64_t f0(int64_t k) const { return k * chunk; }
    int64_t m(int64_t k) const { return std::min(chunk, n_frames - k * chunk); }
    size_t in_bytes(int64_t k) const override {
        const int64_t last = f0(k) + m(k) == n_frames;
        return (size_t)((m(k) - (last ? 1 : 0)) * stride + (last ? frame_len : 0)) * sizeof(ofdm_fc32);
    }
    size_t out_bytes(int64_t k) const override { return (size_t)m(k) * nb; }
    const void *in_direct(int64_t k) const override { return pinned_in ? in + f0(k) * stride : nullptr; }
    void fill_in(int64_t k, void *p) const override { std::memcpy(p, in + f0(k) * stride, in_bytes(k)); }
    void *out_direct(int64_t k) const override { return pinned_out ? out + f0(k) * out_stride : nullptr; }
    int launch(int64_t k, void *d_in, void *d_out) override {
        return ofdm_rx_demod_batch(c, static_cast<const ofdm_fc32 *>(d_in), m(k), stride, frame_len, first_symbol, syms, nullptr, nullptr,
                                   nullptr, 0, static_cast<uint8_t *>(d_out), (int64_t)nb, nullptr);
    }
    void drain_out(int64_t k, const void *p) const override {
        const uint8_t *rows = static_cast<const uint8_t *>(p);
        if ((size_t)out_stride == nb) { std::memcpy(out + f0(k) * out_stride, rows, (size_t)m(k) * nb); return; }
        for (int64_t f = 0; f < m(k); f++) std::memcpy(out + (f0(k) + f) * out_stride, rows + (size_t)f * nb, nb);
    }
};

// ---------------------------------------------------------------- encode(host) = ofdm_tx_encode_batch per chunk
struct EncodeJob : PipeJob {
    ofdm_ctx *c;
    const uint8_t *payload; int64_t n_frames, payload_stride, chunk;
    const int32_t *lens; int32_t payload_bytes;
    ofdm_fc32 *out; int64_t out_stride, frame;
    bool pinned_out;
    int64_t f0(int64_t k) const { return k * chunk; }
    int64_t m(int64_t k) const { return std::min(chunk, n_frames - k * chunk); }
    size_t lens_bytes(int64_t k) const { return lens ? round_up(4 * (size_t)m(k), 16) : 0; }
    size_t in_bytes(int64_t k) const override { return lens_bytes(k) + (size_t)m(k) * (size_t)payload_bytes; }
    size_t out_bytes(int64_t k) const override { return (size_t)m(k) * (size_t)frame * sizeof(ofdm_fc32); }
    void fill_in(int64_t k, void *p) const override { // [lens | rows of payload_bytes]: a row is read for its own length only
        char *b = static_cast<char *>(p);
        if (lens) { // clamped to the row the library stages: a length beyond payload_bytes must not walk into the next row
            int32_t *dl = reinterpret_cast<int32_t *>(b);
            for (int64_t f = 0; f < m(k); f++) dl[f] = std::min<int32_t>(std::max<int32_t>(lens[f0(k) + f], 0), payload_bytes);
        }
        b += lens_bytes(k);
        for (int64_t f = 0; f < m(k); f++) {
            int64_t n = lens ? lens[f0(k) + f] : payload_bytes;
            n = std::min<int64_t>(std::max<int64_t>(n, 0), payload_bytes);
            std::memcpy(b + (size_t)f * payload_bytes, payload + (f0(k) + f) * payload_stride, (size_t)n);
        }
    }
    void *out_direct(int64_t k) const override { return pinned_out ? out + f0(k) * out_stride : nullptr; }
    int launch(int64_t k, void *d_in, void *d_out) override {
        const char *b = static_cast<const char *>(d_in);
        return ofdm_tx_encode_batch(c, reinterpret_cast<const uint8_t *>(b + lens_bytes(k)), m(k), payload_bytes,
                                    lens ? reinterpret_cast<const int32_t *>(b) : nullptr, payload_bytes, static_cast<ofdm_fc32 *>(d_out), frame);
    }
    void drain_out(int64_t k, const void *p) const override {
        const ofdm_fc32 *rows = static_cast<const ofdm_fc32 *>(p);
        if (out_stride == frame) { std::memcpy(out + f0(k) * out_stride, rows, out_bytes(k)); return; }
        for (int64_t f = 0; f < m(k); f++) std::memcpy(out + (f0(k) + f) * out_stride, rows + f * frame, (size_t)frame * sizeof(ofdm_fc32));
    }
};

// ---------------------------------------------------------------- one long capture
struct LongGeom {
    int L, W;
    int64_t valid, lo, hi, halo, own, n_full, tail_lo, tail_len;
};
// Lags [lo, hi) of a capture of n samples as n_full slices of `own` lags (frame i = samples [lo + i own, + own + halo)) and one
// tail frame [tail_lo, tail_lo + tail_len) for whatever does not make a whole slice inside the capture.
bool long_geometry(const ofdm_ctx *c, int64_t n, int64_t lag_lo, int64_t lag_hi, int64_t slice_lags, LongGeom &g) {
    g.L = c->S(); g.W = c->prm.sync_window_reps * g.L;
    g.valid = n - g.W - g.L + 1;
    g.lo = std::max<int64_t>(lag_lo, 0);
    g.hi = (lag_hi <= 0 || lag_hi > g.valid) ? g.valid : lag_hi;
    if (g.valid <= 0 || g.lo >= g.hi) return false;
    g.halo = 2LL * g.W + g.L; // W more lags (a window that opens at the last own lag) + W + L - 1 samples under the last of them, made even
    int64_t own = slice_lags;
    if (own <= 0) {
        // N = 64: the one-tile kernel's 2560-sample tile, all of it; longer periods: ~1000 slices, but never more halo than own lags
        own = g.L == 80 ? 2560 - g.halo : std::max((g.hi - g.lo) / 1024, g.halo);
    }
    own = std::max<int64_t>((own + 1) & ~(int64_t)1, 2);
    g.own = own;
    g.n_full = 0;
    const int64_t by_lags = (g.hi - g.lo) / own;
    const int64_t by_samples = (n - g.lo - g.halo) / own; // frame i ends at lo + (i + 1) own + halo <= n
    g.n_full = std::max<int64_t>(0, std::min(by_lags, by_samples));
    g.tail_lo = g.lo + g.n_full * own;
    g.tail_len = g.tail_lo < g.hi ? std::min(n - g.tail_lo, (g.hi - g.tail_lo) + g.halo) : 0;
    return true;
}

// Pass 1 of the long search: which slice holds the first threshold crossing of [lag_lo, lag_hi) among its OWN lags.  On a hit
// *hit >= 0 and *d_clip = the capture's lag of that slice's peak with the window CLIPPED to the slice's own lags, so
// d_clip - W <= first crossing <= d_clip <= true peak.  Leaves the per-slice outputs in hp->d_scal (s_fd | s_d | s_m).
struct LongHit { int64_t hit = -1, d_clip = -1, start = 0, flen = 0, own = 0, lags2 = 0, ns = 0; };
int sc_long_find(ofdm_ctx *c, HostPipe *hp, const float2 *in, int64_t n, const LongGeom &g, LongHit &h) {
    const int64_t ns = g.n_full + (g.tail_len > 0 ? 1 : 0);
    h.ns = ns;
    if (ns > 0x7fffffff) return OFDM_ERR_INVALID;
    int rc;
    if ((rc = dev_grow(c, hp, hp->d_scal, (size_t)ns * 16 + 64))) return rc;
    if ((rc = pin_grow(c, hp, hp->h_scal, (size_t)ns * 4 + 64))) return rc;
    double *s_fd = static_cast<double *>(hp->d_scal.p);                       // [ns] f64 | [ns] i32 | [ns] f32
    int32_t *s_d = reinterpret_cast<int32_t *>(s_fd + ns);
    float *s_m = reinterpret_cast<float *>(s_d + ns);
    // n_lags = own clips the peak window, so only "d_hat >= 0" and the bracket above are used.  One launch over every lag: a long
    // capture is mostly noise, where the two-launch search of the batch path would read everything twice.
    const int first_lags = c->tune.sc_first_lags;
    c->tune.sc_first_lags = 0;
    rc = OFDM_OK;
    if (g.n_full) rc = ofdm_abi_sc_run(c, in + g.lo, g.n_full, g.own, g.own + g.halo, g.own, s_d, s_fd, s_m);
    if (!rc && g.tail_len > 0)
        rc = ofdm_abi_sc_run(c, in + g.tail_lo, 1, g.tail_len, g.tail_len, g.hi - g.tail_lo, s_d + g.n_full, s_fd + g.n_full, s_m + g.n_full);
    c->tune.sc_first_lags = first_lags;
    if (rc) return rc;
    int32_t *h_d = static_cast<int32_t *>(hp->h_scal.p);
    HIP_TRY(c, hipMemcpyAsync(h_d, s_d, 4 * (size_t)ns, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < ns; i++)
        if (h_d[i] >= 0) { h.hit = i; break; }
    if (h.hit < 0) return OFDM_OK;
    const bool tail = h.hit == g.n_full;
    h.start = tail ? g.tail_lo : g.lo + h.hit * g.own;
    h.flen = tail ? g.tail_len : g.own + g.halo;
    h.own = tail ? g.hi - g.tail_lo : g.own;
    h.lags2 = std::min(h.own + g.W, g.valid - h.start); // the slice's lags with the whole peak window, as far as the capture has them
    h.d_clip = h.start + h_d[h.hit];
    return OFDM_OK;
}

// The detection whose first threshold crossing lies in [lag_lo, lag_hi): -1, or the capture's lag with CFO and metric.
int sc_long(ofdm_ctx *c, const float2 *in, int64_t n, int64_t lag_lo, int64_t lag_hi, int64_t slice_lags, int64_t *d_hat, double *f_delta,
            float *metric) {
    *d_hat = -1;
    if (f_delta) *f_delta = 0.0;
    if (metric) *metric = 0.f;
    LongGeom g;
    if (!long_geometry(c, n, lag_lo, lag_hi, slice_lags, g)) return OFDM_OK;
    HostPipe *hp;
    int rc = pipe_get(c, &hp);
    if (rc) return rc;
    LongHit hit;
    if ((rc = sc_long_find(c, hp, in, n, g, hit))) return rc;
    if (hit.hit < 0) return OFDM_OK;
    double *s_fd = static_cast<double *>(hp->d_scal.p);
    int32_t *s_d = reinterpret_cast<int32_t *>(s_fd + hit.ns);
    float *s_m = reinterpret_cast<float *>(s_d + hit.ns);
    // pass 2: that slice again with its whole peak window (a slice that ends with the capture's last lag had it in pass 1 already)
    struct { double fd; int32_t d; float m; } h;
    h.d = (int32_t)(hit.d_clip - hit.start);
    const bool again = hit.lags2 > hit.own;
    const int64_t at = again ? 0 : hit.hit;
    if (again && (rc = ofdm_abi_sc_run(c, in + hit.start, 1, hit.flen, hit.flen, hit.lags2, s_d, s_fd, s_m))) return rc;
    HIP_TRY(c, hipMemcpyAsync(&h.fd, s_fd + at, 8, hipMemcpyDeviceToHost, c->stream));
    if (again) HIP_TRY(c, hipMemcpyAsync(&h.d, s_d, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&h.m, s_m + at, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (h.d < 0) return OFDM_OK; // cannot happen (pass 1 saw a crossing among these lags); keep "no detection" rather than a wrong lag
    *d_hat = hit.start + h.d;
    if (f_delta) *f_delta = h.fd;
    if (metric) *metric = h.m;
    return OFDM_OK;
}

int decode_long_dev(ofdm_ctx *c, const float2 *in, int64_t n, int64_t lag_lo, int64_t lag_hi, int64_t d_known, int32_t max_symbols,
                    uint8_t *out_dev, int64_t out_cap, int32_t *out_len, int32_t *status, int64_t *offset, double *f_delta,
                    float *metric) {
    *out_len = 0; *status = OFDM_FRAME_NOSYNC;
    if (offset) *offset = 0;
    if (f_delta) *f_delta = 0.0;
    if (metric) *metric = 0.f;
    HostPipe *hp;
    int rc = pipe_get(c, &hp);
    if (rc) return rc;
    int64_t start = 0, sub_lags = 0;
    LongGeom g;
    bool have_sync = false;          // lag_lo > 0: the detection comes from the slice search below, not from the chain's own search
    int64_t k_d = -1; double k_fd = 0.0; float k_m = 0.f;
    const bool whole = lag_lo == 0 && (lag_hi <= 0 || lag_hi >= n - (int64_t)(c->prm.sync_window_reps + 1) * c->S() + 1);
    const bool one_frame = whole && d_known < 0 && (!long_geometry(c, n, 0, 0, 0, g) || g.n_full == 0);
    if (c->prm.sync_mode == OFDM_SYNC_SCHMIDL_COX && !one_frame) { // (a capture no longer than one slice is one frame of the batch path)
        // Decode from a SUB-CAPTURE that starts at or before both the first crossing d1 and the trimmed frame start
        // (peak - L - backoff): its own search finds the same crossing, window, peak and offset as the whole capture's, and every
        // length the chain derives from "samples after the trimmed start" is unchanged.  A known peak d gives d1 >= d - W; pass 1 of
        // the slice search gives a clipped peak dc with dc - W <= d1 <= dc <= peak -- enough to place the sub-capture without
        // pass 2 (the chain's own search redoes that work on a few hundred samples anyway).
        const int L = c->S(), W = c->prm.sync_window_reps * L;
        int64_t d_lo = d_known, d_hi = d_known; // bracket of the first crossing: d1 in [d_lo - W, d_hi]
        if (d_known < 0 && lag_lo > 0) {
            // A partial lag range: the frame's trimmed start (peak - L - backoff) may lie in FRONT of lag_lo, so a sub-capture that holds the
            // whole frame also holds lags the search must not see (the tail or plateau of an earlier packet would win: ADVICE r4).  The
            // slice search is exact about "first crossing in [lag_lo, lag_hi)" (both passes); the chain then runs with that timing.
            if ((rc = sc_long(c, in, n, lag_lo, lag_hi, 0, &k_d, &k_fd, &k_m))) return rc;
            if (k_d < 0) return OFDM_OK;
            have_sync = true;
            start = std::max<int64_t>(k_d - L - c->prm.sync_backoff, 0) & ~(int64_t)1;
        } else {
        if (d_known < 0) {
            if (!long_geometry(c, n, lag_lo, lag_hi, 0, g)) return OFDM_OK;
            LongHit hit;
            if ((rc = sc_long_find(c, hp, in, n, g, hit))) return rc;
            if (hit.hit < 0) return OFDM_OK;
            d_lo = d_hi = hit.d_clip;
        }
        const int64_t back = (int64_t)W + L + c->prm.sync_backoff;
        start = std::max<int64_t>(d_lo - back, 0) & ~(int64_t)1; // even: the sub-capture stays 16-byte aligned for the LDS-DMA kernels
        sub_lags = d_hi - start + W + 2;                           // covers d1 + W; the search clips it to the capture's own last lag
        }
    } else if (c->prm.sync_mode != OFDM_SYNC_SCHMIDL_COX && !whole) {
        return OFDM_ERR_UNSUPPORTED; // the reference's detector is an argmax over the whole capture (src/receiver.rs:20-25)
    }
    if ((rc = dev_grow(c, hp, hp->d_scal, 64))) return rc;
    char *sc = static_cast<char *>(hp->d_scal.p); // f_delta f64 | len | status | offset i32 | metric f32
    if (have_sync)
        rc = ofdm_abi_rx_decode_known(c, reinterpret_cast<const ofdm_fc32 *>(in + start), n - start, (int32_t)(k_d - start), k_fd, k_m, max_symbols,
                                      out_dev, out_cap, reinterpret_cast<int32_t *>(sc + 8), reinterpret_cast<int32_t *>(sc + 12),
                                      reinterpret_cast<int32_t *>(sc + 16), reinterpret_cast<double *>(sc), reinterpret_cast<float *>(sc + 20));
    else
    rc = ofdm_rx_decode_batch(c, reinterpret_cast<const ofdm_fc32 *>(in + start), 1, n - start, n - start, sub_lags, max_symbols, out_dev,
                              out_cap, reinterpret_cast<int32_t *>(sc + 8), reinterpret_cast<int32_t *>(sc + 12),
                              reinterpret_cast<int32_t *>(sc + 16), reinterpret_cast<double *>(sc), reinterpret_cast<float *>(sc + 20));
    if (rc) return rc;
    struct { double fd; int32_t len, status, offset; float metric; } h;
    HIP_TRY(c, hipMemcpyAsync(&h, sc, 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *out_len = h.len; *status = h.status;
    if (offset) *offset = (h.status == OFDM_FRAME_BADTIMING ? 0 : start) + h.offset;
    if (f_delta) *f_delta = h.fd;
    if (metric) *metric = h.metric;
    return OFDM_OK;
}

} // namespace

extern "C" {

void ofdm_host_pipe_destroy(ofdm_ctx *c) {
    HostPipe *hp = c ? c->pipe : nullptr;
    if (!hp) return;
    if (hp->s_in) hipStreamSynchronize(hp->s_in);
    if (hp->s_out) hipStreamSynchronize(hp->s_out);
    for (int s = 0; s < HostPipe::kSlots; s++) {
        if (hp->d_in[s].p) hipFree(hp->d_in[s].p);
        if (hp->d_out[s].p) hipFree(hp->d_out[s].p);
        if (hp->h_in[s].p) hipHostFree(hp->h_in[s].p);
        if (hp->h_out[s].p) hipHostFree(hp->h_out[s].p);
        if (hp->in_done[s]) hipEventDestroy(hp->in_done[s]);
        if (hp->k_done[s]) hipEventDestroy(hp->k_done[s]);
        if (hp->out_done[s]) hipEventDestroy(hp->out_done[s]);
    }
    if (hp->d_long.p) hipFree(hp->d_long.p);
    if (hp->d_scal.p) hipFree(hp->d_scal.p);
    if (hp->h_scal.p) hipHostFree(hp->h_scal.p);
    if (hp->s_in) hipStreamDestroy(hp->s_in);
    if (hp->s_out) hipStreamDestroy(hp->s_out);
    delete hp;
    c->pipe = nullptr;
}

int ofdm_host_alloc(size_t bytes, void **host) {
    if (!host) return OFDM_ERR_INVALID;
    *host = nullptr;
    if (hipHostMalloc(host, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *host = nullptr; return OFDM_ERR_NOMEM; }
    return OFDM_OK;
}
int ofdm_host_free(void *host) {
    if (host && hipHostFree(host) != hipSuccess) { (void)hipGetLastError(); return OFDM_ERR_HIP; }
    return OFDM_OK;
}
int ofdm_host_register(void *host, size_t bytes) {
    if (!host || !bytes) return OFDM_ERR_INVALID;
    if (hipHostRegister(host, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return OFDM_ERR_HIP; }
    return OFDM_OK;
}
int ofdm_host_unregister(void *host) {
    if (!host) return OFDM_ERR_INVALID;
    if (hipHostUnregister(host) != hipSuccess) { (void)hipGetLastError(); return OFDM_ERR_HIP; }
    return OFDM_OK;
}
int ofdm_host_is_pinned(const void *host, size_t bytes) { return host_pinned(host, bytes) ? 1 : 0; }

int ofdm_rx_decode_host(ofdm_ctx *c, const ofdm_fc32 *in_host, int64_t n_frames, int64_t frame_stride, int64_t frame_len, int64_t n_lags,
                        int32_t max_symbols, uint8_t *out_host, int64_t out_stride, int32_t *out_len_host, int32_t *status_host,
                        int32_t *offset_host, double *f_delta_host, float *metric_host, int64_t chunk_frames) {
    if (!c || n_frames < 0 || frame_len <= 0 || max_symbols <= 0 || frame_stride < 0) return OFDM_ERR_INVALID;
    if (n_frames && (!in_host || !out_host || !out_len_host || !status_host)) return OFDM_ERR_INVALID;
    if (n_frames > 1 && frame_stride < frame_len) return OFDM_ERR_INVALID; // rows of a host batch do not overlap
    const int64_t raw = (int64_t)max_symbols * c->bytes_per_symbol();
    const int64_t body = raw > 16 ? raw - 16 : 0;
    const int64_t need = c->prm.ecc == OFDM_ECC_NONE ? body : (body / 7) * 4;
    if (out_stride < need) return OFDM_ERR_INVALID;
    if (!n_frames) return OFDM_OK;
    DeviceGuard dev_guard(c->device);
    DecodeJob j;
    j.c = c; j.in = in_host; j.n_frames = n_frames; j.stride = n_frames > 1 ? frame_stride : frame_len; j.frame_len = frame_len;
    j.n_lags = n_lags; j.max_symbols = max_symbols; j.out = out_host; j.out_stride = out_stride; j.out_len = out_len_host;
    j.status = status_host; j.offset = offset_host; j.fd = f_delta_host; j.metric = metric_host;
    j.ob = round_up((size_t)std::max<int64_t>(need, 4), 4);
    j.chunk = auto_chunk(n_frames, (size_t)j.stride * sizeof(ofdm_fc32), chunk_frames);
    j.n_chunks = (n_frames + j.chunk - 1) / j.chunk;
    j.pinned_in = host_pinned(in_host, (size_t)((n_frames - 1) * j.stride + frame_len) * sizeof(ofdm_fc32));
    return run_pipe(c, j);
}

int ofdm_rx_demod_host(ofdm_ctx *c, const ofdm_fc32 *in_host, int64_t n_frames, int64_t frame_stride, int64_t frame_len,
                       int32_t first_symbol, int32_t syms_per_frame, uint8_t *out_host, int64_t out_stride, int64_t chunk_frames) {
    if (!c || n_frames < 0 || frame_len <= 0 || syms_per_frame < 0 || first_symbol < 0 || frame_stride < 0) return OFDM_ERR_INVALID;
    if (n_frames && syms_per_frame && (!in_host || !out_host)) return OFDM_ERR_INVALID;
    if (n_frames > 1 && frame_stride < frame_len) return OFDM_ERR_INVALID;
    const int64_t nb = (int64_t)syms_per_frame * c->bytes_per_symbol();
    if (out_stride < nb) return OFDM_ERR_INVALID;
    if (!n_frames || !syms_per_frame) return OFDM_OK;
    DeviceGuard dev_guard(c->device);
    DemodJob j;
    j.c = c; j.in = in_host; j.n_frames = n_frames; j.stride = n_frames > 1 ? frame_stride : frame_len; j.frame_len = frame_len;
    j.first_symbol = first_symbol; j.syms = syms_per_frame; j.out = out_host; j.out_stride = out_stride; j.nb = (size_t)nb;
    j.chunk = auto_chunk(n_frames, (size_t)j.stride * sizeof(ofdm_fc32), chunk_frames);
    j.n_chunks = (n_frames + j.chunk - 1) / j.chunk;
    j.pinned_in = host_pinned(in_host, (size_t)((n_frames - 1) * j.stride + frame_len) * sizeof(ofdm_fc32));
    j.pinned_out = out_stride == nb && host_pinned(out_host, (size_t)(n_frames * nb));
    return run_pipe(c, j);
}

int ofdm_tx_encode_host(ofdm_ctx *c, const uint8_t *payload_host, int64_t n_frames, int64_t payload_stride, const int32_t *payload_len_host,
                        int32_t payload_bytes, ofdm_fc32 *out_host, int64_t out_stride, int64_t chunk_frames) {
    if (!c || n_frames < 0 || payload_bytes < 0 || payload_stride < 0) return OFDM_ERR_INVALID;
    if (n_frames && (!out_host || (payload_bytes && !payload_host))) return OFDM_ERR_INVALID;
    const int64_t frame = ofdm_frame_samples(c, payload_bytes);
    if (out_stride < frame) return OFDM_ERR_INVALID;
    if (!n_frames) return OFDM_OK;
    DeviceGuard dev_guard(c->device);
    EncodeJob j;
    j.c = c; j.payload = payload_host; j.n_frames = n_frames; j.payload_stride = payload_stride; j.lens = payload_len_host;
    j.payload_bytes = payload_bytes; j.out = out_host; j.out_stride = out_stride; j.frame = frame;
    j.chunk = auto_chunk(n_frames, (size_t)frame * sizeof(ofdm_fc32), chunk_frames);
    j.n_chunks = (n_frames + j.chunk - 1) / j.chunk;
    j.pinned_out = out_stride == frame && host_pinned(out_host, (size_t)(n_frames * frame) * sizeof(ofdm_fc32));
    return run_pipe(c, j);
}

int ofdm_sc_correlate_long(ofdm_ctx *c, const ofdm_fc32 *in_dev, int64_t n_samples, int64_t lag_lo, int64_t lag_hi, int64_t slice_lags,
                           int64_t *d_hat, double *f_delta, float *metric) {
    if (!c || !d_hat || n_samples < 0 || lag_lo < 0 || slice_lags < 0 || (n_samples && !in_dev)) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    return sc_long(c, reinterpret_cast<const float2 *>(in_dev), n_samples, lag_lo, lag_hi, slice_lags, d_hat, f_delta, metric);
}

int ofdm_rx_decode_long(ofdm_ctx *c, const ofdm_fc32 *in_dev, int64_t n_samples, int64_t lag_lo, int64_t lag_hi, int64_t d_hat_known,
                        int32_t max_symbols, uint8_t *out_dev, int64_t out_cap, int32_t *out_len, int32_t *status, int64_t *offset,
                        double *f_delta, float *metric) {
    if (!c || n_samples <= 0 || lag_lo < 0 || max_symbols <= 0 || !in_dev || !out_dev || !out_len || !status) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    return decode_long_dev(c, reinterpret_cast<const float2 *>(in_dev), n_samples, lag_lo, lag_hi, d_hat_known, max_symbols, out_dev, out_cap,
                           out_len, status, offset, f_delta, metric);
}

int ofdm_rx_decode_long_host(ofdm_ctx *c, const ofdm_fc32 *in_host, int64_t n_samples, int32_t max_symbols, uint8_t *out_host,
                             int64_t out_cap, int32_t *out_len, int32_t *status, int64_t *offset, double *f_delta, float *metric) {
    if (!c || n_samples <= 0 || max_symbols <= 0 || !in_host || !out_host || !out_len || !status) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    HostPipe *hp;
    int rc = pipe_get(c, &hp);
    if (rc) return rc;
    // Once the upload has started, DMA reads the caller's buffer (and later writes out_host): every failure below waits for all of
    // it before the error goes back, as run_pipe does
    auto work = [&]() -> int {
        const size_t bytes = (size_t)n_samples * sizeof(ofdm_fc32);
        if ((rc = dev_grow(c, hp, hp->d_long, bytes + 64))) return rc;
        if ((rc = dev_grow(c, hp, hp->d_out[0], (size_t)std::max<int64_t>(out_cap, 4) + 64))) return rc;
        // upload: in place from pinned memory, else in 8 MB pieces through two pinned bounce slots (the copy of piece k + 1 overlaps
        // the DMA of piece k)
        if (host_pinned(in_host, bytes)) {
            HIP_TRY(c, hipMemcpyAsync(hp->d_long.p, in_host, bytes, hipMemcpyHostToDevice, hp->s_in));
        } else {
            const size_t piece = (size_t)8 << 20;
            for (int s = 0; s < 2; s++)
                if ((rc = pin_grow(c, hp, hp->h_in[s], std::min(piece, bytes)))) return rc;
            int k = 0;
            for (size_t off = 0; off < bytes; off += piece, k++) {
                const int s = k & 1;
                const size_t nb = std::min(piece, bytes - off);
                if (k >= 2) HIP_TRY(c, hipEventSynchronize(hp->in_done[s]));
                std::memcpy(hp->h_in[s].p, reinterpret_cast<const char *>(in_host) + off, nb);
                HIP_TRY(c, hipMemcpyAsync(static_cast<char *>(hp->d_long.p) + off, hp->h_in[s].p, nb, hipMemcpyHostToDevice, hp->s_in));
                HIP_TRY(c, hipEventRecord(hp->in_done[s], hp->s_in));
            }
        }
        HIP_TRY(c, hipEventRecord(hp->in_done[2], hp->s_in));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, hp->in_done[2], 0));
        rc = decode_long_dev(c, static_cast<const float2 *>(hp->d_long.p), n_samples, 0, 0, -1, max_symbols, static_cast<uint8_t *>(hp->d_out[0].p),
                             out_cap, out_len, status, offset, f_delta, metric);
        if (rc) return rc;
        if (*status == OFDM_FRAME_OK && *out_len > 0) {
            HIP_TRY(c, hipMemcpyAsync(out_host, hp->d_out[0].p, (size_t)*out_len, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        return OFDM_OK;
    };
    rc = work();
    if (rc) sync_all(c, hp);
    return rc;
}

} // extern "C"
