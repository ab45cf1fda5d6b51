// ofdm_host.hpp -- C++ host-side mirror of the reference's TX/RX function surface over the C ABI (ofdm_hip.h).
//
// The reference crate exposes free functions (src/lib.rs:8-21): encode, decode, modulate, demodulate, prefix_block,
// unprefix_block, ... on Vec<Complex64>.  This header keeps those names, argument meanings, defaults and error
// behaviour for a C++ host: std::vector<std::complex<double>> in and out, conversion to the fc32 wire format at the
// boundary (utils::sig_to_bytes / bytes_to_sig, src/utils.rs:228-254), device staging through ofdm_dev_alloc /
// ofdm_memcpy_*.  Header-only; link with -lofdm_hip.  Errors: decode returns the reference's
// "Input not long enough, bailing early" (src/receiver.rs:27-29) as a std::runtime_error, as anyhow::Result does.
#pragma once
#include "ofdm_hip.h"

#include <algorithm>
#include <complex>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

namespace ofdm {

using Complex64 = std::complex<double>;
enum class ModulationScheme { Bpsk = OFDM_MOD_BPSK, Qpsk = OFDM_MOD_QPSK, Qam16 = OFDM_MOD_QAM16, Qam64 = OFDM_MOD_QAM64, Qam256 = OFDM_MOD_QAM256 };

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

inline void check(int rc, const char *what) {
    if (rc != OFDM_OK) throw Error(std::string(what) + ": " + ofdm_strerror(rc));
}

// RAII context = one (thread, GPU) handle
// which preamble / training tables a context is built with (src/transmitter.rs:75-96)
enum class Pilots {
    Default, // documented SplitMix64 draws (ofdm_default_pilots)
    StdRng,  // the reference's own rand 0.8 StdRng draws, restated and unverified (ofdm_stdrng_pilots)
};
inline Pilots &default_pilot_choice() { static Pilots p = Pilots::Default; return p; } // used by the free encode / decode

class Context {
  public:
    explicit Context(bool guard_bands = false, ModulationScheme m = ModulationScheme::Bpsk, int n_fft = 64,
                     int ecc = OFDM_ECC_NONE, int cfo_mode = OFDM_CFO_SIGNED, int device = 0,
                     Pilots pilots = default_pilot_choice()) {
        ofdm_params p;
        check(ofdm_default_params(&p), "ofdm_default_params");
        p.n_fft = n_fft; p.cp_len = n_fft / 4; p.guard_bands = guard_bands; p.modulation = (int)m; p.ecc = ecc; p.cfo_mode = cfo_mode;
        if (pilots == Pilots::StdRng) {
            std::vector<double> pre(2 * (size_t)(n_fft + n_fft / 4)), trn(2 * (size_t)n_fft);
            check(ofdm_stdrng_pilots(n_fft, n_fft / 4, pre.data(), trn.data()), "ofdm_stdrng_pilots");
            check(ofdm_create(&p, pre.data(), trn.data(), device, nullptr, &ctx_), "ofdm_create");
        } else {
            check(ofdm_create(&p, nullptr, nullptr, device, nullptr, &ctx_), "ofdm_create");
        }
    }
    ~Context() { ofdm_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    ofdm_ctx *raw() const { return ctx_; }
    int symbol_len() const { return ofdm_symbol_len(ctx_); }

    struct DevBuf { // device allocation tied to the context
        ofdm_ctx *c; void *p = nullptr;
        DevBuf(ofdm_ctx *ctx, size_t bytes) : c(ctx) { check(ofdm_dev_alloc(c, bytes ? bytes : 1, &p), "ofdm_dev_alloc"); }
        ~DevBuf() { ofdm_dev_free(c, p); }
        DevBuf(const DevBuf &) = delete;
    };
    static std::vector<ofdm_fc32> to_fc32(const std::vector<Complex64> &x) { // sig_to_bytes (src/utils.rs:228-236)
        std::vector<ofdm_fc32> o(x.size());
        for (size_t i = 0; i < x.size(); ++i) o[i] = ofdm_fc32{(float)x[i].real(), (float)x[i].imag()};
        return o;
    }
    static std::vector<Complex64> from_fc32(const std::vector<ofdm_fc32> &x) { // bytes_to_sig (src/utils.rs:238-254)
        std::vector<Complex64> o(x.size());
        for (size_t i = 0; i < x.size(); ++i) o[i] = Complex64(x[i].re, x[i].im);
        return o;
    }

    // encode (src/transmitter.rs:11-58): host bytes in, host samples out (ofdm_tx_encode_host does the staging)
    std::vector<Complex64> encode(const std::vector<uint8_t> &data) {
        const int64_t n = ofdm_frame_samples(ctx_, (int64_t)data.size());
        std::vector<ofdm_fc32> host((size_t)n);
        const uint8_t none = 0;
        check(ofdm_tx_encode_host(ctx_, data.empty() ? &none : data.data(), 1, (int64_t)data.size(), nullptr, (int32_t)data.size(),
                                  host.data(), n, 0), "ofdm_tx_encode_host");
        return from_fc32(host);
    }
    // decode (src/receiver.rs:9-96) of ONE capture of any length -- the 2 M-sample buffers of examples/jetson_rx.rs:15-17,84-86
    // included: ofdm_rx_decode_long_host searches a long capture as a batch of overlapping slices.  Takes the samples by value
    // like the reference (which consumes its Vec).  max_symbols <= 0: every symbol up to the capture's end, as the reference.
    struct Decoded { std::vector<uint8_t> bytes; int32_t status = 0; int64_t offset = 0; double f_delta = 0.0; float metric = 0.f; };
    Decoded decode_capture(const ofdm_fc32 *fc, int64_t n, int32_t max_symbols = 0) {
        const int S = symbol_len();
        if (max_symbols <= 0) max_symbols = (int32_t)std::max<int64_t>((n + S - 1) / S - 10, 1);
        const int64_t ob = std::max<int64_t>((int64_t)max_symbols * ofdm_bytes_per_symbol(ctx_), 4);
        Decoded r;
        r.bytes.resize((size_t)ob);
        int32_t len = 0;
        check(ofdm_rx_decode_long_host(ctx_, fc, n, max_symbols, r.bytes.data(), ob, &len, &r.status, &r.offset, &r.f_delta, &r.metric),
              "ofdm_rx_decode_long_host");
        r.bytes.resize(r.status == OFDM_FRAME_OK ? (size_t)len : 0);
        return r;
    }
    std::vector<uint8_t> decode(std::vector<Complex64> samples, int32_t max_symbols = 0) {
        const auto fc = to_fc32(samples);
        Decoded r = decode_capture(fc.data(), (int64_t)fc.size(), max_symbols);
        if (r.status == OFDM_FRAME_SHORT) throw Error("Input not long enough, bailing early"); // src/receiver.rs:27-29
        if (r.status != OFDM_FRAME_OK) throw Error("decode failed, frame status " + std::to_string(r.status));
        return std::move(r.bytes);
    }
    // batches on host memory: H2D / kernels / D2H pipelined inside the library (ofdm_rx_decode_host, ofdm_tx_encode_host)
    struct BatchResult { std::vector<uint8_t> bytes; int64_t row = 0; std::vector<int32_t> len, status, offset; std::vector<double> f_delta; };
    BatchResult decode_batch(const ofdm_fc32 *frames, int64_t n_frames, int64_t frame_stride, int64_t frame_len, int32_t max_symbols,
                             int64_t n_lags = 0, int64_t chunk_frames = 0) {
        BatchResult r;
        r.row = std::max<int64_t>((int64_t)max_symbols * ofdm_bytes_per_symbol(ctx_) - 16, 4);
        r.bytes.resize((size_t)(n_frames * r.row));
        r.len.resize((size_t)n_frames); r.status.resize((size_t)n_frames); r.offset.resize((size_t)n_frames); r.f_delta.resize((size_t)n_frames);
        check(ofdm_rx_decode_host(ctx_, frames, n_frames, frame_stride, frame_len, n_lags, max_symbols, r.bytes.data(), r.row, r.len.data(),
                                  r.status.data(), r.offset.data(), r.f_delta.data(), nullptr, chunk_frames), "ofdm_rx_decode_host");
        return r;
    }
    std::vector<ofdm_fc32> encode_batch(const uint8_t *payload, int64_t n_frames, int32_t payload_bytes, int64_t chunk_frames = 0) {
        const int64_t frame = ofdm_frame_samples(ctx_, payload_bytes);
        std::vector<ofdm_fc32> out((size_t)(n_frames * frame));
        check(ofdm_tx_encode_host(ctx_, payload, n_frames, payload_bytes, nullptr, payload_bytes, out.data(), frame, chunk_frames),
              "ofdm_tx_encode_host");
        return out;
    }
    // modulate / demodulate (src/transmitter.rs:108-140, src/receiver.rs:147-190)
    std::vector<Complex64> modulate(const std::vector<uint8_t> &stream) {
        ofdm_params q; (void)q;
        const int bps = 8 * ofdm_bytes_per_symbol(ctx_) / ofdm_data_carriers(ctx_);
        const size_t n = (stream.size() * 8 + bps - 1) / bps;
        DevBuf din(ctx_, stream.size()), dout(ctx_, n * sizeof(ofdm_fc32));
        check(ofdm_memcpy_h2d(ctx_, din.p, stream.data(), stream.size()), "h2d");
        check(ofdm_qam_map_batch(ctx_, (const uint8_t *)din.p, (int64_t)stream.size(), (ofdm_fc32 *)dout.p), "ofdm_qam_map_batch");
        std::vector<ofdm_fc32> host(n);
        check(ofdm_memcpy_d2h(ctx_, host.data(), dout.p, n * sizeof(ofdm_fc32)), "d2h");
        return from_fc32(host);
    }
    std::vector<uint8_t> demodulate(const std::vector<Complex64> &stream) {
        const auto fc = to_fc32(stream);
        const int bps = 8 * ofdm_bytes_per_symbol(ctx_) / ofdm_data_carriers(ctx_);
        std::vector<uint8_t> out(fc.size() * bps / 8);
        DevBuf din(ctx_, fc.size() * sizeof(ofdm_fc32)), dout(ctx_, out.size());
        check(ofdm_memcpy_h2d(ctx_, din.p, fc.data(), fc.size() * sizeof(ofdm_fc32)), "h2d");
        check(ofdm_qam_demap_batch(ctx_, (const ofdm_fc32 *)din.p, (int64_t)fc.size(), (uint8_t *)dout.p, nullptr),
              "ofdm_qam_demap_batch"); // OFDM_ERR_INVALID unless size % 8 == 0 (assert at src/receiver.rs:153)
        if (!out.empty()) check(ofdm_memcpy_d2h(ctx_, out.data(), dout.p, out.size()), "d2h");
        return out;
    }

  private:
    ofdm_ctx *ctx_ = nullptr;
};

// One context per (thread, parameter set), created on first use and kept: the reference's free functions are called once per
// frame (examples/lab3a.rs:24,34, jetson_rx.rs:86), and ofdm_create -- table uploads, stream and workspace set-up -- must not be
// paid on every call.  The cache is thread_local: its contexts are destroyed (ofdm_destroy: HIP calls) when the THREAD exits.  A host
// that ends threads after the HIP runtime has been torn down (static destruction order, exit() from another thread) must call
// clear_cached_contexts() on that thread first.
namespace detail {
using ContextKey = std::tuple<bool, int, int, int, int, int, int>;
inline std::map<ContextKey, std::unique_ptr<Context>> &context_cache() {
    thread_local std::map<ContextKey, std::unique_ptr<Context>> cache;
    return cache;
}
} // namespace detail
inline void clear_cached_contexts() { detail::context_cache().clear(); } // this thread's contexts, now
inline Context &cached_context(bool guard_bands, ModulationScheme m, int n_fft = 64, int ecc = OFDM_ECC_NONE,
                               int cfo_mode = OFDM_CFO_SIGNED, int device = 0) {
    using Key = detail::ContextKey;
    auto &cache = detail::context_cache();
    const Key k{guard_bands, (int)m, n_fft, ecc, cfo_mode, device, (int)default_pilot_choice()};
    auto it = cache.find(k);
    if (it == cache.end()) it = cache.emplace(k, std::make_unique<Context>(guard_bands, m, n_fft, ecc, cfo_mode, device)).first;
    return *it->second;
}

// free functions with the reference's optional-argument defaults (src/transmitter.rs:16-17, src/receiver.rs:16,83)
inline std::vector<Complex64> encode(const std::vector<uint8_t> &data, std::optional<bool> guard_bands = std::nullopt,
                                     std::optional<ModulationScheme> modulation = std::nullopt) {
    return cached_context(guard_bands.value_or(false), modulation.value_or(ModulationScheme::Bpsk)).encode(data);
}
inline std::vector<uint8_t> decode(std::vector<Complex64> samples, std::optional<bool> guard_bands = std::nullopt,
                                   std::optional<ModulationScheme> modulation = std::nullopt) {
    return cached_context(guard_bands.value_or(false), modulation.value_or(ModulationScheme::Bpsk), 64, OFDM_ECC_NONE, OFDM_CFO_ABS)
        .decode(std::move(samples));
}

// Frame-index data parallelism inside ONE process ("one host thread + one stream per device"): a context per
// entry of `devices` (an ordinal may repeat: several contexts, each with its own stream, on one GPU), a std::thread per context
// for the duration of a call, frames [r F / R, (r + 1) F / R) to context r, results written straight into the caller's arrays --
// no exchange between the devices.  The Python twin is ofdm_amd/dist.py (one process per GPU).
class ShardedContext {
  public:
    explicit ShardedContext(std::vector<int> devices, bool guard_bands = false, ModulationScheme m = ModulationScheme::Bpsk, int n_fft = 64,
                            int ecc = OFDM_ECC_NONE, int cfo_mode = OFDM_CFO_SIGNED) {
        if (devices.empty()) throw Error("ShardedContext: no device");
        for (int d : devices) {
            ctx_.push_back(std::make_unique<Context>(guard_bands, m, n_fft, ecc, cfo_mode, d));
            check(ofdm_use_own_stream(ctx_.back()->raw()), "ofdm_use_own_stream");
        }
    }
    static std::vector<int> all_devices() { // one context per visible GPU
        int n = 0;
        check(ofdm_device_count(&n), "ofdm_device_count");
        std::vector<int> d((size_t)n);
        for (int i = 0; i < n; ++i) d[(size_t)i] = i;
        return d;
    }
    size_t size() const { return ctx_.size(); }
    Context &operator[](size_t r) { return *ctx_[r]; }
    static std::pair<int64_t, int64_t> shard_range(int64_t n, int64_t r, int64_t world) { return {n * r / world, n * (r + 1) / world}; }

    Context::BatchResult decode_batch(const ofdm_fc32 *frames, int64_t n_frames, int64_t frame_stride, int64_t frame_len,
                                      int32_t max_symbols, int64_t n_lags = 0, int64_t chunk_frames = 0) {
        Context::BatchResult r;
        r.row = std::max<int64_t>((int64_t)max_symbols * ofdm_bytes_per_symbol(ctx_[0]->raw()) - 16, 4);
        r.bytes.resize((size_t)(n_frames * r.row));
        r.len.resize((size_t)n_frames); r.status.resize((size_t)n_frames); r.offset.resize((size_t)n_frames); r.f_delta.resize((size_t)n_frames);
        run([&](size_t i, int64_t lo, int64_t hi) {
            return ofdm_rx_decode_host(ctx_[i]->raw(), frames + lo * frame_stride, hi - lo, frame_stride, frame_len, n_lags, max_symbols,
                                       r.bytes.data() + lo * r.row, r.row, r.len.data() + lo, r.status.data() + lo, r.offset.data() + lo,
                                       r.f_delta.data() + lo, nullptr, chunk_frames);
        }, n_frames, "ofdm_rx_decode_host");
        return r;
    }
    std::vector<ofdm_fc32> encode_batch(const uint8_t *payload, int64_t n_frames, int32_t payload_bytes, int64_t chunk_frames = 0) {
        const int64_t frame = ofdm_frame_samples(ctx_[0]->raw(), payload_bytes);
        std::vector<ofdm_fc32> out((size_t)(n_frames * frame));
        run([&](size_t i, int64_t lo, int64_t hi) {
            return ofdm_tx_encode_host(ctx_[i]->raw(), payload + lo * payload_bytes, hi - lo, payload_bytes, nullptr, payload_bytes,
                                       out.data() + lo * frame, frame, chunk_frames);
        }, n_frames, "ofdm_tx_encode_host");
        return out;
    }

  private:
    template <class F> void run(F &&shard, int64_t n_frames, const char *what) {
        const size_t R = ctx_.size();
        std::vector<int> rc(R, OFDM_OK);
        std::vector<std::thread> th;
        th.reserve(R); // no reallocation while threads run
        struct Joiner { // whatever happens below (std::thread's constructor can throw), started shards are joined before the stack unwinds:
            std::vector<std::thread> &t; // a joinable std::thread destroyed means std::terminate with GPU work in flight
            ~Joiner() { for (auto &x : t) if (x.joinable()) x.join(); }
        } joiner{th};
        for (size_t i = 0; i < R; ++i) {
            const auto [lo, hi] = shard_range(n_frames, (int64_t)i, (int64_t)R);
            if (hi > lo) th.emplace_back([&, i, lo = lo, hi = hi] { rc[i] = shard(i, lo, hi); }); // every entry point selects its context's device itself
        }
        for (auto &t : th) t.join();
        for (int v : rc) check(v, what);
    }
    std::vector<std::unique_ptr<Context>> ctx_;
};

// outer Reed-Solomon(255,223) framing of the demos (src/utils.rs:97-180), host side
inline std::vector<uint8_t> create_transmission_bytes(const std::vector<uint8_t> &data) { // utils.rs:97-136
    std::vector<uint8_t> out((size_t)ofdm_rs255_encoded_len((int64_t)data.size()));
    check(ofdm_rs255_encode(data.data(), (int64_t)data.size(), out.data()), "ofdm_rs255_encode");
    return out;
}
inline std::optional<std::vector<uint8_t>> decipher_transmission_bytes(const std::vector<uint8_t> &code) { // utils.rs:150-180
    std::vector<uint8_t> out((size_t)ofdm_rs255_decoded_len((int64_t)code.size()));
    const int rc = ofdm_rs255_decode(code.data(), (int64_t)code.size(), out.data(), nullptr);
    if (rc == OFDM_ERR_UNCORRECTABLE) return std::nullopt; // the reference returns None
    check(rc, "ofdm_rs255_decode");
    return out;
}

// utils::Analysis (src/utils.rs:38-69)
struct Analysis {
    uint32_t num_errs = 0, num_block_errs = 0;
    double err_rate = 0.0;
    Analysis(const std::vector<uint8_t> &l, const std::vector<uint8_t> &r) {
        if (l.size() != r.size()) throw Error("Analysis: length mismatch");
        for (size_t i = 0; i < l.size(); ++i)
            if (l[i] != r[i]) { num_errs += (uint32_t)__builtin_popcount((unsigned)(l[i] ^ r[i])); num_block_errs++; }
        err_rate = l.empty() ? 0.0 : (double)num_errs / ((double)l.size() * 8.0);
    }
};

} // namespace ofdm
