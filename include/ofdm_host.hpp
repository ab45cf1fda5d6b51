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

#include <complex>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
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

    // encode (src/transmitter.rs:11-58)
    std::vector<Complex64> encode(const std::vector<uint8_t> &data) {
        const int64_t n = ofdm_frame_samples(ctx_, (int64_t)data.size());
        DevBuf din(ctx_, data.size()), dout(ctx_, (size_t)n * sizeof(ofdm_fc32));
        check(ofdm_memcpy_h2d(ctx_, din.p, data.data(), data.size()), "h2d");
        check(ofdm_tx_encode_batch(ctx_, (const uint8_t *)din.p, 1, (int64_t)data.size(), nullptr, (int32_t)data.size(),
                                   (ofdm_fc32 *)dout.p, n), "ofdm_tx_encode_batch");
        std::vector<ofdm_fc32> host((size_t)n);
        check(ofdm_memcpy_d2h(ctx_, host.data(), dout.p, host.size() * sizeof(ofdm_fc32)), "d2h");
        return from_fc32(host);
    }
    // decode (src/receiver.rs:9-96).  Takes the samples by value like the reference (which consumes its Vec).
    std::vector<uint8_t> decode(std::vector<Complex64> samples) {
        const auto fc = to_fc32(samples);
        const int S = symbol_len();
        const int64_t n = (int64_t)fc.size();
        const int32_t max_sym = (int32_t)std::max<int64_t>((n + S - 1) / S - 10, 1);
        const int64_t ob = (int64_t)max_sym * ofdm_bytes_per_symbol(ctx_);
        DevBuf din(ctx_, fc.size() * sizeof(ofdm_fc32)), dout(ctx_, (size_t)ob), dmeta(ctx_, 2 * sizeof(int32_t));
        check(ofdm_memcpy_h2d(ctx_, din.p, fc.data(), fc.size() * sizeof(ofdm_fc32)), "h2d");
        int32_t *meta = (int32_t *)dmeta.p;
        check(ofdm_rx_decode_batch(ctx_, (const ofdm_fc32 *)din.p, 1, n, n, 0, max_sym, (uint8_t *)dout.p, ob, meta, meta + 1,
                                   nullptr, nullptr, nullptr), "ofdm_rx_decode_batch");
        int32_t m[2];
        check(ofdm_memcpy_d2h(ctx_, m, dmeta.p, sizeof(m)), "d2h");
        if (m[1] == OFDM_FRAME_SHORT) throw Error("Input not long enough, bailing early"); // src/receiver.rs:27-29
        if (m[1] != OFDM_FRAME_OK) throw Error("decode failed, frame status " + std::to_string(m[1]));
        std::vector<uint8_t> out((size_t)m[0]);
        if (!out.empty()) check(ofdm_memcpy_d2h(ctx_, out.data(), dout.p, out.size()), "d2h");
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

// free functions with the reference's optional-argument defaults (src/transmitter.rs:16-17, src/receiver.rs:16,83)
inline std::vector<Complex64> encode(const std::vector<uint8_t> &data, std::optional<bool> guard_bands = std::nullopt,
                                     std::optional<ModulationScheme> modulation = std::nullopt) {
    Context c(guard_bands.value_or(false), modulation.value_or(ModulationScheme::Bpsk));
    return c.encode(data);
}
inline std::vector<uint8_t> decode(std::vector<Complex64> samples, std::optional<bool> guard_bands = std::nullopt,
                                   std::optional<ModulationScheme> modulation = std::nullopt) {
    Context c(guard_bands.value_or(false), modulation.value_or(ModulationScheme::Bpsk), 64, OFDM_ECC_NONE, OFDM_CFO_ABS);
    return c.decode(std::move(samples));
}

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
