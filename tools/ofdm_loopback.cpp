// ofdm_loopback -- the reference's loop-back examples (examples/lab3a.rs:11-46, lab3b.rs:12-38) on the GPU path:
//   text -> encode -> channel(FIR + optional CFO + noise, seeded) -> decode -> Analysis + print.
// The channel here is bench plumbing (src/channel.rs:26-74 re-stated with a seeded SplitMix64 instead of thread_rng);
// encode / decode are the library.  Build: g++ -std=c++17 -I include tools/ofdm_loopback.cpp -L ofdm_amd -lofdm_hip
#include "ofdm_host.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>

using namespace ofdm;

static const double CHANNEL[64] = {0, 0, 0, 0, 0, 0, 0, -0.0, -0.1912, 0.9316, 0.2821, -0.1990, 0.1630, -0.1017, 0.0544, -0.0261, 0.0090, 0.0, -0.0034};

static uint64_t sm64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static double u01(uint64_t &s) { return (double)(sm64(s) >> 11) * (1.0 / 9007199254740992.0); }

static std::vector<Complex64> channel(const std::vector<Complex64> &tx, double snr_db, bool timing_error, uint64_t seed, double *fd_used) {
    std::vector<Complex64> y(tx.size() + 63, Complex64(0, 0));
    for (size_t i = 0; i < tx.size(); ++i)
        for (int k = 8; k <= 18; ++k) y[i + k] += tx[i] * CHANNEL[k]; // convolve(CHANNEL): taps 8..18 are the non-zero ones
    uint64_t st = seed;
    double fd = 0.0;
    if (timing_error) {
        fd = M_PI * (u01(st) / 80.0); // src/channel.rs:54
        for (size_t i = 0; i < y.size(); ++i) y[i] *= std::exp(Complex64(0, fd * (double)(i + 1)));
    }
    if (fd_used) *fd_used = fd;
    Complex64 mean(0, 0), var(0, 0);
    for (auto &v : y) mean += v;
    mean /= (double)y.size();
    for (auto &v : y) var += (mean - v) * (mean - v); // complex pseudo-variance (src/signals/mod.rs:239-249)
    var /= (double)y.size();
    const Complex64 scale = std::sqrt(0.5 * var / std::pow(10.0, snr_db / 10.0));
    for (auto &v : y) { const double re = u01(st) * 2 - 1, im = u01(st) * 2 - 1; v += scale * Complex64(re, im); }
    return y;
}

// fc32 wire format (src/utils.rs:228-254): native-endian f32 pairs (re, im), what UHD's tx_samples_from_file /
// rx_samples_to_file --type float read and write (data/transmit.sh, data/receive.sh).
static bool write_fc32(const char *path, const std::vector<Complex64> &x) {
    FILE *fh = std::fopen(path, "wb");
    if (!fh) return false;
    for (auto &v : x) { const float p[2] = {(float)v.real(), (float)v.imag()}; std::fwrite(p, sizeof(float), 2, fh); }
    std::fclose(fh);
    return true;
}
static bool read_fc32(const char *path, std::vector<Complex64> &x, long start, long stop) {
    FILE *fh = std::fopen(path, "rb");
    if (!fh) return false;
    float p[2];
    long i = 0;
    while (std::fread(p, sizeof(float), 2, fh) == 2) { // chunks_exact(8): a trailing partial sample is dropped
        if (i >= start && (stop < 0 || i < stop)) x.emplace_back(p[0], p[1]);
        ++i;
    }
    std::fclose(fh);
    return true;
}

int main(int argc, char **argv) {
    const char *corpus = "I met a traveller from an antique land, Who said: Two vast and trunkless legs of stone Stand in the desert. ";
    size_t num_bytes = 400;
    bool guard_bands = false, timing_error = false, ecc = false; // ecc: lab3c's ecc_enabled (outer RS(255,223), utils.rs:97-180)
    ModulationScheme modulation = ModulationScheme::Qpsk;
    const char *tx_file = nullptr, *rx_file = nullptr; // examples/lab3c.rs: --transmit f / --receive f [--start a --stop b]
    long start = 0, stop = -1;
    int devices = 0; long frames = 0; // --devices N --frames F: F frames through N contexts side by side (ShardedContext)
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--timing-error")) timing_error = true;          // lab3b
        else if (!std::strcmp(argv[i], "--guard")) guard_bands = true;
        else if (!std::strcmp(argv[i], "--ecc")) ecc = true;
        else if (!std::strcmp(argv[i], "--bpsk")) modulation = ModulationScheme::Bpsk;
        else if (!std::strcmp(argv[i], "--qam64")) modulation = ModulationScheme::Qam64;
        else if (!std::strcmp(argv[i], "--bytes") && i + 1 < argc) num_bytes = (size_t)std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "--transmit") && i + 1 < argc) tx_file = argv[++i];
        else if (!std::strcmp(argv[i], "--receive") && i + 1 < argc) rx_file = argv[++i];
        else if (!std::strcmp(argv[i], "--start") && i + 1 < argc) start = std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "--stop") && i + 1 < argc) stop = std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) devices = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--frames") && i + 1 < argc) frames = std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "--pilots") && i + 1 < argc)                 // "stdrng": the reference's own tables (restated)
            default_pilot_choice() = !std::strcmp(argv[++i], "stdrng") ? Pilots::StdRng : Pilots::Default;
    }
    try {
        std::vector<uint8_t> source(num_bytes);
        for (size_t i = 0; i < num_bytes; ++i) source[i] = (uint8_t)corpus[i % std::strlen(corpus)]; // create_transmission_text
        const std::vector<uint8_t> text = source;
        if (ecc) source = create_transmission_bytes(source);                       // create_transmission_text(num_bytes, true)
        auto finish = [&](std::vector<uint8_t> got) -> std::vector<uint8_t> {      // decipher_transmission_text
            if (!ecc) return got;
            auto plain = decipher_transmission_bytes(got);
            if (!plain) { std::printf("outer code: uncorrectable block\n"); return {}; }
            plain->resize(std::min(plain->size(), text.size()));
            return *plain;
        };
        if (devices > 0) {
            // Frame-index split inside one process: F frames with different payloads, encoded and decoded by N contexts
            // (context r on GPU r mod the visible count, each with its own stream and host thread); the result must equal the
            // single-context result byte for byte.
            if (frames <= 0) frames = 64;
            int ngpu = 0;
            check(ofdm_device_count(&ngpu), "ofdm_device_count");
            std::vector<int> devs;
            for (int r = 0; r < devices; ++r) devs.push_back(r % std::max(ngpu, 1));
            std::vector<uint8_t> pay((size_t)frames * num_bytes);
            for (long f = 0; f < frames; ++f)
                for (size_t i = 0; i < num_bytes; ++i) pay[(size_t)f * num_bytes + i] = (uint8_t)corpus[(i + 7 * (size_t)f) % std::strlen(corpus)];
            ShardedContext sh(devs, guard_bands, modulation);
            Context one(guard_bands, modulation);
            const auto tx_sh = sh.encode_batch(pay.data(), frames, (int32_t)num_bytes, 5);
            const auto tx_one = one.encode_batch(pay.data(), frames, (int32_t)num_bytes);
            if (tx_sh.size() != tx_one.size() || std::memcmp(tx_sh.data(), tx_one.data(), tx_one.size() * sizeof(ofdm_fc32))) { std::printf("sharded encode differs\n"); return 5; }
            const int64_t flen = (int64_t)(tx_one.size() / (size_t)frames), stride = (flen + 63 + 64 + 1) & ~(int64_t)1;
            std::vector<ofdm_fc32> cap((size_t)(frames * stride));
            for (long f = 0; f < frames; ++f) { // channel per frame, a different delay and CFO draw for each
                std::vector<ofdm_fc32> one_tx(tx_one.begin() + f * flen, tx_one.begin() + (f + 1) * flen);
                auto rx = channel(Context::from_fc32(one_tx), 30.0, timing_error, 2021 + (uint64_t)f, nullptr);
                const auto fc = Context::to_fc32(rx);
                const size_t delay = (size_t)(f * 13 % 60);
                for (size_t i = 0; i < fc.size() && delay + i < (size_t)stride; ++i) cap[(size_t)(f * stride) + delay + i] = fc[i];
            }
            const int32_t D = (int32_t)ofdm_data_symbols(one.raw(), (int64_t)num_bytes);
            const auto r_sh = sh.decode_batch(cap.data(), frames, stride, stride, D, 0, 3);
            const auto r_one = one.decode_batch(cap.data(), frames, stride, stride, D);
            if (r_sh.len != r_one.len || r_sh.status != r_one.status || r_sh.offset != r_one.offset || r_sh.bytes != r_one.bytes) { std::printf("sharded decode differs\n"); return 5; }
            uint32_t errs = 0; long ok = 0;
            for (long f = 0; f < frames; ++f) {
                if (r_sh.status[(size_t)f] != OFDM_FRAME_OK || r_sh.len[(size_t)f] != (int32_t)num_bytes) continue;
                std::vector<uint8_t> got(r_sh.bytes.begin() + f * r_sh.row, r_sh.bytes.begin() + f * r_sh.row + (long)num_bytes);
                std::vector<uint8_t> want(pay.begin() + f * (long)num_bytes, pay.begin() + (f + 1) * (long)num_bytes);
                errs += Analysis(want, got).num_errs; ++ok;
            }
            std::printf("sharded over %d contexts on %d GPU(s): %ld frames, identical to the single-context result; %ld decoded, num_errs: %u\n",
                        devices, std::max(ngpu, 1), frames, ok, errs);
            return ok == frames && errs == 0 ? 0 : 1;
        }
        if (rx_file) { // decode a stored capture (or a slice of it)
            std::vector<Complex64> cap;
            if (!read_fc32(rx_file, cap, start, stop)) { std::printf("cannot read %s\n", rx_file); return 4; }
            auto received = finish(decode(std::move(cap), guard_bands, modulation));
            std::printf("received %zu bytes\n%.*s\n", received.size(), (int)std::min<size_t>(received.size(), 100), (const char *)received.data());
            if (received.size() != text.size()) return 2;
            return Analysis(text, received).num_errs == 0 ? 0 : 1;
        }
        auto tx = encode(source, guard_bands, modulation);                         // ofdm::encode!
        if (tx_file) { // write the frame for tx_samples_from_file
            if (!write_fc32(tx_file, tx)) { std::printf("cannot write %s\n", tx_file); return 4; }
            std::printf("wrote %zu samples (%zu bytes) to %s\n", tx.size(), tx.size() * 8, tx_file);
            return 0;
        }
        double fd = 0;
        auto rx = channel(tx, 30.0, timing_error, 2021, &fd);                      // ofdm::channel!(snr: 30.0[, timing_error])
        auto received = finish(decode(std::move(rx), guard_bands, modulation));    // ofdm::decode!
        if (received.size() != text.size()) { std::printf("length mismatch: %zu vs %zu\n", received.size(), text.size()); return 2; }
        Analysis a(text, received);
        std::printf("Analysis { num_errs: %u, num_block_errs: %u, err_rate: %g }  samples: %zu  f_delta: %g\n", a.num_errs,
                    a.num_block_errs, a.err_rate, tx.size(), fd);
        std::printf("%.*s\n", (int)std::min<size_t>(received.size(), 100), (const char *)received.data());
        return a.num_errs == 0 ? 0 : 1;
    } catch (const std::exception &e) {
        std::printf("error: %s\n", e.what());
        return 3;
    }
}
