// ofdm_abi.hip -- the extern "C" boundary of libofdm_hip.so (include/ofdm_hip.h).
// Context, parameter validation, host-side constant tables (f64 -> f32), workspace and the orchestration of
// the TX / RX pipelines.  No exceptions cross the boundary; every HIP failure is mapped to OFDM_ERR_HIP.
#include "ofdm_ctx.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace ofdm {
int sc_tile_lags();
}
using namespace ofdm;

namespace {

constexpr double kPi = 3.14159265358979323846;

struct cd { double re, im; };
static inline cd cd_mul(cd a, cd b) { return cd{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// host f64 radix-2 FFT for the constant tables (product code: independent of oracle/)
static void host_fft(std::vector<cd> &x, bool inverse) {
    const size_t n = x.size();
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(x[i], x[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t k = 0; k < len / 2; k++) {
            double ang = (inverse ? 2.0 : -2.0) * kPi * (double)k / (double)len;
            cd w{std::cos(ang), std::sin(ang)};
            for (size_t i = k; i < n; i += len) {
                cd u = x[i], v = cd_mul(x[i + len / 2], w);
                x[i] = cd{u.re + v.re, u.im + v.im};
                x[i + len / 2] = cd{u.re - v.re, u.im - v.im};
            }
        }
    }
    if (inverse)
        for (auto &v : x) { v.re /= (double)n; v.im /= (double)n; }
}

static uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static double uniform_pm1(uint64_t &s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0; }

static bool valid_nfft(int n) { return n >= 64 && n <= 4096 && (n & (n - 1)) == 0; }

} // namespace

static SymParams base_params(ofdm_ctx *c) {
    SymParams p;
    p.tune = &c->tune;
    p.trace = &c->trace;
    p.tw = c->d_tw;
    p.inv_training = c->d_inv_trn;
    p.sym_len = c->S();
    p.bps = c->prm.modulation;
    p.guard = c->prm.guard_bands;
    return p;
}

extern "C" {

int ofdm_abi_version(void) { return OFDM_HIP_ABI_VERSION; }

const char *ofdm_strerror(int status) {
    switch (status) {
    case OFDM_OK: return "ok";
    case OFDM_ERR_INVALID: return "invalid argument";
    case OFDM_ERR_UNSUPPORTED: return "unsupported configuration";
    case OFDM_ERR_NO_DEVICE: return "no usable HIP device";
    case OFDM_ERR_HIP: return "HIP runtime error";
    case OFDM_ERR_NOMEM: return "out of device memory";
    case OFDM_ERR_UNCORRECTABLE: return "uncorrectable Reed-Solomon block";
    default: return "unknown status";
    }
}

int ofdm_device_count(int *count) {
    if (!count) return OFDM_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return OFDM_ERR_NO_DEVICE; }
    *count = n;
    return OFDM_OK;
}

int ofdm_default_params(ofdm_params *p) {
    if (!p) return OFDM_ERR_INVALID;
    std::memset(p, 0, sizeof(*p));
    p->n_fft = 64;           // src/transmitter.rs:33,52
    p->cp_len = 16;
    p->modulation = OFDM_MOD_BPSK; // transmitter.rs:17
    p->guard_bands = 0;            // transmitter.rs:16
    p->ecc = OFDM_ECC_NONE;
    p->sync_window_reps = 3;
    p->sync_backoff = 4;
    p->cfo_mode = OFDM_CFO_SIGNED;
    p->sync_threshold = 0.5f;
    return OFDM_OK;
}

int ofdm_default_pilots(int32_t n_fft, int32_t cp_len, double *preamble, double *training) {
    if (!valid_nfft(n_fft) || cp_len != n_fft / 4) return OFDM_ERR_INVALID;
    if (preamble) {
        uint64_t s = 100; // transmitter.rs:76
        for (int i = 0; i < n_fft + cp_len; i++) {
            double re = uniform_pm1(s), im = uniform_pm1(s);
            preamble[2 * i] = re * 0.25;
            preamble[2 * i + 1] = im * 0.25;
        }
    }
    if (training) {
        uint64_t s = 50; // transmitter.rs:89
        for (int i = 0; i < n_fft; i++) {
            double re = uniform_pm1(s), im = uniform_pm1(s);
            training[2 * i] = re * 1.0;
            training[2 * i + 1] = im * 1.0;
        }
    }
    return OFDM_OK;
}

// ---- rand 0.8 `StdRng` (= ChaCha12Rng of rand_chacha 0.3) restated from the crates' published algorithm, for the
// reference's pilot tables (src/transmitter.rs:75-96).  The ChaCha core is pinned by the published RFC 7539 / ChaCha12
// test vectors (tests/test_abi_cpu.py); seeding (rand_core 0.6 seed_from_u64: PCG32 expansion) and the f64
// gen_range(-1.0..1.0) mapping (rand 0.8.3 UniformFloat::sample_single: 52 mantissa bits -> [1,2) - 1, * scale + low)
// cannot be checked against a running `rand` here: the tables are "unverified" (DESIGN.md section 2).
namespace {
struct StdRng {
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t buf[16];
    int idx = 16;
    static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    static void qr(uint32_t *s, int a, int b, int c, int d) {
        s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 16);
        s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 12);
        s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 8);
        s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 7);
    }
    static void block(const uint32_t key[8], const uint32_t w12_15[4], int rounds, uint32_t out[16]) {
        uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        for (int i = 0; i < 8; i++) st[4 + i] = key[i];
        for (int i = 0; i < 4; i++) st[12 + i] = w12_15[i];
        uint32_t s[16];
        for (int i = 0; i < 16; i++) s[i] = st[i];
        for (int r = 0; r < rounds / 2; r++) {
            qr(s, 0, 4, 8, 12); qr(s, 1, 5, 9, 13); qr(s, 2, 6, 10, 14); qr(s, 3, 7, 11, 15);
            qr(s, 0, 5, 10, 15); qr(s, 1, 6, 11, 12); qr(s, 2, 7, 8, 13); qr(s, 3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) out[i] = s[i] + st[i];
    }
    explicit StdRng(uint64_t state) { // SeedableRng::seed_from_u64: PCG32 output per 4 seed bytes (little endian)
        for (int i = 0; i < 8; i++) {
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            const uint32_t rot = (uint32_t)(state >> 59);
            key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
    }
    uint32_t next_u32() {
        if (idx == 16) { // 64-bit block counter in words 12-13, stream id 0 in words 14-15
            const uint32_t w[4] = {(uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
            block(key, w, 12, buf);
            ++counter;
            idx = 0;
        }
        return buf[idx++];
    }
    uint64_t next_u64() { const uint64_t lo = next_u32(); return lo | ((uint64_t)next_u32() << 32); } // BlockRng: low word first
    double range_pm1() { // gen_range(-1.0..1.0)
        const uint64_t bits = (next_u64() >> 12) | 0x3ff0000000000000ull;
        double v;
        std::memcpy(&v, &bits, 8);
        return (v - 1.0) * 2.0 + -1.0;
    }
};
} // namespace

int ofdm_stdrng_pilots(int32_t n_fft, int32_t cp_len, double *preamble, double *training) {
    if (!valid_nfft(n_fft) || cp_len != n_fft / 4) return OFDM_ERR_INVALID;
    if (preamble) {
        StdRng r(100); // transmitter.rs:76
        for (int i = 0; i < n_fft + cp_len; i++) {
            const double re = r.range_pm1(), im = r.range_pm1(); // re drawn before im (transmitter.rs:80)
            preamble[2 * i] = re * 0.25;
            preamble[2 * i + 1] = im * 0.25;
        }
    }
    if (training) {
        StdRng r(50); // transmitter.rs:89
        for (int i = 0; i < n_fft; i++) {
            const double re = r.range_pm1(), im = r.range_pm1();
            training[2 * i] = re;
            training[2 * i + 1] = im;
        }
    }
    return OFDM_OK;
}

int ofdm_chacha_block(const uint32_t *key8, const uint32_t *words12_15, int32_t rounds, uint32_t *out16) {
    if (!key8 || !words12_15 || !out16 || rounds <= 0 || (rounds & 1)) return OFDM_ERR_INVALID;
    StdRng::block(key8, words12_15, rounds, out16);
    return OFDM_OK;
}

int ofdm_destroy(ofdm_ctx *c) {
    if (!c) return OFDM_OK;
    int prev_dev = -1;
    const bool have_prev = hipGetDevice(&prev_dev) == hipSuccess;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    ofdm_host_pipe_destroy(c);
    for (auto &w : c->ws) if (w.ptr) hipFree(w.ptr);
    if (c->d_tw) hipFree(c->d_tw);
    if (c->d_inv_trn) hipFree(c->d_inv_trn);
    if (c->d_header) hipFree(c->d_header);
    if (c->d_stats) hipFree(c->d_stats);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
    if (have_prev) hipSetDevice(prev_dev);
    return OFDM_OK;
}

int ofdm_create(const ofdm_params *p, const double *preamble, const double *training, int device, void *stream,
                ofdm_ctx **out) {
    if (!p || !out) return OFDM_ERR_INVALID;
    *out = nullptr;
    if (!valid_nfft(p->n_fft) || p->cp_len != p->n_fft / 4) return OFDM_ERR_INVALID;
    switch (p->modulation) {
    case OFDM_MOD_BPSK: case OFDM_MOD_QPSK: case OFDM_MOD_QAM16: case OFDM_MOD_QAM64: case OFDM_MOD_QAM256: break;
    default: return OFDM_ERR_INVALID;
    }
    if (p->guard_bands != 0 && p->guard_bands != 1) return OFDM_ERR_INVALID;
    if (p->ecc != OFDM_ECC_NONE && p->ecc != OFDM_ECC_HAMMING74) return OFDM_ERR_INVALID;
    if (p->sync_window_reps < 1 || p->sync_window_reps > 3) return OFDM_ERR_INVALID;
    if (p->sync_backoff < 0 || p->sync_backoff > p->cp_len) return OFDM_ERR_INVALID;
    if (p->cfo_mode < OFDM_CFO_OFF || p->cfo_mode > OFDM_CFO_ABS) return OFDM_ERR_INVALID;
    if (!(p->sync_threshold > 0.f && p->sync_threshold <= 1.f)) return OFDM_ERR_INVALID;
    if (p->sync_mode != OFDM_SYNC_SCHMIDL_COX && p->sync_mode != OFDM_SYNC_REFERENCE) return OFDM_ERR_INVALID;
    if (p->rx_path != OFDM_RX_AUTO && p->rx_path != OFDM_RX_STAGED) return OFDM_ERR_INVALID; // (2 was the one-pass kernel of rounds 2-4: removed)
    for (int r : p->reserved) if (r != 0) return OFDM_ERR_INVALID;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return OFDM_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return OFDM_ERR_NO_DEVICE;
    DeviceGuard dev_guard(device); // the caller's current device is restored on every return path
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return OFDM_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return OFDM_ERR_NO_DEVICE; // gfx950 code objects only

    ofdm_ctx *c = new (std::nothrow) ofdm_ctx();
    if (!c) return OFDM_ERR_NOMEM;
    c->prm = *p;
    c->device = device;
    c->trace.reset();
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int rc = OFDM_OK;
    do {
        c->stream = (hipStream_t)stream; // NULL = the device's default (null) stream

        if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { rc = OFDM_ERR_HIP; break; }

        const int N = p->n_fft, CP = p->cp_len, S = N + CP;
        std::vector<double> pre(2 * S), trn(2 * N);
        ofdm_default_pilots(N, CP, pre.data(), trn.data());
        if (preamble) std::memcpy(pre.data(), preamble, sizeof(double) * 2 * S);
        if (training) std::memcpy(trn.data(), training, sizeof(double) * 2 * N);

        std::vector<float2> tw(N), inv(N), hdr(10 * S);
        for (int m = 0; m < N; m++) {
            double a = -2.0 * kPi * (double)m / (double)N;
            tw[m] = make_float2((float)std::cos(a), (float)std::sin(a));
            double re = trn[2 * m], im = trn[2 * m + 1], ns = re * re + im * im;
            inv[m] = make_float2((float)(re / ns), (float)(-im / ns));
        }
        // header = [locking_signal(S)] [preamble x4] [prefix_block(training) x5]  (src/transmitter.rs:22-34)
        std::vector<cd> lock(S);
        for (int i = 0; i < S; i++) lock[i] = cd{0.5 * ((double)i / (2.0 * (double)S) + 0.5), 0.0}; // transmitter.rs:63-66
        const int mid = (S + 1) / 2;                                                               // fft_shift, mod.rs:65
        for (int i = 0; i < S; i++) { cd v = lock[(i + mid) % S]; hdr[i] = make_float2((float)v.re, (float)v.im); }
        for (int r = 0; r < 4; r++)
            for (int i = 0; i < S; i++) hdr[(1 + r) * S + i] = make_float2((float)pre[2 * i], (float)pre[2 * i + 1]);
        std::vector<cd> t(N);
        for (int i = 0; i < N; i++) t[i] = cd{trn[2 * i], trn[2 * i + 1]};
        host_fft(t, true);
        for (int r = 0; r < 5; r++) {
            float2 *blk = hdr.data() + (5 + r) * S;
            for (int i = 0; i < CP; i++) blk[i] = make_float2((float)t[N - CP + i].re, (float)t[N - CP + i].im);
            for (int i = 0; i < N; i++) blk[CP + i] = make_float2((float)t[i].re, (float)t[i].im);
        }
        float hmax = 0.f;
        for (auto &v : hdr) { hmax = std::fmax(hmax, v.x); hmax = std::fmax(hmax, v.y); }
        c->header_max = hmax;

        if (hipMalloc(&c->d_tw, sizeof(float2) * N) != hipSuccess || hipMalloc(&c->d_inv_trn, sizeof(float2) * N) != hipSuccess ||
            hipMalloc(&c->d_header, sizeof(float2) * 10 * S) != hipSuccess ||
            hipMalloc(&c->d_stats, 2 * sizeof(int32_t)) != hipSuccess) { rc = OFDM_ERR_NOMEM; break; }
        if (hipMemcpy(c->d_tw, tw.data(), sizeof(float2) * N, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(c->d_inv_trn, inv.data(), sizeof(float2) * N, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(c->d_header, hdr.data(), sizeof(float2) * 10 * S, hipMemcpyHostToDevice) != hipSuccess) { rc = OFDM_ERR_HIP; break; }
    } while (0);
    if (rc != OFDM_OK) { ofdm_destroy(c); return rc; }
    *out = c;
    return OFDM_OK;
}

int ofdm_set_stream(ofdm_ctx *c, void *stream) {
    if (!c) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (c->own_stream && c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); c->own_stream = false; }
    c->stream = (hipStream_t)stream;
    return OFDM_OK;
}
int ofdm_use_own_stream(ofdm_ctx *c) {
    if (!c) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (c->own_stream) return OFDM_OK;
    hipStream_t s = nullptr;
    HIP_TRY(c, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (c->stream) hipStreamSynchronize(c->stream);
    c->stream = s;
    c->own_stream = true;
    return OFDM_OK;
}
int ofdm_synchronize(ofdm_ctx *c) {
    if (!c) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return OFDM_OK;
}
int ofdm_last_hip_error(const ofdm_ctx *c) { return c ? c->last_hip : 0; }

namespace {
struct TuneKey { const char *name; int Tuning::*field; bool profile_only; };
const TuneKey kTuneKeys[] = { // the laboratory keys: ofdm_hip_tuning.h (private); the public ones are handled by name below
#define OFDM_TUNE_KEY(name, field, prof) {name, &Tuning::field, prof},
#include "ofdm_hip_tuning.h"
#undef OFDM_TUNE_KEY
};
} // namespace

int ofdm_set_tuning(ofdm_ctx *c, const char *key, int64_t value) {
    if (!c || !key) return OFDM_ERR_INVALID;
    if (std::strcmp(key, "grid_cap") == 0) { if (value < 0) return OFDM_ERR_INVALID; c->tune.grid_cap = value; return OFDM_OK; }
    for (const TuneKey &k : kTuneKeys)
        if (std::strcmp(key, k.name) == 0) {
            if (k.profile_only && !kProfile) return OFDM_ERR_UNSUPPORTED; // the ablation branches are not in this build
            if (value < 0 || value > 0x7fffffff) return OFDM_ERR_INVALID;
            // keys that size LDS tiles or pick template instantiations take only the values their kernels were built for
            if (k.field == &Tuning::sc_first_lags && value > 1280) return OFDM_ERR_INVALID;       // the first launch's 128-chunk tile holds 1280 samples
            if (k.field == &Tuning::tx_waves && (value < 1 || value > 32)) return OFDM_ERR_INVALID;
            if (k.field == &Tuning::demod64_burst && value != 16 && value != 8 && value != 4 && value != 1) return OFDM_ERR_INVALID;
            if (k.field == &Tuning::sc_wg_per_cu && value > 16) return OFDM_ERR_INVALID;
            c->tune.*(k.field) = (int)value;
            return OFDM_OK;
        }
    return OFDM_ERR_INVALID;
}
int ofdm_get_tuning(const ofdm_ctx *c, const char *key, int64_t *value) {
    if (!c || !key || !value) return OFDM_ERR_INVALID;
    if (std::strcmp(key, "grid_cap") == 0) { *value = c->tune.grid_cap; return OFDM_OK; }
    if (std::strcmp(key, "profile_build") == 0) { *value = kProfile ? 1 : 0; return OFDM_OK; }
    if (std::strcmp(key, "stat_sc_slow_frames") == 0 || std::strcmp(key, "stat_sc_redo_frames") == 0) {
        // counters of the LAST Schmidl-Cox search of this context (synchronises its stream); -1 when that search kept no such list
        const bool slow = key[8] == 's';
        *value = -1;
        if (!c->sc_stats.dev || !(slow ? c->sc_stats.has_slow : c->sc_stats.has_redo)) return OFDM_OK;
        const int32_t *src = c->sc_stats.dev + (slow ? 0 : 1);
        DeviceGuard dev_guard(c->device);
        int32_t v = 0;
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(&v, src, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return OFDM_ERR_HIP;
        *value = v;
        return OFDM_OK;
    }
    for (const TuneKey &k : kTuneKeys)
        if (std::strcmp(key, k.name) == 0) { *value = c->tune.*(k.field); return OFDM_OK; }
    return OFDM_ERR_INVALID;
}
int ofdm_last_dispatch(const ofdm_ctx *c, char *buf, size_t n) {
    if (!c || !buf || !n) return OFDM_ERR_INVALID;
    const size_t len = (size_t)c->trace.len < n - 1 ? (size_t)c->trace.len : n - 1;
    std::memcpy(buf, c->trace.buf, len);
    buf[len] = 0;
    return (int)c->trace.len; // the full length, like snprintf
}

int ofdm_dev_alloc(ofdm_ctx *c, size_t bytes, void **dev) {
    if (!c || !dev) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    hipError_t e = hipMalloc(dev, bytes ? bytes : 1);
    if (e != hipSuccess) { c->last_hip = (int)e; *dev = nullptr; return OFDM_ERR_NOMEM; }
    return OFDM_OK;
}
int ofdm_dev_free(ofdm_ctx *c, void *dev) {
    if (!c) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (dev) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(dev)); }
    return OFDM_OK;
}
int ofdm_memcpy_h2d(ofdm_ctx *c, void *dev, const void *host, size_t bytes) {
    if (!c || (bytes && (!dev || !host))) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    HIP_TRY(c, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    return OFDM_OK;
}
int ofdm_memcpy_d2h(ofdm_ctx *c, void *host, const void *dev, size_t bytes) {
    if (!c || (bytes && (!dev || !host))) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    HIP_TRY(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return OFDM_OK;
}
int ofdm_memset(ofdm_ctx *c, void *dev, int value, size_t bytes) {
    if (!c || (bytes && !dev)) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    HIP_TRY(c, hipMemsetAsync(dev, value, bytes, c->stream));
    return OFDM_OK;
}

int ofdm_symbol_len(const ofdm_ctx *c) { return c ? c->S() : OFDM_ERR_INVALID; }
int ofdm_data_carriers(const ofdm_ctx *c) { return c ? c->carriers() : OFDM_ERR_INVALID; }
int ofdm_bytes_per_symbol(const ofdm_ctx *c) { return c ? c->bytes_per_symbol() : OFDM_ERR_INVALID; }
int64_t ofdm_coded_len(const ofdm_ctx *c, int64_t payload_bytes) {
    if (!c || payload_bytes < 0) return OFDM_ERR_INVALID;
    return c->prm.ecc == OFDM_ECC_HAMMING74 ? ((payload_bytes + 3) / 4) * 7 : payload_bytes;
}
int64_t ofdm_data_symbols(const ofdm_ctx *c, int64_t payload_bytes) {
    if (!c || payload_bytes < 0) return OFDM_ERR_INVALID;
    int64_t nsym = ((16 + ofdm_coded_len(c, payload_bytes)) * 8 + c->prm.modulation - 1) / c->prm.modulation;
    return (nsym + c->carriers() - 1) / c->carriers();
}
int64_t ofdm_frame_samples(const ofdm_ctx *c, int64_t payload_bytes) {
    if (!c || payload_bytes < 0) return OFDM_ERR_INVALID;
    return (10 + ofdm_data_symbols(c, payload_bytes)) * (int64_t)c->S();
}

// ------------------------------------------------------------------ stage level
int ofdm_fft_batch(ofdm_ctx *c, const ofdm_fc32 *in, ofdm_fc32 *out, int64_t n_vec, int inverse) {
    if (!c || n_vec < 0 || (n_vec && (!in || !out))) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    SymParams p = base_params(c);
    const int N = c->prm.n_fft;
    p.in = reinterpret_cast<const float2 *>(in); p.out = reinterpret_cast<float2 *>(out);
    p.n_frames = n_vec; p.frame_stride = N; p.frame_len = N; p.syms_per_frame = 1; p.in_sym_stride = N; p.in_skip = 0;
    HIP_TRY(c, run_fft(N, p, inverse != 0, c->stream, c->num_cu));
    return OFDM_OK;
}
int ofdm_ifft_cp_batch(ofdm_ctx *c, const ofdm_fc32 *freq, ofdm_fc32 *out, int64_t n_sym) {
    if (!c || n_sym < 0 || (n_sym && (!freq || !out))) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    SymParams p = base_params(c);
    const int N = c->prm.n_fft;
    p.in = reinterpret_cast<const float2 *>(freq); p.out = reinterpret_cast<float2 *>(out);
    p.n_frames = n_sym; p.frame_stride = N; p.frame_len = N; p.syms_per_frame = 1; p.in_sym_stride = N; p.in_skip = 0;
    HIP_TRY(c, run_ifft_cp(N, p, c->stream, c->num_cu));
    return OFDM_OK;
}
int ofdm_tx_symbols_batch(ofdm_ctx *c, const uint8_t *bytes, int64_t n_bytes, ofdm_fc32 *out, int64_t n_sym) {
    if (!c || n_bytes < 0 || n_sym < 0 || (n_sym && !out) || (n_bytes && !bytes)) return OFDM_ERR_INVALID;
    const int bps_bytes = c->bytes_per_symbol();
    if (n_sym * (int64_t)bps_bytes < n_bytes) return OFDM_ERR_INVALID; // every byte must land in a symbol
    if (!n_sym) return OFDM_OK;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    SymParams p = base_params(c);
    p.n_frames = n_sym; p.syms_per_frame = 1;
    p.payload = bytes; p.payload_stride = bps_bytes; p.payload_len = nullptr; p.payload_bytes = bps_bytes;
    p.tx_raw_total = n_bytes;
    p.out = reinterpret_cast<float2 *>(out); p.out_stride_s = c->S(); p.frame_max = nullptr;
    if (c->prm.n_fft == 4096) { // 64 x 64 two-stage kernel (kernels_fast.hip)
        hipError_t e = c->tune.no_demod4096 ? hipErrorNotSupported : run_tx4096(p, c->stream, c->num_cu); // A/B switch shared with the RX side
        if (e == hipSuccess) return OFDM_OK;
        if (e != hipErrorNotSupported) { c->last_hip = (int)e; return OFDM_ERR_HIP; }
    }
    if (c->prm.n_fft < 4096) { // R x 64 two-stage kernels (kernels_mid.hip; N = 64 is the one-row case)
        hipError_t e = c->tune.no_mid_kernels ? hipErrorNotSupported : run_tx_mid(c->prm.n_fft, p, c->stream, c->num_cu);
        if (e == hipSuccess) return OFDM_OK;
        if (e != hipErrorNotSupported) { c->last_hip = (int)e; return OFDM_ERR_HIP; }
    }
    HIP_TRY(c, run_tx_symbols(c->prm.n_fft, p, c->stream, c->num_cu));
    return OFDM_OK;
}
int ofdm_unprefix_batch(ofdm_ctx *c, const ofdm_fc32 *in, ofdm_fc32 *out, int64_t n_sym) {
    if (!c || n_sym < 0 || (n_sym && (!in || !out))) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    SymParams p = base_params(c);
    const int N = c->prm.n_fft, S = c->S();
    p.in = reinterpret_cast<const float2 *>(in); p.out = reinterpret_cast<float2 *>(out);
    p.n_frames = n_sym; p.frame_stride = S; p.frame_len = S; p.syms_per_frame = 1; p.in_sym_stride = S; p.in_skip = c->prm.cp_len;
    HIP_TRY(c, run_fft(N, p, false, c->stream, c->num_cu));
    return OFDM_OK;
}
int ofdm_qam_map_batch(ofdm_ctx *c, const uint8_t *bytes, int64_t n_bytes, ofdm_fc32 *out) {
    if (!c || n_bytes < 0 || (n_bytes && (!bytes || !out))) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    HIP_TRY(c, run_qam_map(bytes, n_bytes, c->prm.modulation, reinterpret_cast<float2 *>(out), c->stream));
    return OFDM_OK;
}
int ofdm_qam_demap_batch(ofdm_ctx *c, const ofdm_fc32 *sym, int64_t n_sym, uint8_t *bytes, uint8_t *idx) {
    if (!c || n_sym < 0 || (n_sym && !sym)) return OFDM_ERR_INVALID;
    if (n_sym % 8 != 0) return OFDM_ERR_INVALID; // assert_eq!(remainder.len(), 0), src/receiver.rs:153
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    HIP_TRY(c, run_qam_demap(reinterpret_cast<const float2 *>(sym), n_sym, c->prm.modulation, bytes, idx, c->stream));
    return OFDM_OK;
}
int ofdm_encode_block_batch(ofdm_ctx *c, const ofdm_fc32 *data, ofdm_fc32 *bins, int64_t n_sym) {
    if (!c || n_sym < 0 || (n_sym && (!data || !bins))) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    HIP_TRY(c, run_encode_block(reinterpret_cast<const float2 *>(data), reinterpret_cast<float2 *>(bins), n_sym,
                                c->prm.n_fft, c->prm.guard_bands, c->stream));
    return OFDM_OK;
}
int ofdm_normalize_batch(ofdm_ctx *c, ofdm_fc32 *x, int64_t n_frames, int64_t frame_stride, int64_t frame_len) {
    if (!c || n_frames < 0 || frame_len < 0 || frame_stride < frame_len || (n_frames && !x)) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    if (!n_frames) return OFDM_OK;
    void *mx;
    int rc = ws_get(c, 0, sizeof(unsigned) * (size_t)n_frames, &mx);
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(mx, 0, sizeof(unsigned) * (size_t)n_frames, c->stream));
    HIP_TRY(c, run_frame_max(reinterpret_cast<float2 *>(x), n_frames, frame_stride, frame_len, (unsigned *)mx, c->stream));
    HIP_TRY(c, run_frame_scale(reinterpret_cast<float2 *>(x), n_frames, frame_stride, frame_len, (unsigned *)mx, c->stream));
    return OFDM_OK;
}
int ofdm_hamming74_encode(ofdm_ctx *c, const uint8_t *in, int64_t n_bytes, uint8_t *out) {
    if (!c || n_bytes < 0 || (n_bytes && (!in || !out))) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    HIP_TRY(c, run_ham_encode(in, 1, 0, nullptr, n_bytes, out, 0, nullptr, c->stream));
    return OFDM_OK;
}
int ofdm_hamming74_decode(ofdm_ctx *c, const uint8_t *in, int64_t n_bytes, uint8_t *out, uint32_t *corrected) {
    if (!c || n_bytes < 0 || (n_bytes >= 7 && (!in || !out))) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    HIP_TRY(c, run_ham_decode(in, n_bytes, out, corrected, c->stream));
    return OFDM_OK;
}

// Schmidl-Cox parameters of a batch; false when no lag fits the capture (nothing can synchronise)
static bool sc_make_params(ofdm_ctx *c, const float2 *in, int64_t n_frames, int64_t frame_stride, int64_t frame_len,
                           int64_t n_lags, int32_t *d_hat, double *f_delta, float *metric, ScParams &p) {
    const int L = c->S(), W = c->prm.sync_window_reps * L;
    const int64_t valid = frame_len - W - L + 1;
    if (valid <= 0) return false;
    if (n_lags <= 0 || n_lags > valid) n_lags = valid;
    p.in = in; p.n_frames = n_frames; p.frame_stride = frame_stride; p.frame_len = frame_len; p.n_lags = n_lags;
    p.L = L; p.W = W; p.threshold = (double)c->prm.sync_threshold;
    p.d_hat = d_hat; p.f_delta = f_delta; p.metric = metric;
    p.tiles_per_frame = 1; p.mode = 0;
    p.tune = &c->tune; p.trace = &c->trace;
    c->sc_stats = ScStats();
    c->sc_stats.dev = c->d_stats;
    p.stats = &c->sc_stats;
    return true;
}

static int sc_run_impl(ofdm_ctx *c, const float2 *in, int64_t n_frames, int64_t frame_stride, int64_t frame_len,
                       int64_t n_lags, int32_t *d_hat, double *f_delta, float *metric, bool tail_mapped) {
    ScParams p;
    p.tail_mapped = tail_mapped;
    if (!sc_make_params(c, in, n_frames, frame_stride, frame_len, n_lags, d_hat, f_delta, metric, p)) { // no lag fits
        HIP_TRY(c, hipMemsetAsync(d_hat, 0xFF, sizeof(int32_t) * (size_t)n_frames, c->stream));
        if (f_delta) HIP_TRY(c, hipMemsetAsync(f_delta, 0, sizeof(double) * (size_t)n_frames, c->stream));
        if (metric) HIP_TRY(c, hipMemsetAsync(metric, 0, sizeof(float) * (size_t)n_frames, c->stream));
        return OFDM_OK;
    }
    n_lags = p.n_lags;
    // N = 64 (L = 80): k_sc80, every lag exactly in one streaming pass, whatever the slot length; its A/B predecessor (and the path
    // of unaligned batches): one-tile frames through the coarse-then-fine f32 filter + exact f64 decisions (k_sc_cf)
    if (sc_fast_ok(p) || (!c->tune.no_sc80 && sc80_wanted(p))) {
        void *wsp;
        int rc = ws_get(c, 6, sc_fast_workspace_bytes(n_frames, p.n_lags), &wsp);
        if (rc) return rc;
        HIP_TRY(c, run_sc_fast(p, wsp, c->num_cu, c->stream));
        return OFDM_OK;
    }
    // L = 160 .. 1280 (N = 128 .. 1024): ONE streaming pass per frame -- exact sums at every 10th lag from running prefixes, lag-by-lag
    // evaluation only where a bound allows a crossing / a new maximum -- that stops once the peak window is closed (kernels_scstream.hip)
    if (sc_stream_ok(p) && !c->tune.no_sc_stream) {
        HIP_TRY(c, run_sc_stream(p, c->num_cu, c->stream));
        return OFDM_OK;
    }
    // long periods (N >= 128): one streaming pass for chunk sums, then an exact search only where the chunk bounds allow a
    // crossing / a new maximum (kernels_scbig.hip); LDS footprint independent of L, so N = 4096 works too
    const bool one_small_tile = n_lags <= sc_tile_lags() && sc_lds_bytes(p) <= 64 * 1024; // bounded search, window in LDS: k_sc_tile reads less
    if (sc_big_ok(p) && !one_small_tile && !c->tune.no_sc_big) {
        void *wsp;
        int rc = ws_get(c, 6, sc_big_workspace_bytes(p), &wsp);
        if (rc) return rc;
        HIP_TRY(c, run_sc_big(p, wsp, c->num_cu, c->stream));
        return OFDM_OK;
    }
    if (sc_lds_bytes(p) > 160 * 1024) return OFDM_ERR_UNSUPPORTED; // window does not fit one CU's LDS
    const int CH = sc_tile_lags();
    const int64_t tiles = (n_lags + CH - 1) / CH;
    if (tiles > 0x7fffffff) return OFDM_ERR_INVALID;
    if (tiles == 1) {
        HIP_TRY(c, run_sc(p, c->stream));
        return OFDM_OK;
    }
    // long captures: first threshold crossing per tile -> per frame -> peak search from there (one tile)
    void *cross, *d1;
    int rc = ws_get(c, 6, sizeof(long long) * (size_t)(n_frames * tiles), &cross);
    if (rc) return rc;
    rc = ws_get(c, 7, sizeof(int32_t) * (size_t)n_frames, &d1);
    if (rc) return rc;
    p.tiles_per_frame = (int)tiles; p.mode = 1; p.cross = (long long *)cross;
    HIP_TRY(c, run_sc(p, c->stream));
    HIP_TRY(c, run_sc_min_cross((const long long *)cross, (int)tiles, n_frames, (int32_t *)d1, c->stream));
    p.tiles_per_frame = 1; p.mode = 2; p.lag_base = (const int32_t *)d1;
    HIP_TRY(c, run_sc(p, c->stream));
    return OFDM_OK;
}

int ofdm_abi_sc_run(ofdm_ctx *c, const float2 *in, int64_t n_frames, int64_t frame_stride, int64_t frame_len,
                    int64_t n_lags, int32_t *d_hat, double *f_delta, float *metric) {
    // Tight rows of odd length (stride == length): k_sc80 is given an even slot length, i.e. reads one sample past an odd slot -- the next
    // row's first one, except behind the last row.  That row goes through the general paths on its own, the others through k_sc80.
    if (n_frames > 1 && (frame_len & 1) && frame_stride == frame_len && !c->tune.no_sc80) {
        ScParams head;
        head.tail_mapped = true;
        if (sc_make_params(c, in, n_frames - 1, frame_stride, frame_len, n_lags, d_hat, f_delta, metric, head) && sc80_wanted(head)) {
            int rc = sc_run_impl(c, in, n_frames - 1, frame_stride, frame_len, n_lags, d_hat, f_delta, metric, true);
            if (rc) return rc;
            const int64_t l = n_frames - 1;
            return sc_run_impl(c, in + l * frame_stride, 1, frame_len, frame_len, n_lags, d_hat + l, f_delta ? f_delta + l : nullptr,
                               metric ? metric + l : nullptr, false);
        }
    }
    return sc_run_impl(c, in, n_frames, frame_stride, frame_len, n_lags, d_hat, f_delta, metric, false);
}

int ofdm_sc_correlate_batch(ofdm_ctx *c, const ofdm_fc32 *in, int64_t n_frames, int64_t frame_stride,
                            int64_t frame_len, int64_t n_lags, int32_t *d_hat, double *f_delta, float *metric) {
    if (!c || n_frames < 0 || frame_len <= 0 || frame_stride < 0 || (n_frames && (!in || !d_hat))) return OFDM_ERR_INVALID;
    if (n_frames > 1 && frame_stride <= 0) return OFDM_ERR_INVALID;
    if (!n_frames) return OFDM_OK;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    return ofdm_abi_sc_run(c, reinterpret_cast<const float2 *>(in), n_frames, frame_stride, frame_len, n_lags, d_hat, f_delta, metric);
}
int ofdm_xcorr_batch(ofdm_ctx *c, const ofdm_fc32 *a, int64_t n_frames, int64_t a_stride, int64_t a_len, const ofdm_fc32 *b,
                     int32_t nb, int32_t *idx_max, float *peak, ofdm_fc32 *out, int64_t out_stride) {
    if (!c || n_frames < 0 || a_len <= 0 || nb <= 0 || nb > 8192 || nb > a_len || a_len > 0x3fffffff) return OFDM_ERR_INVALID;
    if (n_frames && (!a || !b || !idx_max)) return OFDM_ERR_INVALID;
    if (n_frames > 1 && a_stride <= 0) return OFDM_ERR_INVALID;
    if (out && out_stride < 2 * a_len - 1) return OFDM_ERR_INVALID;
    if (!n_frames) return OFDM_OK;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    void *w;
    int rc = ws_get(c, 6, xcorr_workspace_bytes(n_frames, a_len, nb), &w);
    if (rc) return rc;
    HIP_TRY(c, run_xcorr(reinterpret_cast<const float2 *>(a), n_frames, a_stride, a_len, reinterpret_cast<const float2 *>(b), nb, w,
                         idx_max, peak, reinterpret_cast<float2 *>(out), out_stride, c->num_cu, c->stream));
    return OFDM_OK;
}

int ofdm_frequency_correction_batch(ofdm_ctx *c, const ofdm_fc32 *in, int64_t n_pairs, int64_t stride,
                                    int64_t right_offset, double *f_delta) {
    if (!c || n_pairs < 0 || (n_pairs && (!in || !f_delta))) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    HIP_TRY(c, run_freq_correction(reinterpret_cast<const float2 *>(in), n_pairs, stride, right_offset, c->S(), f_delta, c->stream));
    return OFDM_OK;
}
int ofdm_cfo_rotate_batch(ofdm_ctx *c, ofdm_fc32 *x, int64_t n_frames, int64_t frame_stride, int64_t frame_len,
                          const double *f_delta, const int32_t *first_index) {
    if (!c || n_frames < 0 || frame_len < 0 || (n_frames && (!x || !f_delta))) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    HIP_TRY(c, run_cfo_rotate(reinterpret_cast<float2 *>(x), n_frames, frame_stride, frame_len, f_delta, first_index, c->stream));
    return OFDM_OK;
}
int ofdm_estimate_channel_batch(ofdm_ctx *c, const ofdm_fc32 *in, int64_t n_frames, int64_t frame_stride,
                                int64_t frame_len, const int32_t *offset, const double *f_delta, ofdm_fc32 *hk) {
    if (!c || n_frames < 0 || frame_len <= 0 || (n_frames && (!in || !hk))) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    SymParams p = base_params(c);
    p.in = reinterpret_cast<const float2 *>(in); p.out = reinterpret_cast<float2 *>(hk);
    p.n_frames = n_frames; p.frame_stride = frame_stride; p.frame_len = frame_len;
    p.offset = offset; p.f_delta = f_delta;
    HIP_TRY(c, run_chest(c->prm.n_fft, p, c->stream, c->num_cu));
    return OFDM_OK;
}

static int demod_run(ofdm_ctx *c, const float2 *in, int64_t n_frames, int64_t frame_stride, int64_t frame_len,
                     int first_symbol, int syms_per_frame, const int32_t *offset, const double *f_delta,
                     const int32_t *nsym_frame, const float2 *hk, int64_t hk_stride, uint8_t *out, int64_t out_stride,
                     float2 *soft) {
    SymParams p = base_params(c);
    p.in = in; p.n_frames = n_frames; p.frame_stride = frame_stride; p.frame_len = frame_len;
    p.first_symbol = first_symbol; p.syms_per_frame = syms_per_frame; p.in_sym_stride = c->S(); p.in_skip = c->prm.cp_len;
    p.offset = offset; p.f_delta = f_delta; p.nsym_frame = nsym_frame;
    p.hk = hk; p.hk_stride = hk_stride; p.out_bytes = out; p.out_stride = out_stride; p.soft = soft;
    if (c->prm.n_fft == 64) { // regular, aligned streams take the wave-centric fast path (kernels_fast.hip)
        hipError_t e = c->tune.no_fast64 ? hipErrorNotSupported : run_demod64_fast(p, c->stream, c->num_cu);
        if (e == hipSuccess) return OFDM_OK;
        if (e != hipErrorNotSupported) { c->last_hip = (int)e; return OFDM_ERR_HIP; }
    }
    if (c->prm.n_fft == 4096) { // 64 x 64 two-stage kernel for regular streams (kernels_fast.hip)
        hipError_t e = c->tune.no_demod4096 ? hipErrorNotSupported : run_demod4096(p, c->stream, c->num_cu); // A/B switch
        if (e == hipSuccess) return OFDM_OK;
        if (e != hipErrorNotSupported) { c->last_hip = (int)e; return OFDM_ERR_HIP; }
    }
    if (c->prm.n_fft > 64 && c->prm.n_fft < 4096) { // R x 64 two-stage kernels for regular streams (kernels_mid.hip)
        hipError_t e = c->tune.no_mid_kernels ? hipErrorNotSupported : run_demod_mid(c->prm.n_fft, p, c->stream, c->num_cu); // A/B switch
        if (e == hipSuccess) return OFDM_OK;
        if (e != hipErrorNotSupported) { c->last_hip = (int)e; return OFDM_ERR_HIP; }
    }
    HIP_TRY(c, run_demod(c->prm.n_fft, p, c->stream, c->num_cu));
    return OFDM_OK;
}

int ofdm_rx_demod_batch(ofdm_ctx *c, const ofdm_fc32 *in, int64_t n_frames, int64_t frame_stride, int64_t frame_len,
                        int32_t first_symbol, int32_t syms_per_frame, const int32_t *offset, const double *f_delta,
                        const ofdm_fc32 *hk, int64_t hk_stride, uint8_t *out, int64_t out_stride, ofdm_fc32 *soft) {
    if (!c || n_frames < 0 || syms_per_frame < 0 || first_symbol < 0 || frame_len <= 0) return OFDM_ERR_INVALID;
    if (n_frames && syms_per_frame && (!in || !out)) return OFDM_ERR_INVALID;
    if (out_stride < (int64_t)syms_per_frame * c->bytes_per_symbol()) return OFDM_ERR_INVALID;
    if (hk && hk_stride != 0 && hk_stride != c->prm.n_fft) return OFDM_ERR_INVALID;
    if (!n_frames || !syms_per_frame) return OFDM_OK;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    return demod_run(c, reinterpret_cast<const float2 *>(in), n_frames, frame_stride, frame_len, first_symbol,
                     syms_per_frame, offset, f_delta, nullptr, reinterpret_cast<const float2 *>(hk), hk_stride, out,
                     out_stride, reinterpret_cast<float2 *>(soft));
}

// ------------------------------------------------------------------ pipelines
int ofdm_tx_encode_batch(ofdm_ctx *c, const uint8_t *payload, int64_t n_frames, int64_t payload_stride,
                         const int32_t *payload_len, int32_t payload_bytes, ofdm_fc32 *out, int64_t out_stride) {
    if (!c || n_frames < 0 || payload_bytes < 0 || payload_stride < 0) return OFDM_ERR_INVALID;
    if (n_frames && (!out || (payload_bytes && !payload))) return OFDM_ERR_INVALID;
    if (n_frames > 1 && payload_stride < payload_bytes) return OFDM_ERR_INVALID; // rows are prefetched for payload_bytes (include/ofdm_hip.h)
    const int64_t frame = ofdm_frame_samples(c, payload_bytes);
    if (out_stride < frame) return OFDM_ERR_INVALID;
    if (!n_frames) return OFDM_OK;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    const int S = c->S();
    const uint8_t *src = payload; int64_t src_stride = payload_stride; const int32_t *src_len = payload_len;
    int32_t src_bytes = payload_bytes;
    if (c->prm.ecc == OFDM_ECC_HAMMING74) {
        const int64_t coded = ofdm_coded_len(c, payload_bytes);
        void *cw, *cl;
        int rc = ws_get(c, 1, (size_t)(coded ? coded : 1) * (size_t)n_frames, &cw);
        if (rc) return rc;
        rc = ws_get(c, 2, sizeof(int32_t) * (size_t)n_frames, &cl);
        if (rc) return rc;
        HIP_TRY(c, run_ham_encode(payload, n_frames, payload_stride, payload_len, payload_bytes, (uint8_t *)cw, coded,
                                  (int32_t *)cl, c->stream));
        src = (const uint8_t *)cw; src_stride = coded; src_len = payload_len ? (const int32_t *)cl : nullptr;
        src_bytes = (int32_t)coded;
    }
    SymParams p = base_params(c);
    p.n_frames = n_frames; p.syms_per_frame = (int)ofdm_data_symbols(c, payload_bytes);
    p.payload = src; p.payload_stride = src_stride; p.payload_len = src_len; p.payload_bytes = src_bytes;
    p.out = reinterpret_cast<float2 *>(out); p.out_stride_s = out_stride;
    if (c->prm.n_fft == 64) { // one workgroup per frame, frame built in LDS, single pass over HBM
        hipError_t fe = c->tune.no_txframe64 ? hipErrorNotSupported : run_txframe64(p, c->d_header, c->header_max, c->stream, c->num_cu);
        if (fe == hipSuccess) return OFDM_OK;
        if (fe != hipErrorNotSupported) { c->last_hip = (int)fe; return OFDM_ERR_HIP; }
    }
    if (c->prm.n_fft < 4096) { // R x 64 two-stage kernel, frames built twice: one pass over HBM (kernels_mid.hip)
        hipError_t fe = c->tune.no_mid_kernels ? hipErrorNotSupported : run_txframe_mid(c->prm.n_fft, p, c->d_header, c->header_max, c->stream, c->num_cu);
        if (fe == hipSuccess) return OFDM_OK;
        if (fe != hipErrorNotSupported) { c->last_hip = (int)fe; return OFDM_ERR_HIP; }
    }
    if (c->prm.n_fft == 4096) { // 64 x 64 two-stage kernel, frames built twice: one pass over HBM (kernels_fast.hip)
        hipError_t fe = c->tune.no_demod4096 ? hipErrorNotSupported : run_txframe4096(p, c->d_header, c->header_max, c->stream, c->num_cu);
        if (fe == hipSuccess) return OFDM_OK;
        if (fe != hipErrorNotSupported) { c->last_hip = (int)fe; return OFDM_ERR_HIP; }
    }
    void *mx;
    int rc = ws_get(c, 0, sizeof(unsigned) * (size_t)n_frames, &mx);
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(mx, 0, sizeof(unsigned) * (size_t)n_frames, c->stream));
    p.frame_max = (unsigned *)mx;
    HIP_TRY(c, run_tx_symbols(c->prm.n_fft, p, c->stream, c->num_cu));
    c->trace.add("k_tx_finish");
    HIP_TRY(c, run_tx_finish(reinterpret_cast<float2 *>(out), n_frames, out_stride, 10 * S, frame, c->d_header,
                             c->header_max, (const unsigned *)mx, c->stream));
    return OFDM_OK;
}

namespace { struct KnownSync { int32_t d_hat; double f_delta; float metric; }; }
static int rx_decode_impl(ofdm_ctx *c, const ofdm_fc32 *in, int64_t n_frames, int64_t frame_stride, int64_t frame_len,
                          int64_t n_lags, int32_t max_symbols, uint8_t *out, int64_t out_stride, int32_t *out_len,
                          int32_t *status, int32_t *offset, double *f_delta, float *metric, const KnownSync *known);

int ofdm_rx_decode_batch(ofdm_ctx *c, const ofdm_fc32 *in, int64_t n_frames, int64_t frame_stride, int64_t frame_len,
                         int64_t n_lags, int32_t max_symbols, uint8_t *out, int64_t out_stride, int32_t *out_len,
                         int32_t *status, int32_t *offset, double *f_delta, float *metric) {
    return rx_decode_impl(c, in, n_frames, frame_stride, frame_len, n_lags, max_symbols, out, out_stride, out_len, status, offset, f_delta,
                          metric, nullptr);
}
int ofdm_abi_rx_decode_known(ofdm_ctx *c, const ofdm_fc32 *in, int64_t frame_len, int32_t d_hat, double f_delta, float metric,
                             int32_t max_symbols, uint8_t *out, int64_t out_stride, int32_t *out_len, int32_t *status, int32_t *offset,
                             double *f_delta_out, float *metric_out) {
    const KnownSync k{d_hat, f_delta, metric};
    return rx_decode_impl(c, in, 1, frame_len, frame_len, 0, max_symbols, out, out_stride, out_len, status, offset, f_delta_out, metric_out, &k);
}

static int rx_decode_impl(ofdm_ctx *c, const ofdm_fc32 *in, int64_t n_frames, int64_t frame_stride, int64_t frame_len,
                          int64_t n_lags, int32_t max_symbols, uint8_t *out, int64_t out_stride, int32_t *out_len,
                          int32_t *status, int32_t *offset, double *f_delta, float *metric, const KnownSync *known) {
    if (!c || n_frames < 0 || frame_len <= 0 || max_symbols <= 0) return OFDM_ERR_INVALID;
    if (n_frames && (!in || !out || !out_len || !status)) return OFDM_ERR_INVALID;
    if (n_frames > 1 && frame_stride <= 0) return OFDM_ERR_INVALID;
    const int bps_bytes = c->bytes_per_symbol();
    const int64_t raw_bytes = (int64_t)max_symbols * bps_bytes;
    const int64_t raw_stride = (raw_bytes + 3) & ~(int64_t)3;   // rows of the raw-byte workspace start on dwords (6-byte BPSK symbols: odd counts)
    // rows must hold what k_rx_finish can write: the whole body without an outer code, floor(body / 7) * 4 bytes after
    // Hamming(7,4) decoding (include/ofdm_hip.h)
    const int64_t body_max = raw_bytes > 16 ? raw_bytes - 16 : 0;
    if (out_stride < (c->prm.ecc == OFDM_ECC_NONE ? body_max : (body_max / 7) * 4)) return OFDM_ERR_INVALID;
    if (!n_frames) return OFDM_OK;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    const int N = c->prm.n_fft;
    void *w_dhat, *w_fd, *w_off, *w_nsym, *w_hk = nullptr, *w_raw = nullptr;
    int rc;
    if ((rc = ws_get(c, 0, sizeof(int32_t) * (size_t)n_frames, &w_dhat))) return rc;
    if ((rc = ws_get(c, 1, sizeof(double) * (size_t)n_frames, &w_fd))) return rc;
    if ((rc = ws_get(c, 2, sizeof(int32_t) * (size_t)n_frames, &w_off))) return rc;
    if ((rc = ws_get(c, 3, sizeof(int32_t) * (size_t)n_frames, &w_nsym))) return rc;
    int32_t *offs = offset ? offset : (int32_t *)w_off;
    double *fd = f_delta ? f_delta : (double *)w_fd;
    const float2 *x = reinterpret_cast<const float2 *>(in);
    if ((rc = ws_get(c, 5, (size_t)raw_stride * (size_t)n_frames, &w_raw))) return rc;
    if (known) {
        // 1k. the caller brings the timing of the one capture (ofdm_rx_decode_long with lag_lo > 0)
        if (n_frames != 1 || c->prm.sync_mode != OFDM_SYNC_SCHMIDL_COX) return OFDM_ERR_INVALID;
        c->trace.add("k_set_sync+k_rx_prepare");
        HIP_TRY(c, run_set_sync((int32_t *)w_dhat, fd, metric, known->d_hat, known->f_delta, known->metric, c->stream));
        HIP_TRY(c, run_rx_prepare(n_frames, (const int32_t *)w_dhat, fd, frame_len, c->S(), c->prm.sync_backoff,
                                  c->prm.cfo_mode, max_symbols, bps_bytes, status, offs, (int32_t *)w_nsym, c->stream));
    } else if (c->prm.sync_mode == OFDM_SYNC_REFERENCE) {
        // 1r. the reference's own detector (src/receiver.rs:20-25): cross-correlation with the locking signal, offset =
        //     idx_max - N (= lag - 1), then frequency_correction on chunks 3 and 4 (receiver.rs:39; always |.|)
        if (frame_len > 0x3fffffff) return OFDM_ERR_UNSUPPORTED;
        void *w_x;
        if ((rc = ws_get(c, 6, xcorr_workspace_bytes(n_frames, frame_len, c->S()), &w_x))) return rc;
        c->trace.add("k_xcorr+k_rx_prepare_ref");
        HIP_TRY(c, run_xcorr(x, n_frames, frame_stride, frame_len, c->d_header, c->S(), w_x, (int32_t *)w_dhat, metric, nullptr, 0,
                             c->num_cu, c->stream));
        // The REPORTED offset may be negative (idx_max - N, quirk Q1; -N for an all-zero capture) and goes to the caller's array
        // only; the kernels below read the workspace copy, which is 0 for every frame whose status is not OFDM_FRAME_OK, so that
        // no fetch can start in front of the capture.
        HIP_TRY(c, run_rx_prepare_ref(n_frames, (const int32_t *)w_dhat, frame_len, c->S(), max_symbols, bps_bytes, status, offset,
                                      (int32_t *)w_off, (int32_t *)w_nsym, c->stream));
        offs = (int32_t *)w_off;
        if (c->prm.cfo_mode == OFDM_CFO_OFF) HIP_TRY(c, hipMemsetAsync(fd, 0, sizeof(double) * (size_t)n_frames, c->stream));
        else { c->trace.add("k_freq_corr"); HIP_TRY(c, run_freq_correction(x + 3 * c->S(), n_frames, frame_stride, c->S(), c->S(), fd, c->stream, offs, status)); }
    } else {
    // 1. timing + CFO: Schmidl-Cox over the repeated preamble (replaces xcorr_fft, src/receiver.rs:20-25,39)
    rc = ofdm_abi_sc_run(c, x, n_frames, frame_stride, frame_len, n_lags, (int32_t *)w_dhat, fd, metric);
    if (rc) return rc;
    // 2. trimmed start, length check, live symbols (receiver.rs:21-36)
    c->trace.add("k_rx_prepare");
    HIP_TRY(c, run_rx_prepare(n_frames, (const int32_t *)w_dhat, fd, frame_len, c->S(), c->prm.sync_backoff,
                              c->prm.cfo_mode, max_symbols, bps_bytes, status, offs, (int32_t *)w_nsym, c->stream));
    }
    // 3+4. channel estimate from the 5 training blocks and per data symbol CP strip + FFT + equalise + pilot phase +
    //      demap (receiver.rs:44-83).  N = 64: one fused wave-centric kernel; otherwise the generic pair.
    bool fused = false, finished = false;
    if (N == 1024) { // one workgroup per frame: channel estimate kept in registers, 16 x 64 FFT (kernels_fast.hip)
        const bool off = c->tune.no_rxframe1024 != 0; // A/B switch
        SymParams p = base_params(c);
        p.in = x; p.n_frames = n_frames; p.frame_stride = frame_stride; p.frame_len = frame_len;
        p.offset = offs; p.f_delta = fd; p.nsym_frame = (const int32_t *)w_nsym;
        p.out_bytes = (uint8_t *)w_raw; p.out_stride = raw_stride;
        // the kernel also parses the length header, truncates and Hamming-decodes into the caller's rows when they are 4-byte aligned
        bool fin = false;
        hipError_t e = off ? hipErrorNotSupported
                           : run_rxframe1024(p, nullptr, c->stream, c->num_cu, out, out_stride, out_len, c->prm.ecc, &fin);
        if (e == hipSuccess) { fused = true; finished = fin; }
        else if (e != hipErrorNotSupported) { c->last_hip = (int)e; return OFDM_ERR_HIP; }
    }
    if (N == 64) {
        SymParams p = base_params(c);
        p.in = x; p.n_frames = n_frames; p.frame_stride = frame_stride; p.frame_len = frame_len;
        p.offset = offs; p.f_delta = fd; p.nsym_frame = (const int32_t *)w_nsym;
        p.out_bytes = (uint8_t *)w_raw; p.out_stride = raw_stride;
        p.syms_per_frame = max_symbols; // (the launcher's "is there room for a whole frame in the capture" test)
        // without an outer code the kernel also parses the length header and writes the payload to its final place
        const bool fin = c->prm.ecc == OFDM_ECC_NONE && (reinterpret_cast<uintptr_t>(out) & 3) == 0 && (out_stride & 3) == 0;
        void *w_cut;
        if ((rc = ws_get(c, 8, sizeof(int32_t) * (size_t)(n_frames + 4), &w_cut))) return rc;
        hipError_t e = fin ? run_rxframe64(p, nullptr, c->stream, c->num_cu, out, out_stride, out_len, nullptr, nullptr, (int32_t *)w_cut)
                           : run_rxframe64(p, nullptr, c->stream, c->num_cu, nullptr, 0, nullptr, nullptr, nullptr, (int32_t *)w_cut);
        if (e == hipSuccess) { fused = true; finished = fin; }
        else if (e != hipErrorNotSupported) { c->last_hip = (int)e; return OFDM_ERR_HIP; }
    }
    if (!fused) {
        if ((rc = ws_get(c, 4, sizeof(float2) * (size_t)N * (size_t)n_frames, &w_hk))) return rc;
        SymParams p = base_params(c);
        p.in = x; p.out = (float2 *)w_hk; p.n_frames = n_frames; p.frame_stride = frame_stride; p.frame_len = frame_len;
        p.offset = offs; p.f_delta = fd;
        HIP_TRY(c, run_chest(N, p, c->stream, c->num_cu));
        rc = demod_run(c, x, n_frames, frame_stride, frame_len, 10, max_symbols, offs, fd, (const int32_t *)w_nsym,
                       (const float2 *)w_hk, N, (uint8_t *)w_raw, raw_stride, nullptr);
        if (rc) return rc;
    }
    // 5. length header, truncate [, Hamming decode] (receiver.rs:85-95)
    if (!finished) {
        c->trace.add("k_rx_finish");
        HIP_TRY(c, run_rx_finish((const uint8_t *)w_raw, raw_stride, n_frames, status, (const int32_t *)w_nsym, bps_bytes,
                                 c->prm.ecc, out, out_stride, out_len, c->stream));
    }
    return OFDM_OK;
}

// CHANNEL (src/channel.rs:26-31): 64 taps, the non-zero ones are 8..16 and 18
static const double kChannelTaps[64] = {
    0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -0.0000, -0.1912, 0.9316, 0.2821, -0.1990, 0.1630, -0.1017, 0.0544, -0.0261,
    0.0090, 0.0000, -0.0034};

int ofdm_channel_taps(double *taps64) {
    if (!taps64) return OFDM_ERR_INVALID;
    for (int i = 0; i < 64; i++) taps64[i] = kChannelTaps[i];
    return OFDM_OK;
}

int ofdm_channel_batch(ofdm_ctx *c, const ofdm_fc32 *tx, int64_t n_frames, int64_t tx_stride, int64_t tx_len, double snr_db,
                       int32_t timing_error, uint64_t seed, const int32_t *delay, const double *f_delta_in, ofdm_fc32 *out,
                       int64_t out_stride, int64_t out_len, double *f_delta_out) {
    if (!c || n_frames < 0 || tx_len <= 0 || tx_stride < 0 || out_len < tx_len + 63 || out_stride < out_len) return OFDM_ERR_INVALID;
    if (n_frames && (!tx || !out)) return OFDM_ERR_INVALID;
    if (n_frames > 1 && tx_stride <= 0) return OFDM_ERR_INVALID;
    if (!(snr_db == snr_db)) return OFDM_ERR_INVALID;
    if (!n_frames) return OFDM_OK;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    ChannelParams p;
    p.tx = reinterpret_cast<const float2 *>(tx); p.n_frames = n_frames; p.tx_stride = tx_stride; p.tx_len = tx_len;
    p.snr_lin = std::pow(10.0, snr_db / 10.0); // channel.rs:40
    p.timing_error = timing_error != 0; p.seed = seed; p.delay = delay; p.f_delta_in = f_delta_in;
    p.out = reinterpret_cast<float2 *>(out); p.out_stride = out_stride; p.out_len = out_len; p.f_delta_out = f_delta_out;
    for (int i = 0; i < 64; i++)
        if (kChannelTaps[i] != 0.0 && p.n_taps < 16) { p.tap_idx[p.n_taps] = i; p.tap_val[p.n_taps] = (float)kChannelTaps[i]; p.n_taps++; }
    HIP_TRY(c, run_channel(p, c->num_cu, c->stream));
    return OFDM_OK;
}

int ofdm_hbm_read_probe(ofdm_ctx *c, const ofdm_fc32 *in, int64_t n_symbols, int32_t pattern) {
    if (!c || n_symbols < 0 || pattern < 0 || pattern > 2 || (n_symbols && !in)) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->trace.reset();
    void *sink;
    int rc = ws_get(c, 7, 64, &sink);
    if (rc) return rc;
    HIP_TRY(c, run_read_probe(reinterpret_cast<const float2 *>(in), n_symbols, pattern, (unsigned *)sink, c->num_cu, c->stream));
    return OFDM_OK;
}

int ofdm_timer_start(ofdm_ctx *c) {
    if (!c) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    return OFDM_OK;
}
int ofdm_timer_stop_ms(ofdm_ctx *c, float *ms) {
    if (!c || !ms) return OFDM_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return OFDM_OK;
}

} // extern "C"
