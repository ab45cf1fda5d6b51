// outer_code.hip -- the demos' outer Reed-Solomon(255,223) framing (src/utils.rs:97-180), host side, byte level.
//
// The reference applies RS outside encode/decode, with reed-solomon = "0.2.1" (Cargo.toml:35; not under
// /root/reference).  That crate is the "Reed-Solomon codes for coders" construction: GF(2^8) with primitive polynomial
// 0x11d, generator element 2, generator polynomial prod_{i<32} (x - 2^i) (first consecutive root 0), code word =
// data followed by the 32 remainder bytes.  The encoder is pinned by that text's published vectors
// (tests/test_abi_cpu.py); the decoder (syndromes, Berlekamp-Massey, Chien, Forney) corrects up to 16 bytes per block,
// which is unique decoding: any correct decoder returns the same bytes.  Beyond 16 errors the crate's behaviour
// (failure vs. miscorrection) is not reproduced: "parity unpinned".
// This is off the roofline by design: tens of bytes per OFDM symbol, on the host.
#include "../../include/ofdm_hip.h"
#include <cstring>
#include <vector>

namespace {
constexpr int kN = 255, kK = 223, kPar = kN - kK;

struct Gf {
    uint8_t exp[512], log[256];
    uint8_t gen[kPar + 1]; // generator polynomial, highest degree first
    Gf() {
        int x = 1;
        for (int i = 0; i < 255; i++) { exp[i] = (uint8_t)x; log[x] = (uint8_t)i; x <<= 1; if (x & 0x100) x ^= 0x11d; }
        for (int i = 255; i < 512; i++) exp[i] = exp[i - 255];
        log[0] = 0;
        uint8_t g[kPar + 1] = {1};
        int deg = 0;
        for (int i = 0; i < kPar; i++) { // g *= (x - 2^i)
            uint8_t ng[kPar + 1] = {0};
            for (int j = 0; j <= deg; j++) { ng[j] ^= g[j]; ng[j + 1] ^= mul(g[j], exp[i]); }
            ++deg;
            std::memcpy(g, ng, sizeof g);
        }
        std::memcpy(gen, g, sizeof gen);
    }
    uint8_t mul(uint8_t a, uint8_t b) const { return (a && b) ? exp[log[a] + log[b]] : 0; }
    uint8_t div(uint8_t a, uint8_t b) const { return a ? exp[log[a] + 255 - log[b]] : 0; } // b != 0
    uint8_t inv(uint8_t a) const { return exp[255 - log[a]]; }
};
const Gf &gf() { static const Gf g; return g; }

void encode_block(const uint8_t *data /*223*/, uint8_t *out /*255*/) {
    const Gf &f = gf();
    uint8_t rem[kPar] = {0};
    for (int i = 0; i < kK; i++) { // synthetic division by the monic generator
        const uint8_t c = data[i] ^ rem[0];
        std::memmove(rem, rem + 1, kPar - 1);
        rem[kPar - 1] = 0;
        if (c) for (int j = 0; j < kPar; j++) rem[j] ^= f.mul(f.gen[j + 1], c);
    }
    std::memcpy(out, data, kK);
    std::memcpy(out + kK, rem, kPar);
}

// Corrects `cw` in place. Returns the number of corrected bytes, or -1 if the block cannot be corrected.
int correct_block(uint8_t *cw /*255*/) {
    const Gf &f = gf();
    uint8_t syn[kPar];
    bool clean = true;
    for (int i = 0; i < kPar; i++) { // S_i = C(2^i), cw[0] is the highest-degree coefficient
        uint8_t s = 0;
        for (int j = 0; j < kN; j++) s = f.mul(s, f.exp[i]) ^ cw[j];
        syn[i] = s;
        clean = clean && s == 0;
    }
    if (clean) return 0;
    // Berlekamp-Massey: sigma(x), lowest degree first
    uint8_t sigma[kPar + 2] = {1}, prev[kPar + 2] = {1};
    int L = 0, m = 1;
    uint8_t b = 1;
    for (int n = 0; n < kPar; n++) {
        uint8_t d = syn[n];
        for (int i = 1; i <= L; i++) d ^= f.mul(sigma[i], syn[n - i]);
        if (d == 0) { ++m; continue; }
        uint8_t t[kPar + 2];
        std::memcpy(t, sigma, sizeof t);
        const uint8_t coef = f.div(d, b);
        for (int i = 0; i + m < kPar + 2; i++) sigma[i + m] ^= f.mul(coef, prev[i]);
        if (2 * L <= n) { L = n + 1 - L; std::memcpy(prev, t, sizeof prev); b = d; m = 1; } else ++m;
    }
    if (L > kPar / 2) return -1;
    // Chien search: error at byte index j (degree 254 - j) <=> sigma(X^-1) = 0 with X = 2^(254 - j)
    int pos[kPar / 2], nerr = 0;
    for (int j = 0; j < kN; j++) {
        const int xinv = (255 - (254 - j)) % 255; // log of X^-1
        uint8_t v = 0;
        for (int i = L; i >= 0; i--) v = f.mul(v, f.exp[xinv]) ^ sigma[i];
        if (v == 0) { if (nerr == kPar / 2) return -1; pos[nerr++] = j; }
    }
    if (nerr != L) return -1;
    // Forney (first consecutive root 0): e = X * Omega(X^-1) / sigma'(X^-1), Omega = S(x) sigma(x) mod x^32
    uint8_t omega[kPar] = {0};
    for (int i = 0; i < kPar; i++)
        for (int j = 0; j <= L && j <= i; j++) omega[i] ^= f.mul(sigma[j], syn[i - j]);
    for (int k = 0; k < nerr; k++) {
        const int lx = 254 - pos[k], lxi = (255 - lx) % 255;
        uint8_t num = 0, den = 0;
        for (int i = kPar - 1; i >= 0; i--) num = f.mul(num, f.exp[lxi]) ^ omega[i];
        for (int i = 1; i <= L; i += 2) { // formal derivative: odd-degree terms, evaluated at X^-1
            uint8_t term = sigma[i];
            for (int e = 0; e < i - 1; e++) term = f.mul(term, f.exp[lxi]);
            den ^= term;
        }
        if (den == 0) return -1;
        cw[pos[k]] ^= f.mul(f.exp[lx], f.div(num, den));
    }
    for (int i = 0; i < kPar; i++) { // a corrected block must be a code word
        uint8_t s = 0;
        for (int j = 0; j < kN; j++) s = f.mul(s, f.exp[i]) ^ cw[j];
        if (s) return -1;
    }
    return nerr;
}
} // namespace

extern "C" {

int64_t ofdm_rs255_encoded_len(int64_t n_bytes) { return n_bytes < 0 ? 0 : (n_bytes / kK + 1) * kN; }
int64_t ofdm_rs255_decoded_len(int64_t n_code) { return n_code < 0 ? 0 : (n_code / kN + 1) * kK; }

int ofdm_rs255_encode(const uint8_t *data, int64_t n_bytes, uint8_t *out) {
    if (n_bytes < 0 || !out || (n_bytes > 0 && !data)) return OFDM_ERR_INVALID;
    // create_transmission_bytes (src/utils.rs:97-136): full 223-byte blocks, then ALWAYS one more zero-padded block
    const int64_t blocks = n_bytes / kK + 1;
    for (int64_t b = 0; b < blocks; b++) {
        uint8_t buf[kK] = {0};
        const int64_t have = n_bytes - b * kK;
        if (have > 0) std::memcpy(buf, data + b * kK, (size_t)(have < kK ? have : kK));
        encode_block(buf, out + b * kN);
    }
    return OFDM_OK;
}

int ofdm_rs255_decode(const uint8_t *code, int64_t n_code, uint8_t *out, int32_t *corrected) {
    if (n_code < 0 || !out || (n_code > 0 && !code)) return OFDM_ERR_INVALID;
    // decipher_transmission_bytes (src/utils.rs:150-180): 255-byte chunks, then the zero-padded remainder (even if empty)
    const int64_t blocks = n_code / kN + 1;
    int32_t total = 0;
    for (int64_t b = 0; b < blocks; b++) {
        uint8_t buf[kN] = {0};
        const int64_t have = n_code - b * kN;
        if (have > 0) std::memcpy(buf, code + b * kN, (size_t)(have < kN ? have : kN));
        const int r = correct_block(buf);
        if (r < 0) return OFDM_ERR_UNCORRECTABLE; // the reference returns None
        total += r;
        std::memcpy(out + b * kK, buf, kK);
    }
    if (corrected) *corrected = total;
    return OFDM_OK;
}

} // extern "C"
