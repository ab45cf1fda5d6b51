/*
 * ofdm_oracle.c -- CPU (f64) restatement of the jkelleyrtp/ofdm TX/RX hot path.
 * TEST INFRASTRUCTURE ONLY -- see ofdm_oracle.h for the rules and the pinning status.
 *
 * Plain C99, no dependencies beyond libm.  Arithmetic follows the reference line by line where the
 * reference has code (citations in ofdm_oracle.h and at each function); where it relies on a crate that
 * is not under /root/reference (rustfft 6, num 0.3 Complex64, bincode 1.3) the crate's published
 * contract is restated: unnormalised forward DFT, naive complex mul/div, exp via from_polar,
 * fixint little-endian u128 header.
 */
#include "ofdm_oracle.h"

#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ num::Complex64 arithmetic */
static inline oc64 c_new(double re, double im) { oc64 z = {re, im}; return z; }
static inline oc64 c_add(oc64 a, oc64 b) { return c_new(a.re + b.re, a.im + b.im); }
static inline oc64 c_sub(oc64 a, oc64 b) { return c_new(a.re - b.re, a.im - b.im); }
static inline oc64 c_mul(oc64 a, oc64 b) { return c_new(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline oc64 c_conj(oc64 a) { return c_new(a.re, -a.im); }
static inline double c_norm_sqr(oc64 a) { return a.re * a.re + a.im * a.im; }
/* num-complex Div: naive (non-Smith) form */
static inline oc64 c_div(oc64 a, oc64 b) {
    double ns = b.re * b.re + b.im * b.im;
    return c_new((a.re * b.re + a.im * b.im) / ns, (a.im * b.re - a.re * b.im) / ns);
}
static inline oc64 c_scale(oc64 a, double s) { return c_new(a.re * s, a.im * s); }
/* num-complex exp: from_polar(exp(re), im) */
static inline oc64 c_exp(oc64 a) {
    double r = exp(a.re);
    return c_new(r * cos(a.im), r * sin(a.im));
}

/* ------------------------------------------------------------------ PRNG (documented; NOT rand::StdRng) */
uint64_t orc_splitmix64(uint64_t *state) {
    uint64_t z = (*state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
double orc_uniform_01(uint64_t *state) { return (double)(orc_splitmix64(state) >> 11) * (1.0 / 9007199254740992.0); }
double orc_uniform_pm1(uint64_t *state) { return orc_uniform_01(state) * 2.0 - 1.0; }

/* ------------------------------------------------------------------ FFT (rustfft contract: unnormalised DFT) */
static int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

/* Twiddle policy.  Default (cache off): every transform takes its twiddles from libm, as the reference builds a new
 * FftPlanner + plan (twiddle generation included) on EVERY call (signals/mod.rs:41-58) -- the "ref-faithful" CPU baseline.
 * orc_set_fft_cache(1): per-thread tables exp(-2 pi i k / n), k < n/2, built once per size -- the "ref-optimised" baseline.
 * Results are bit-identical either way: k/len and (k n/len)/n are the same double (n/len is a power of two). */
static int g_fft_cache = 0;
void orc_set_fft_cache(int on) { g_fft_cache = on; }
static __thread oc64 *t_fft_tab[32];
static const oc64 *fft_table(int n) {
    int lg = 0;
    while ((1 << lg) < n) lg++;
    if (!t_fft_tab[lg]) {
        oc64 *t = (oc64 *)malloc(sizeof(oc64) * (size_t)(n / 2 > 0 ? n / 2 : 1));
        for (int k = 0; k < n / 2; k++) {
            double ang = -2.0 * M_PI * (double)k / (double)n;
            t[k] = c_new(cos(ang), sin(ang));
        }
        t_fft_tab[lg] = t; /* lives as long as the thread */
    }
    return t_fft_tab[lg];
}

static void fft_pow2(oc64 *x, int n, int inverse) {
    /* iterative radix-2 DIT, twiddles from libm per stage index (no recurrence, keeps 1e-15 accuracy) */
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { oc64 t = x[i]; x[i] = x[j]; x[j] = t; }
    }
    double sgn = inverse ? 1.0 : -1.0;
    const oc64 *tab = g_fft_cache ? fft_table(n) : NULL;
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1;
        for (int k = 0; k < half; k++) {
            oc64 w;
            if (tab) { w = tab[k * (n / len)]; if (inverse) w.im = -w.im; }
            else {
                double ang = sgn * 2.0 * M_PI * (double)k / (double)len;
                w = c_new(cos(ang), sin(ang));
            }
            for (int i = k; i < n; i += len) {
                oc64 u = x[i], v = c_mul(x[i + half], w);
                x[i] = c_add(u, v);
                x[i + half] = c_sub(u, v);
            }
        }
    }
}

static void fft_bluestein(oc64 *x, int n, int inverse) {
    int m = 1;
    while (m < 2 * n - 1) m <<= 1;
    oc64 *w = (oc64 *)malloc(sizeof(oc64) * (size_t)n);
    oc64 *a = (oc64 *)calloc((size_t)m, sizeof(oc64));
    oc64 *b = (oc64 *)calloc((size_t)m, sizeof(oc64));
    double sgn = inverse ? 1.0 : -1.0;
    for (int i = 0; i < n; i++) {
        long long sq = ((long long)i * (long long)i) % (2LL * n); /* exact reduction of i^2 mod 2n */
        double ang = sgn * M_PI * (double)sq / (double)n;
        w[i] = c_new(cos(ang), sin(ang));
    }
    for (int i = 0; i < n; i++) a[i] = c_mul(x[i], w[i]);
    b[0] = c_conj(w[0]);
    for (int i = 1; i < n; i++) b[i] = b[m - i] = c_conj(w[i]);
    fft_pow2(a, m, 0);
    fft_pow2(b, m, 0);
    for (int i = 0; i < m; i++) a[i] = c_mul(a[i], b[i]);
    fft_pow2(a, m, 1);
    for (int i = 0; i < n; i++) x[i] = c_mul(c_scale(a[i], 1.0 / (double)m), w[i]);
    free(w); free(a); free(b);
}

/* signals/mod.rs:27-58: fft = unnormalised forward; ifft = unnormalised inverse then normalize_by(1/len) */
void orc_fft(oc64 *x, int n, int inverse) {
    if (n <= 1) return;
    if (is_pow2(n)) fft_pow2(x, n, inverse);
    else fft_bluestein(x, n, inverse);
    if (inverse) {
        double s = 1.0 / (double)n;
        for (int i = 0; i < n; i++) x[i] = c_scale(x[i], s); /* normalize_by: *val *= scale (mod.rs:132-138) */
    }
}

static void rotate_left(oc64 *x, int n, int mid) {
    oc64 *t = (oc64 *)malloc(sizeof(oc64) * (size_t)n);
    memcpy(t, x, sizeof(oc64) * (size_t)n);
    for (int i = 0; i < n; i++) x[i] = t[(i + mid) % n]; /* r.iter().chain(l.iter()) */
    free(t);
}
void orc_fft_shift(oc64 *x, int n) { if (n > 0) rotate_left(x, n, (n + 1) / 2); }  /* mod.rs:65: mid=floor((len+1)/2) */
void orc_ifft_shift(oc64 *x, int n) { if (n > 0) rotate_left(x, n, n / 2); }       /* mod.rs:84: mid=floor(len/2) */

/* signals/mod.rs:186-217 */
int orc_xcorr_fft(const oc64 *a_in, int na, const oc64 *b_in, int nb, oc64 *out) {
    int pad = 2 * na - 1;
    oc64 *a = out;
    oc64 *b = (oc64 *)calloc((size_t)pad, sizeof(oc64));
    memset(a, 0, sizeof(oc64) * (size_t)pad);
    memcpy(a, a_in, sizeof(oc64) * (size_t)na);
    memcpy(b, b_in, sizeof(oc64) * (size_t)(nb < pad ? nb : pad));
    orc_fft(a, pad, 0);
    orc_fft(b, pad, 0);
    for (int i = 0; i < pad; i++) a[i] = c_mul(a[i], c_conj(b[i]));
    orc_fft(a, pad, 1);
    orc_fft_shift(a, pad);
    free(b);
    double max = 0.0; /* Complex64::default().norm_sqr() */
    int idx_max = 0;
    for (int i = 0; i < pad; i++) {
        double v = c_norm_sqr(a[i]);
        if (v > max) { idx_max = i; max = v; }
    }
    return idx_max;
}

/* signals/mod.rs:219-237 */
void orc_convolve(const oc64 *a_in, int na, const oc64 *b_in, int nb, oc64 *out) {
    int pad = na + nb - 1;
    oc64 *b = (oc64 *)calloc((size_t)pad, sizeof(oc64));
    memset(out, 0, sizeof(oc64) * (size_t)pad);
    memcpy(out, a_in, sizeof(oc64) * (size_t)na);
    memcpy(b, b_in, sizeof(oc64) * (size_t)nb);
    orc_fft(out, pad, 0);
    orc_fft(b, pad, 0);
    for (int i = 0; i < pad; i++) out[i] = c_mul(out[i], b[i]);
    orc_fft(out, pad, 1);
    free(b);
}

/* signals/mod.rs:251-259 */
oc64 orc_mean(const oc64 *x, int n) {
    oc64 s = c_new(0, 0);
    for (int i = 0; i < n; i++) s = c_add(s, x[i]);
    s.re /= (double)n;
    s.im /= (double)n;
    return s;
}
/* signals/mod.rs:239-249: sum((mean - x)^2)/len, complex square, NOT conjugated */
oc64 orc_variance(const oc64 *x, int n) {
    oc64 m = orc_mean(x, n), s = c_new(0, 0);
    for (int i = 0; i < n; i++) {
        oc64 d = c_sub(m, x[i]);
        s = c_add(s, c_mul(d, d));
    }
    return c_new(s.re / (double)n, s.im / (double)n);
}
/* receiver.rs:242-246 */
double orc_angle(oc64 z) { return atan2(z.im, z.re); }

/* ------------------------------------------------------------------ utils.rs */
void orc_to_bools(uint8_t b, uint8_t out[8]) { for (int i = 0; i < 8; i++) out[i] = (b & (1u << i)) != 0; }
uint8_t orc_bools_to_u8(const uint8_t in[8]) {
    uint8_t o = 0;
    for (int i = 0; i < 8; i++) o |= (uint8_t)((in[i] ? 1u : 0u) << i);
    return o;
}
void orc_analysis(const uint8_t *l, const uint8_t *r, size_t n, uint32_t *num_errs, uint32_t *num_block_errs,
                  double *err_rate) {
    uint32_t e = 0, be = 0;
    for (size_t i = 0; i < n; i++) {
        if (l[i] != r[i]) { e += (uint32_t)__builtin_popcount((unsigned)(l[i] ^ r[i])); be += 1; }
    }
    *num_errs = e;
    *num_block_errs = be;
    *err_rate = (double)e / ((double)n * 8.0);
}
void orc_sig_to_fc32(const oc64 *x, size_t n, float *out) {
    for (size_t i = 0; i < n; i++) { out[2 * i] = (float)x[i].re; out[2 * i + 1] = (float)x[i].im; }
}
void orc_fc32_to_sig(const float *in, size_t n, oc64 *out) {
    for (size_t i = 0; i < n; i++) out[i] = c_new((double)in[2 * i], (double)in[2 * i + 1]);
}

/* ------------------------------------------------------------------ carrier map */
int orc_carrier_class(int bin, int n_fft, int guard) {
    if (!guard) return 0;
    int i = bin / (n_fft / 64); /* EXT-4: class of bin = reference class of floor(bin/k) */
    if (i >= 59 || i <= 5 || i == 32) return 1;                 /* transmitter.rs:153 */
    if (i == 6 || i == 25 || i == 39 || i == 58) return 2;      /* transmitter.rs:156 */
    return 0;
}
int orc_data_carriers(int n_fft, int guard) { return guard ? 48 * (n_fft / 64) : n_fft; }

/* ------------------------------------------------------------------ transmitter.rs */
void orc_locking_signal(int len, oc64 *out) {
    for (int i = 0; i < len; i++) out[i] = c_new(0.5 * ((double)i / (2.0 * (double)len) + 0.5), 0.0);
    orc_fft_shift(out, len);
}
void orc_default_preamble(int len, oc64 *out) {
    uint64_t st = 100;
    for (int i = 0; i < len; i++) {
        double re = orc_uniform_pm1(&st), im = orc_uniform_pm1(&st);
        out[i] = c_scale(c_new(re, im), 0.25);
    }
}
void orc_default_training(int len, oc64 *out) {
    uint64_t st = 50;
    for (int i = 0; i < len; i++) {
        double re = orc_uniform_pm1(&st), im = orc_uniform_pm1(&st);
        out[i] = c_scale(c_new(re, im), 1.0);
    }
}

/* ------------------------------------------------------------------ rand 0.8 StdRng restated (ChaCha12 keystream)
 * transmitter.rs:76-80,89-93 draw the pilot tables from rand::rngs::StdRng::seed_from_u64(seed) with
 * gen_range(-1.0..1.0).  rand = "0.8.3" (Cargo.toml:23; Cargo.lock is git-ignored) is not under /root/reference, so
 * this follows the crates' published algorithm: rand_core 0.6 seed_from_u64 (PCG32 per 4 key bytes), rand_chacha 0.3
 * ChaCha12 with a 64-bit block counter and stream id 0, BlockRng::next_u64 = two consecutive keystream words (low
 * first), UniformFloat<f64>::sample_single = ((u64 >> 12) as mantissa of [1,2)) - 1, then * (high-low) + low.
 * The keystream is pinned by published ChaCha vectors in the tests; the rest is unverified ("parity unpinned"). */
typedef struct { uint32_t in[16]; uint32_t ks[16]; int used; } orc_stdrng;
#define ORC_ROTL(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define ORC_QR(a, b, c, d) \
    a += b; d ^= a; d = ORC_ROTL(d, 16); c += d; b ^= c; b = ORC_ROTL(b, 12); \
    a += b; d ^= a; d = ORC_ROTL(d, 8);  c += d; b ^= c; b = ORC_ROTL(b, 7);
void orc_chacha_keystream_block(const uint32_t in[16], int rounds, uint32_t out[16]) {
    uint32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3], x4 = in[4], x5 = in[5], x6 = in[6], x7 = in[7], x8 = in[8],
             x9 = in[9], x10 = in[10], x11 = in[11], x12 = in[12], x13 = in[13], x14 = in[14], x15 = in[15];
    for (int i = rounds; i > 0; i -= 2) {
        ORC_QR(x0, x4, x8, x12) ORC_QR(x1, x5, x9, x13) ORC_QR(x2, x6, x10, x14) ORC_QR(x3, x7, x11, x15)
        ORC_QR(x0, x5, x10, x15) ORC_QR(x1, x6, x11, x12) ORC_QR(x2, x7, x8, x13) ORC_QR(x3, x4, x9, x14)
    }
    const uint32_t x[16] = {x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15};
    for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}
static void orc_stdrng_seed(orc_stdrng *r, uint64_t seed) {
    static const uint32_t sigma[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u}; /* "expand 32-byte k" */
    for (int i = 0; i < 4; i++) r->in[i] = sigma[i];
    uint64_t pcg = seed;
    for (int i = 0; i < 8; i++) {
        pcg = pcg * 6364136223846793005ULL + 11634580027462260723ULL;
        uint32_t xs = (uint32_t)(((pcg >> 18) ^ pcg) >> 27), rot = (uint32_t)(pcg >> 59);
        r->in[4 + i] = rot ? ((xs >> rot) | (xs << (32 - rot))) : xs;
    }
    r->in[12] = r->in[13] = r->in[14] = r->in[15] = 0;
    r->used = 16;
}
static uint32_t orc_stdrng_word(orc_stdrng *r) {
    if (r->used == 16) {
        orc_chacha_keystream_block(r->in, 12, r->ks);
        if (++r->in[12] == 0) ++r->in[13];
        r->used = 0;
    }
    return r->ks[r->used++];
}
static double orc_stdrng_pm1(orc_stdrng *r) {
    uint64_t lo = orc_stdrng_word(r), hi = orc_stdrng_word(r);
    union { uint64_t u; double d; } v;
    v.u = (((hi << 32) | lo) >> 12) | ((uint64_t)1023 << 52);
    return (v.d - 1.0) * (1.0 - -1.0) + -1.0;
}
void orc_stdrng_preamble(int len, oc64 *out) {
    orc_stdrng r;
    orc_stdrng_seed(&r, 100);
    for (int i = 0; i < len; i++) {
        double re = orc_stdrng_pm1(&r), im = orc_stdrng_pm1(&r);
        out[i] = c_scale(c_new(re, im), 0.25);
    }
}
void orc_stdrng_training(int len, oc64 *out) {
    orc_stdrng r;
    orc_stdrng_seed(&r, 50);
    for (int i = 0; i < len; i++) {
        double re = orc_stdrng_pm1(&r), im = orc_stdrng_pm1(&r);
        out[i] = c_scale(c_new(re, im), 1.0);
    }
}

/* ------------------------------------------------------------------ outer Reed-Solomon framing, utils.rs:97-180
 * reed-solomon = "0.2.1" (Cargo.toml:35) is not under /root/reference.  Its published construction: GF(2^8) mod
 * x^8+x^4+x^3+x^2+1 (0x11d), alpha = 2, generator prod_{i<nsym}(x - alpha^i), code word = message then remainder.
 * Encoder pinned by the published "Reed-Solomon codes for coders" vectors (tests).  The decoder here solves the key
 * equation with the extended Euclidean algorithm (the library uses Berlekamp-Massey): two routes to the same unique
 * answer for <= nsym/2 errors. */
static uint8_t rs_exp[512], rs_log[256];
static void rs_init(void) {
    if (rs_exp[0]) return;
    int x = 1;
    for (int i = 0; i < 255; i++) { rs_exp[i] = (uint8_t)x; rs_log[x] = (uint8_t)i; x <<= 1; if (x & 0x100) x ^= 0x11d; }
    for (int i = 255; i < 512; i++) rs_exp[i] = rs_exp[i - 255];
}
static uint8_t rs_mul(uint8_t a, uint8_t b) { return (a && b) ? rs_exp[rs_log[a] + rs_log[b]] : 0; }
static uint8_t rs_inv(uint8_t a) { return rs_exp[255 - rs_log[a]]; }
static uint8_t rs_eval_lo(const uint8_t *p, int deg, uint8_t x) { /* p[i] multiplies x^i */
    uint8_t v = 0;
    for (int i = deg; i >= 0; i--) v = rs_mul(v, x) ^ p[i];
    return v;
}
int orc_rs_encode(const uint8_t *msg, int n, int nsym, uint8_t *out) { /* out[n + nsym] */
    rs_init();
    if (n < 0 || nsym <= 0 || n + nsym > 255) return -1;
    uint8_t g[256] = {1};
    int deg = 0;
    for (int i = 0; i < nsym; i++) { /* highest degree first */
        uint8_t ng[256] = {0};
        for (int j = 0; j <= deg; j++) { ng[j] ^= g[j]; ng[j + 1] ^= rs_mul(g[j], rs_exp[i]); }
        memcpy(g, ng, sizeof g);
        deg++;
    }
    uint8_t buf[256 + 255] = {0};
    memcpy(buf, msg, (size_t)n);
    for (int i = 0; i < n; i++) {
        uint8_t c = buf[i];
        if (c) for (int j = 1; j <= nsym; j++) buf[i + j] ^= rs_mul(g[j], c);
    }
    memcpy(out, msg, (size_t)n);
    memcpy(out + n, buf + n, (size_t)nsym);
    return 0;
}
static int rs_deg(const uint8_t *p, int maxdeg) { while (maxdeg > 0 && p[maxdeg] == 0) maxdeg--; return maxdeg; }
/* corrects cw[len] in place (len <= 255); returns corrected count or -1 */
int orc_rs_correct(uint8_t *cw, int len, int nsym) {
    rs_init();
    uint8_t S[64] = {0};
    int any = 0;
    for (int i = 0; i < nsym; i++) {
        uint8_t v = 0;
        for (int j = 0; j < len; j++) v = rs_mul(v, rs_exp[i]) ^ cw[j];
        S[i] = v; any |= v;
    }
    if (!any) return 0;
    /* extended Euclid on (x^nsym, S(x)), polynomials lowest degree first */
    uint8_t r0[80] = {0}, r1[80] = {0}, t0[80] = {0}, t1[80] = {0};
    r0[nsym] = 1;
    memcpy(r1, S, (size_t)nsym);
    t1[0] = 1;
    int d0 = nsym, d1 = rs_deg(r1, nsym - 1);
    while (d1 >= nsym / 2 && !(d1 == 0 && r1[0] == 0)) {
        /* r0 = q r1 + rem ; t2 = t0 - q t1 */
        uint8_t rem[80], tt[80];
        memcpy(rem, r0, sizeof rem); memcpy(tt, t0, sizeof tt);
        int dr = d0;
        while (dr >= d1 && !(dr == 0 && rem[0] == 0)) {
            uint8_t c = rs_mul(rem[dr], rs_inv(r1[d1]));
            int sh = dr - d1;
            for (int i = 0; i <= d1; i++) rem[i + sh] ^= rs_mul(c, r1[i]);
            for (int i = 0; i + sh < 80; i++) tt[i + sh] ^= rs_mul(c, t1[i]);
            dr = rs_deg(rem, dr);
            if (dr == 0 && rem[0] == 0) break;
        }
        memcpy(r0, r1, sizeof r0); memcpy(t0, t1, sizeof t0); d0 = d1;
        memcpy(r1, rem, sizeof r1); memcpy(t1, tt, sizeof t1); d1 = rs_deg(r1, 79);
    }
    if (t1[0] == 0) return -1;
    uint8_t k = rs_inv(t1[0]), sigma[80], omega[80];
    for (int i = 0; i < 80; i++) { sigma[i] = rs_mul(t1[i], k); omega[i] = rs_mul(r1[i], k); }
    int L = rs_deg(sigma, 79);
    if (L > nsym / 2) return -1;
    int found = 0;
    for (int j = 0; j < len; j++) {
        int lx = len - 1 - j;                 /* X = alpha^lx */
        uint8_t xi = rs_exp[(255 - lx) % 255]; /* X^-1 */
        if (rs_eval_lo(sigma, L, xi) != 0) continue;
        uint8_t den = 0, p = 1;               /* sigma'(X^-1): odd-degree terms */
        for (int i = 1; i <= L; i += 2) { den ^= rs_mul(sigma[i], p); p = rs_mul(p, rs_mul(xi, xi)); }
        if (den == 0) return -1;
        uint8_t mag = rs_mul(rs_exp[lx % 255], rs_mul(rs_eval_lo(omega, nsym - 1, xi), rs_inv(den)));
        cw[j] ^= mag;
        found++;
    }
    if (found != L) return -1;
    for (int i = 0; i < nsym; i++) {
        uint8_t v = 0;
        for (int j = 0; j < len; j++) v = rs_mul(v, rs_exp[i]) ^ cw[j];
        if (v) return -1;
    }
    return found;
}
/* create_transmission_bytes (utils.rs:97-136): out holds 255 * (n / 223 + 1) bytes */
long orc_create_transmission_bytes(const uint8_t *data, long n, uint8_t *out) {
    long blocks = n / 223 + 1;
    for (long b = 0; b < blocks; b++) {
        uint8_t scratch[223] = {0};
        long have = n - b * 223;
        if (have > 0) memcpy(scratch, data + b * 223, (size_t)(have < 223 ? have : 223));
        orc_rs_encode(scratch, 223, 32, out + b * 255);
    }
    return blocks * 255;
}
/* decipher_transmission_bytes (utils.rs:150-180): out holds 223 * (n / 255 + 1) bytes; -1 = None */
long orc_decipher_transmission_bytes(const uint8_t *code, long n, uint8_t *out) {
    long blocks = n / 255 + 1;
    for (long b = 0; b < blocks; b++) {
        uint8_t scratch[255] = {0};
        long have = n - b * 255;
        if (have > 0) memcpy(scratch, code + b * 255, (size_t)(have < 255 ? have : 255));
        if (orc_rs_correct(scratch, 255, 32) < 0) return -1;
        memcpy(out + b * 223, scratch, 223);
    }
    return blocks * 223;
}

static inline int stream_bit(const uint8_t *bytes, size_t nbytes, size_t bit) {
    size_t by = bit >> 3;
    if (by >= nbytes) return 0;
    return (bytes[by] >> (bit & 7)) & 1; /* LSB-first, utils.rs:21-27 */
}
static inline unsigned gray_decode(unsigned g) {
    unsigned l = g;
    for (unsigned s = g >> 1; s; s >>= 1) l ^= s;
    return l;
}
/* EXT-1 per-axis level: first bit of the group is the Gray MSB (1 => positive half), levels in [-1,1] */
static double axis_level(const int *bits, int m) {
    unsigned g = 0;
    for (int i = 0; i < m; i++) g = (g << 1) | (unsigned)bits[i];
    unsigned l = gray_decode(g);
    unsigned M = 1u << m;
    return ((double)(2 * (int)l - (int)(M - 1))) / (double)(M - 1);
}
size_t orc_modulate_count(size_t nbytes, int modulation) {
    return (nbytes * 8 + (size_t)modulation - 1) / (size_t)modulation;
}
/* transmitter.rs:108-140 (Bpsk, Qpsk) + EXT-1 (16/64/256-QAM) */
size_t orc_modulate(const uint8_t *bytes, size_t nbytes, int modulation, oc64 *out) {
    size_t nsym = orc_modulate_count(nbytes, modulation);
    for (size_t s = 0; s < nsym; s++) {
        int bits[8];
        for (int j = 0; j < modulation; j++) bits[j] = stream_bit(bytes, nbytes, s * (size_t)modulation + (size_t)j);
        if (modulation == ORC_BPSK) {
            out[s] = c_new(bits[0] ? 1.0 : -1.0, 0.0);
        } else {
            int m = modulation / 2;
            out[s] = c_new(axis_level(bits, m), axis_level(bits + m, m));
        }
    }
    return nsym;
}
/* transmitter.rs:144-165 */
void orc_encode_block(const oc64 *stream, size_t avail, size_t *consumed, int n_fft, int guard, oc64 *out) {
    size_t used = 0;
    for (int i = 0; i < n_fft; i++) {
        int cls = orc_carrier_class(i, n_fft, guard);
        if (cls == 1) out[i] = c_new(0.0, 0.0);
        else if (cls == 2) out[i] = c_new(1.0, 0.0);
        else if (used < avail) out[i] = stream[used++];
        else out[i] = c_new(0.0, 0.0); /* unwrap_or_else(|| 0) */
    }
    *consumed = used;
}
/* transmitter.rs:168-181 */
void orc_prefix_block(const oc64 *freq, int n_fft, int cp, oc64 *out) {
    oc64 *t = (oc64 *)malloc(sizeof(oc64) * (size_t)n_fft);
    memcpy(t, freq, sizeof(oc64) * (size_t)n_fft);
    orc_fft(t, n_fft, 1);
    memcpy(out, t + (n_fft - cp), sizeof(oc64) * (size_t)cp);
    memcpy(out + cp, t, sizeof(oc64) * (size_t)n_fft);
    free(t);
}
/* transmitter.rs:183-194: signed max over re and im, starting from 0 */
void orc_normalize(oc64 *x, size_t n) {
    double max = 0.0;
    for (size_t i = 0; i < n; i++) { max = fmax(x[i].re, max); max = fmax(x[i].im, max); }
    for (size_t i = 0; i < n; i++) { x[i].re = x[i].re / max; x[i].im = x[i].im / max; }
}
size_t orc_frame_len(size_t payload_bytes, int n_fft, int cp, int guard, int modulation) {
    size_t S = (size_t)(n_fft + cp);
    size_t nsym = orc_modulate_count(16 + payload_bytes, modulation);
    size_t nd = (size_t)orc_data_carriers(n_fft, guard);
    return 10 * S + S * ((nsym + nd - 1) / nd);
}
/* transmitter.rs:11-58.  Header + payload are modulated as ONE byte stream (identical to the reference's
 * modulate(header).chain(modulate(data)) for BPSK/QPSK, where a byte is a whole number of symbols; required
 * for 64-QAM so that the receiver's "drain 16 bytes" (receiver.rs:86-93) stays byte-aligned). */
size_t orc_encode(const uint8_t *data, size_t nbytes, int n_fft, int cp, int guard, int modulation,
                  const oc64 *preamble, const oc64 *training, oc64 *out) {
    int S = n_fft + cp;
    size_t pos = 0;
    orc_locking_signal(S, out);
    pos += (size_t)S;
    for (int r = 0; r < 4; r++) { memcpy(out + pos, preamble, sizeof(oc64) * (size_t)S); pos += (size_t)S; }
    for (int r = 0; r < 5; r++) { orc_prefix_block(training, n_fft, cp, out + pos); pos += (size_t)S; }

    size_t tot = 16 + nbytes;
    uint8_t *stream_bytes = (uint8_t *)malloc(tot);
    memset(stream_bytes, 0, 16);
    uint64_t len64 = (uint64_t)nbytes; /* bincode fixint LE u128 (packets/mod.rs:20-32) */
    for (int i = 0; i < 8; i++) stream_bytes[i] = (uint8_t)(len64 >> (8 * i));
    memcpy(stream_bytes + 16, data, nbytes);
    size_t nsym = orc_modulate_count(tot, modulation);
    oc64 *sym = (oc64 *)malloc(sizeof(oc64) * (nsym ? nsym : 1));
    orc_modulate(stream_bytes, tot, modulation, sym);
    oc64 *blk = (oc64 *)malloc(sizeof(oc64) * (size_t)n_fft);
    size_t used = 0;
    while (used < nsym) { /* while complex_stream.peek().is_some() */
        size_t c = 0;
        orc_encode_block(sym + used, nsym - used, &c, n_fft, guard, blk);
        used += c;
        orc_prefix_block(blk, n_fft, cp, out + pos);
        pos += (size_t)S;
    }
    orc_normalize(out, pos);
    free(blk); free(sym); free(stream_bytes);
    return pos;
}

/* ------------------------------------------------------------------ receiver.rs */
void orc_unprefix_block(const oc64 *in, int n_fft, int cp, oc64 *out) {
    memcpy(out, in + cp, sizeof(oc64) * (size_t)n_fft);
    orc_fft(out, n_fft, 0);
}
/* receiver.rs:106-145.  pilot_count is 4 in the reference (64 carriers); 4k for n_fft = 64k (EXT-4). */
size_t orc_decode_block(const oc64 *in, int n_fft, int guard, oc64 *out) {
    double pilot_count = 4.0 * (double)(n_fft / 64);
    double phase = 0.0;
    size_t cnt = 0;
    for (int i = 0; i < n_fft; i++) {
        int cls = orc_carrier_class(i, n_fft, guard);
        if (cls == 1) continue;
        if (cls == 2) { phase = phase + orc_angle(c_div(in[i], c_new(1.0, 0.0))); continue; }
        out[cnt++] = in[i];
    }
    phase /= pilot_count;
    oc64 rot = c_exp(c_scale(c_new(0.0, -1.0), phase));
    for (size_t i = 0; i < cnt; i++) out[i] = c_mul(out[i], rot);
    return cnt;
}
static unsigned axis_decide(double x, int m) {
    unsigned M = 1u << m;
    double u = x * (double)(M - 1);
    if (!(u == u)) return 0; /* NaN */
    double f = floor(u * 0.5) + (double)(M / 2);
    if (f < 0.0) f = 0.0;
    if (f > (double)(M - 1)) f = (double)(M - 1);
    unsigned l = (unsigned)f;
    return l ^ (l >> 1); /* Gray code, MSB = first stream bit */
}
static void symbol_bits(oc64 s, int modulation, int *bits) {
    if (modulation == ORC_BPSK) {
        bits[0] = s.re > 0.0; /* receiver.rs:162 */
    } else if (modulation == ORC_QPSK) {
        /* receiver.rs:169-175 match arms, in order (tie rules Q6) */
        double re = s.re, im = s.im;
        int l, r;
        if (re >= 0.0 && im >= 0.0) { l = 1; r = 1; }
        else if (re >= 0.0 && im <= 0.0) { l = 1; r = 0; }
        else if (re < 0.0 && im > 0.0) { l = 0; r = 1; }
        else if (re < 0.0 && im < 0.0) { l = 0; r = 0; }
        else { l = 0; r = 0; }
        bits[0] = l; bits[1] = r;
    } else {
        int m = modulation / 2;
        unsigned gi = axis_decide(s.re, m), gq = axis_decide(s.im, m);
        for (int i = 0; i < m; i++) {
            bits[i] = (int)((gi >> (m - 1 - i)) & 1u);
            bits[m + i] = (int)((gq >> (m - 1 - i)) & 1u);
        }
    }
}
void orc_demap_indices(const oc64 *sym, size_t nsym, int modulation, uint8_t *idx) {
    for (size_t s = 0; s < nsym; s++) {
        int bits[8];
        symbol_bits(sym[s], modulation, bits);
        unsigned v = 0;
        for (int j = 0; j < modulation; j++) v |= (unsigned)bits[j] << j;
        idx[s] = (uint8_t)v;
    }
}
/* receiver.rs:147-190: 8 symbols per step, bits packed LSB-first; returns 0 if nsym % 8 != 0 (assert) */
size_t orc_demodulate(const oc64 *sym, size_t nsym, int modulation, uint8_t *out) {
    if (nsym % 8 != 0) return 0;
    size_t nbytes = nsym * (size_t)modulation / 8;
    memset(out, 0, nbytes);
    for (size_t s = 0; s < nsym; s++) {
        int bits[8];
        symbol_bits(sym[s], modulation, bits);
        for (int j = 0; j < modulation; j++) {
            size_t b = s * (size_t)modulation + (size_t)j;
            out[b >> 3] |= (uint8_t)(bits[j] << (b & 7));
        }
    }
    return nbytes;
}
/* receiver.rs:212-229 */
void orc_estimate_channel(const oc64 *blocks, int n_fft, int cp, const oc64 *training, oc64 *hk) {
    int S = n_fft + cp;
    oc64 *t = (oc64 *)malloc(sizeof(oc64) * (size_t)n_fft);
    for (int i = 0; i < n_fft; i++) hk[i] = c_new(0, 0);
    for (int b = 0; b < 5; b++) {
        orc_unprefix_block(blocks + (size_t)b * (size_t)S, n_fft, cp, t);
        for (int i = 0; i < n_fft; i++) hk[i] = c_add(hk[i], c_div(t[i], training[i]));
    }
    for (int i = 0; i < n_fft; i++) { hk[i].re /= 5.0; hk[i].im /= 5.0; }
    free(t);
}
/* receiver.rs:231-240 (80 -> len) */
double orc_frequency_correction(const oc64 *left, const oc64 *right, int len) {
    double sum = 0.0;
    for (int i = 0; i < len; i++) sum += orc_angle(c_div(right[i], left[i]));
    return fabs((sum / (double)len) / (double)len);
}
/* receiver.rs:44-50 */
void orc_cfo_rotate(oc64 *x, size_t n, double f_delta, size_t first_index) {
    for (size_t i = 0; i < n; i++) {
        oc64 e = c_scale(c_scale(c_new(0.0, -1.0), f_delta), (double)(first_index + i));
        x[i] = c_mul(x[i], c_exp(e));
    }
}

/* ------------------------------------------------------------------ EXT-3 Schmidl-Cox */
static void sc_at(const oc64 *r, long d, int L, int W, oc64 *P, double *E, double *R) {
    oc64 p = c_new(0, 0);
    double e = 0, q = 0;
    for (int m = 0; m < W; m++) {
        oc64 a = r[d + m], b = r[d + m + L];
        p = c_add(p, c_mul(c_conj(a), b));
        e += c_norm_sqr(a);
        q += c_norm_sqr(b);
    }
    *P = p; *E = e; *R = q;
}
int orc_sc_sync(const oc64 *r, size_t n, int L, int window_reps, long n_lags, double threshold, oc64 *p_hat,
                double *metric, double *f_delta) {
    int W = window_reps * L;
    long valid = (long)n - (long)W - (long)L + 1;
    if (n_lags <= 0 || n_lags > valid) n_lags = valid;
    long best = -1, d1 = -1, last = n_lags;
    double bnum = 0.0, bden = 1.0;
    oc64 bp = c_new(0, 0);
    for (long d = 0; d < n_lags && d <= last; d++) {
        oc64 P; double E, R;
        sc_at(r, d, L, W, &P, &E, &R);
        double num = c_norm_sqr(P), den = E * R;
        if (!(den > 0.0)) continue;
        if (d1 < 0) {
            if (!(num >= threshold * den)) continue; /* packet detect: first lag with M(d) >= threshold */
            d1 = d;
            last = d1 + (long)W; /* then the first maximum of M over [d1, d1 + W] */
            best = d; bnum = num; bden = den; bp = P;
            continue;
        }
        /* M(d) > M(best)  <=>  num * bden > bnum * den */
        if (num * bden > bnum * den) { best = d; bnum = num; bden = den; bp = P; }
    }
    if (p_hat) *p_hat = bp;
    if (metric) *metric = best >= 0 ? bnum / bden : 0.0;
    if (f_delta) *f_delta = best >= 0 ? atan2(bp.im, bp.re) / (double)L : 0.0;
    return (int)best;
}
void orc_sc_metric(const oc64 *r, size_t n, int L, int window_reps, long n_lags, double *metric, oc64 *p) {
    int W = window_reps * L;
    long valid = (long)n - (long)W - (long)L + 1;
    if (n_lags <= 0 || n_lags > valid) n_lags = valid;
    for (long d = 0; d < n_lags; d++) {
        oc64 P; double E, R;
        sc_at(r, d, L, W, &P, &E, &R);
        double den = E * R;
        metric[d] = den > 0.0 ? c_norm_sqr(P) / den : 0.0;
        if (p) p[d] = P;
    }
}

/* ------------------------------------------------------------------ EXT-2 Hamming(7,4) */
static inline unsigned ham_enc_nibble(unsigned d) {
    unsigned d0 = d & 1, d1 = (d >> 1) & 1, d2 = (d >> 2) & 1, d3 = (d >> 3) & 1;
    unsigned p0 = d0 ^ d1 ^ d3, p1 = d0 ^ d2 ^ d3, p2 = d1 ^ d2 ^ d3;
    return (d & 0xF) | (p0 << 4) | (p1 << 5) | (p2 << 6);
}
static inline unsigned ham_dec_word(unsigned c, unsigned *fixed) {
    unsigned b0 = c & 1, b1 = (c >> 1) & 1, b2 = (c >> 2) & 1, b3 = (c >> 3) & 1;
    unsigned s0 = ((c >> 4) & 1) ^ b0 ^ b1 ^ b3, s1 = ((c >> 5) & 1) ^ b0 ^ b2 ^ b3, s2 = ((c >> 6) & 1) ^ b1 ^ b2 ^ b3;
    unsigned syn = s0 | (s1 << 1) | (s2 << 2);
    static const int flip[8] = {-1, 4, 5, 0, 6, 1, 2, 3}; /* syndrome -> bit position */
    if (syn) { c ^= 1u << flip[syn]; if (fixed) (*fixed)++; }
    return c & 0xF;
}
size_t orc_hamming74_encoded_len(size_t nbytes) { return ((nbytes + 3) / 4) * 7; }
size_t orc_hamming74_encode(const uint8_t *data, size_t nbytes, uint8_t *out) {
    size_t nblk = (nbytes + 3) / 4;
    for (size_t b = 0; b < nblk; b++) {
        uint64_t acc = 0;
        for (int i = 0; i < 8; i++) {
            size_t by = b * 4 + (size_t)(i >> 1);
            unsigned byte = by < nbytes ? data[by] : 0u;
            unsigned nib = (i & 1) ? (byte >> 4) : (byte & 0xF);
            acc |= (uint64_t)ham_enc_nibble(nib) << (7 * i);
        }
        for (int i = 0; i < 7; i++) out[b * 7 + (size_t)i] = (uint8_t)(acc >> (8 * i));
    }
    return nblk * 7;
}
size_t orc_hamming74_decode(const uint8_t *code, size_t nbytes, uint8_t *out, uint32_t *corrected) {
    size_t nblk = nbytes / 7; /* a trailing partial block is dropped */
    unsigned fixed = 0;
    for (size_t b = 0; b < nblk; b++) {
        uint64_t acc = 0;
        for (int i = 0; i < 7; i++) acc |= (uint64_t)code[b * 7 + (size_t)i] << (8 * i);
        for (int i = 0; i < 4; i++) {
            unsigned lo = ham_dec_word((unsigned)((acc >> (14 * i)) & 0x7F), &fixed);
            unsigned hi = ham_dec_word((unsigned)((acc >> (14 * i + 7)) & 0x7F), &fixed);
            out[b * 4 + (size_t)i] = (uint8_t)(lo | (hi << 4));
        }
    }
    if (corrected) *corrected = fixed;
    return nblk * 4;
}

/* ------------------------------------------------------------------ RX chains */
static orc_rx_info rx_chain(const oc64 *samples, size_t n, long offset, double f_delta, int n_fft, int cp,
                            int guard, int modulation, const oc64 *training, int max_symbols, uint8_t *out,
                            size_t out_cap, oc64 *soft, size_t soft_cap) {
    orc_rx_info info;
    memset(&info, 0, sizeof(info));
    info.offset = offset;
    info.f_delta = f_delta;
    int S = n_fft + cp;
    if (offset < 0 || (size_t)offset > n) { info.status = -3; return info; } /* split_off panics */
    size_t len = n - (size_t)offset;
    if (len < (size_t)(10 * S)) { info.status = -1; return info; } /* receiver.rs:27-29 */
    /* split_into_chunks + pad_chunk (receiver.rs:192-210) */
    size_t nchunks = (len + (size_t)S - 1) / (size_t)S;
    size_t ndata_chunks = nchunks - 10;
    if (max_symbols > 0 && ndata_chunks > (size_t)max_symbols) ndata_chunks = (size_t)max_symbols;
    size_t used_chunks = 10 + ndata_chunks;
    oc64 *buf = (oc64 *)calloc(used_chunks * (size_t)S, sizeof(oc64));
    size_t ncopy = len < used_chunks * (size_t)S ? len : used_chunks * (size_t)S;
    memcpy(buf, samples + offset, sizeof(oc64) * ncopy);
    /* CFO derotation over every sample, sample_id from the trimmed start (receiver.rs:44-50) */
    orc_cfo_rotate(buf, used_chunks * (size_t)S, f_delta, 0);
    oc64 *hk = (oc64 *)malloc(sizeof(oc64) * (size_t)n_fft);
    orc_estimate_channel(buf + 5 * (size_t)S, n_fft, cp, training, hk);
    size_t nd = (size_t)orc_data_carriers(n_fft, guard);
    oc64 *stream = (oc64 *)malloc(sizeof(oc64) * (ndata_chunks * nd + 1));
    oc64 *blk = (oc64 *)malloc(sizeof(oc64) * (size_t)n_fft);
    size_t ns = 0;
    for (size_t c = 0; c < ndata_chunks; c++) {
        orc_unprefix_block(buf + (10 + c) * (size_t)S, n_fft, cp, blk);
        for (int i = 0; i < n_fft; i++) blk[i] = c_div(blk[i], hk[i]); /* receiver.rs:68-70 */
        ns += orc_decode_block(blk, n_fft, guard, stream + ns);
    }
    if (soft) memcpy(soft, stream, sizeof(oc64) * (ns < soft_cap ? ns : soft_cap));
    info.n_symbols = ndata_chunks;
    uint8_t *dec = (uint8_t *)malloc(ns * (size_t)modulation / 8 + 16);
    size_t nb = orc_demodulate(stream, ns, modulation, dec);
    if (nb < 16) { info.status = -4; } /* drain(0..16) panics */
    else {
        /* header parse + truncate (receiver.rs:86-95); u128 LE, only the low 64 bits can be < usize::MAX */
        uint64_t lo = 0, hi = 0;
        for (int i = 0; i < 8; i++) { lo |= (uint64_t)dec[i] << (8 * i); hi |= (uint64_t)dec[8 + i] << (8 * i); }
        size_t body = nb - 16;
        size_t keep = (hi == 0 && lo < (uint64_t)body) ? (size_t)lo : body;
        info.n_bytes = keep;
        memcpy(out, dec + 16, keep < out_cap ? keep : out_cap);
    }
    free(dec); free(blk); free(stream); free(hk); free(buf);
    return info;
}

orc_rx_info orc_decode_given(const oc64 *samples, size_t n, long offset, double f_delta, int n_fft, int cp,
                             int guard, int modulation, const oc64 *training, int max_symbols, uint8_t *out,
                             size_t out_cap, oc64 *soft, size_t soft_cap) {
    return rx_chain(samples, n, offset, f_delta, n_fft, cp, guard, modulation, training, max_symbols, out, out_cap,
                    soft, soft_cap);
}

/* receiver.rs:9-96 */
orc_rx_info orc_decode_ref(const oc64 *samples, size_t n, int n_fft, int cp, int guard, int modulation,
                           const oc64 *training, uint8_t *out, size_t out_cap, oc64 *soft, size_t soft_cap) {
    int S = n_fft + cp;
    oc64 *lock = (oc64 *)malloc(sizeof(oc64) * (size_t)S);
    orc_locking_signal(S, lock);
    oc64 *cross = (oc64 *)malloc(sizeof(oc64) * (2 * n - 1));
    int idxmax = orc_xcorr_fft(samples, (int)n, lock, S, cross);
    double peak = sqrt(c_norm_sqr(cross[idxmax]));
    long offset = (long)idxmax - ((long)((2 * n - 1 - 1) / 2) + 1); /* receiver.rs:21 */
    free(cross); free(lock);
    orc_rx_info info;
    memset(&info, 0, sizeof(info));
    info.offset = offset;
    info.metric = peak;
    if (offset < 0 || (size_t)offset > n) { info.status = -3; return info; }
    if (n - (size_t)offset < (size_t)(10 * S)) { info.status = -1; return info; }
    /* frequency_correction(&chunks[3], &chunks[4]) (receiver.rs:39) */
    double fd = orc_frequency_correction(samples + offset + 3 * S, samples + offset + 4 * S, S);
    info = rx_chain(samples, n, offset, fd, n_fft, cp, guard, modulation, training, 0, out, out_cap, soft, soft_cap);
    info.metric = peak;
    return info;
}

orc_rx_info orc_decode_sc(const oc64 *samples, size_t n, int n_fft, int cp, int guard, int modulation,
                          const oc64 *training, int window_reps, long sync_lags, double threshold, int backoff,
                          int cfo_abs, int max_symbols, uint8_t *out, size_t out_cap, oc64 *soft, size_t soft_cap) {
    int S = n_fft + cp;
    orc_rx_info info;
    memset(&info, 0, sizeof(info));
    oc64 P; double metric, fd;
    int d = orc_sc_sync(samples, n, S, window_reps, sync_lags, threshold, &P, &metric, &fd);
    if (d < 0) { info.status = -2; return info; }
    long offset = (long)d - (long)S - (long)backoff;
    if (offset < 0) offset = 0;
    if (cfo_abs == 1) fd = fabs(fd);   /* OFDM_CFO_ABS: the reference's abs() (receiver.rs:239) */
    else if (cfo_abs == 2) fd = 0.0;   /* OFDM_CFO_OFF: no derotation */
    info = rx_chain(samples, n, offset, fd, n_fft, cp, guard, modulation, training, max_symbols, out, out_cap, soft,
                    soft_cap);
    info.metric = metric;
    return info;
}

size_t orc_rx_demod(const oc64 *samples, size_t n_symbols, int n_fft, int cp, int guard, int modulation,
                    const oc64 *hk, uint8_t *out, oc64 *soft) {
    int S = n_fft + cp;
    size_t nd = (size_t)orc_data_carriers(n_fft, guard);
    oc64 *stream = (oc64 *)malloc(sizeof(oc64) * (n_symbols * nd + 1));
    oc64 *blk = (oc64 *)malloc(sizeof(oc64) * (size_t)n_fft);
    size_t ns = 0;
    for (size_t s = 0; s < n_symbols; s++) {
        orc_unprefix_block(samples + s * (size_t)S, n_fft, cp, blk);
        if (hk) for (int i = 0; i < n_fft; i++) blk[i] = c_div(blk[i], hk[i]);
        ns += orc_decode_block(blk, n_fft, guard, stream + ns);
    }
    if (soft) memcpy(soft, stream, sizeof(oc64) * ns);
    size_t nb = orc_demodulate(stream, ns, modulation, out);
    free(blk); free(stream);
    return nb;
}

/* ------------------------------------------------------------------ channel.rs */
const double ORC_CHANNEL[64] = { /* channel.rs:26-31 */
    0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -0.0000, -0.1912, 0.9316, 0.2821, -0.1990, 0.1630, -0.1017, 0.0544, -0.0261,
    0.0090, 0.0000, -0.0034, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0,
    0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0,
    0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

/* channel.rs:33-74.  rand::thread_rng replaced by SplitMix64(seed): one U(0,1) draw for f_delta when
 * timing_error, then (re, im) U(-1,1) draws per output sample. */
size_t orc_channel(const oc64 *tx, size_t n, double snr_db, int timing_error, uint64_t seed, oc64 *out,
                   double *f_delta_used) {
    uint64_t st = seed;
    double snr = pow(10.0, snr_db / 10.0);
    oc64 h[64];
    for (int i = 0; i < 64; i++) h[i] = c_new(ORC_CHANNEL[i], 0.0);
    size_t m = n + 63;
    orc_convolve(tx, (int)n, h, 64, out);
    double fd = 0.0;
    if (timing_error) {
        fd = M_PI * (orc_uniform_01(&st) / 80.0); /* channel.rs:54 */
        for (size_t i = 0; i < m; i++) {
            oc64 comp = c_scale(c_scale(c_new(0.0, 1.0), fd), (double)(i + 1));
            out[i] = c_mul(out[i], c_exp(comp));
        }
    }
    if (f_delta_used) *f_delta_used = fd;
    oc64 var = orc_variance(out, (int)m);
    oc64 nv = c_new(var.re / snr, var.im / snr);
    double complex sq = csqrt((0.5 * nv.re) + (0.5 * nv.im) * I); /* (0.5 * noise_var).sqrt() */
    oc64 scale = c_new(creal(sq), cimag(sq));
    for (size_t i = 0; i < m; i++) {
        double re = orc_uniform_pm1(&st), im = orc_uniform_pm1(&st);
        out[i] = c_add(out[i], c_mul(scale, c_new(re, im)));
    }
    return m;
}
