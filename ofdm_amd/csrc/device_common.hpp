// device_common.hpp -- shared device-side building blocks for the gfx950 OFDM kernels.
// Written for CDNA4 only: 64-lane wavefronts, LDS-staged Stockham passes, no MFMA (none of these stages is
// a dense contraction; the kernels are HBM / VALU bound).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ofdm {

typedef float2 cf;

__device__ __forceinline__ cf cf_make(float x, float y) { return make_float2(x, y); }
__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf cmul(cf a, cf b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cf cmulc(cf a, cf b) { // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ cf cscale(cf a, float s) { return make_float2(a.x * s, a.y * s); }
// multiply by -j (forward) or +j (inverse)
template <bool INV> __device__ __forceinline__ cf mul_mj(cf a) {
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
template <bool INV> __device__ __forceinline__ cf twid(cf w) { return INV ? make_float2(w.x, -w.y) : w; }

// ---- carrier map (reference: src/transmitter.rs:150-161, src/receiver.rs:122-133), tiled k = N/64
// class of reference bin c (0..63): 0 data, 1 null, 2 pilot
__device__ __forceinline__ int carrier_class64(int c, int guard) {
    if (!guard) return 0;
    if (c >= 59 || c <= 5 || c == 32) return 1;
    if (c == 6 || c == 25 || c == 39 || c == 58) return 2;
    return 0;
}
// number of DATA classes below reference bin c (guard on): data classes are 7..58 minus {25,32,39,58}
__device__ __forceinline__ int data_classes_below64(int c) {
    if (c <= 7) return 0;
    int n = c - 7;            // classes 7..c-1
    n -= (c > 25) + (c > 32) + (c > 39) + (c > 58);
    return n > 48 ? 48 : n;
}

// ---- radix-8 / radix-4 butterflies (decimation in frequency, natural-order outputs)
template <bool INV> __device__ __forceinline__ void bfly8(cf *v) {
    const float h = 0.70710678118654752440f;
    cf a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
    cf a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    cf a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
    cf a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
    // a5 *= W8^1, a6 *= W8^2, a7 *= W8^3   (W8 = exp(-+ j pi/4))
    a5 = INV ? make_float2((a5.x - a5.y) * h, (a5.x + a5.y) * h) : make_float2((a5.x + a5.y) * h, (a5.y - a5.x) * h);
    a6 = mul_mj<INV>(a6);
    a7 = INV ? make_float2((-a7.x - a7.y) * h, (a7.x - a7.y) * h) : make_float2((a7.y - a7.x) * h, (-a7.x - a7.y) * h);
    cf b0 = cadd(a0, a2), b2 = csub(a0, a2);
    cf b1 = cadd(a1, a3), b3 = mul_mj<INV>(csub(a1, a3));
    cf b4 = cadd(a4, a6), b6 = csub(a4, a6);
    cf b5 = cadd(a5, a7), b7 = mul_mj<INV>(csub(a5, a7));
    v[0] = cadd(b0, b1); v[4] = csub(b0, b1);
    v[2] = cadd(b2, b3); v[6] = csub(b2, b3);
    v[1] = cadd(b4, b5); v[5] = csub(b4, b5);
    v[3] = cadd(b6, b7); v[7] = csub(b6, b7);
}
// radix-4 on v[o], v[o+2], v[o+4], v[o+6]
template <bool INV> __device__ __forceinline__ void bfly4(cf *v, int o) {
    cf s0 = cadd(v[o], v[o + 4]), d0 = csub(v[o], v[o + 4]);
    cf s1 = cadd(v[o + 2], v[o + 6]), d1 = mul_mj<INV>(csub(v[o + 2], v[o + 6]));
    v[o] = cadd(s0, s1); v[o + 4] = csub(s0, s1);
    v[o + 2] = cadd(d0, d1); v[o + 6] = csub(d0, d1);
}

// ---- Stockham plan: log2(N) = 3a + 2b, radix-8 passes first then radix-4 (b in {0,1,2})
template <int N> struct Plan {
    static constexpr int LOG2 = (N == 64) ? 6 : (N == 128) ? 7 : (N == 256) ? 8 : (N == 512) ? 9
                               : (N == 1024) ? 10 : (N == 2048) ? 11 : 12;
    static constexpr int B4 = (LOG2 % 3 == 0) ? 0 : (LOG2 % 3 == 2) ? 1 : 2; // number of radix-4 passes
    static constexpr int A8 = (LOG2 - 2 * B4) / 3;                           // number of radix-8 passes
    static constexpr int PASSES = A8 + B4;
    static constexpr int T = N / 8;                 // threads per OFDM symbol, 8 points each
    static constexpr int WG = (T > 256) ? T : 256;  // workgroup size
    static constexpr int G = WG / T;                // symbols per workgroup iteration
    static constexpr int LDS_SYM = N + 8;           // per-symbol LDS stride (pad 8 points: conflict-free reads)
    static constexpr bool WAVE_LOCAL = (T <= 64);   // a symbol's threads live in one wavefront: no barriers
    // twiddle registers: ONE base twiddle per radix-8 pass after the first, two per radix-4 pass (one per butterfly);
    // the powers w^2 .. w^7 are formed by successive multiplication (<= 7 roundings, ~4e-7 relative)
    static constexpr int NTW = (A8 - 1) + B4 * 2;
    __host__ __device__ static constexpr int radix(int p) { return p < A8 ? 8 : 4; }
};

// XOR swizzle inside aligned 8-point groups: turns the stride-8 writes of the first pass into conflict-free
// ds_write_b64 groups while keeping unit-stride reads conflict-free.
__device__ __forceinline__ int swz(int i) { return i ^ ((i >> 3) & 7); }

template <int T> __device__ __forceinline__ void group_sync() {
    if (T > 64) __syncthreads();
    else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// Load the per-thread twiddles for all passes after the first (loop invariant for a persistent thread).
// tw[m] = exp(-2 pi i m / N), m < N (float2 table in global memory, produced on the host in f64).
template <int N> __device__ __forceinline__ void load_twiddles(const cf *__restrict__ tw, int t, cf *w) {
    typedef Plan<N> P;
    int ns = 8, wi = 0;
#pragma unroll
    for (int p = 1; p < P::PASSES; ++p) {
        if (P::radix(p) == 8) {
            w[wi++] = tw[(t % ns) * (N / (ns * 8))];
            ns *= 8;
        } else {
            const int step = N / (ns * 4);
            w[wi++] = tw[(t % ns) * step];
            w[wi++] = tw[((t + P::T) % ns) * step];
            ns *= 4;
        }
    }
}

// In-register/LDS FFT of one symbol spread over T threads: on entry v[m] = x[t + m*T], on exit v[m] = X[t + m*T].
// buf: this symbol's LDS slab (LDS_SYM points).  Unnormalised in both directions.
template <int N, bool INV> __device__ __forceinline__ void fft_symbol(cf *v, cf *buf, int t, const cf *w) {
    typedef Plan<N> P;
    int ns = 1, wi = 0;
#pragma unroll
    for (int p = 0; p < P::PASSES; ++p) {
        if (p > 0) {
            // read this pass's inputs (unit stride across the symbol's threads)
            group_sync<P::T>();
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = buf[swz(t + m * P::T)];
            group_sync<P::T>();
        }
        if (P::radix(p) == 8) {
            if (p > 0) {
                const cf w1 = twid<INV>(w[wi]);
                cf wr = w1;
#pragma unroll
                for (int r = 1; r < 8; ++r) { v[r] = cmul(v[r], wr); if (r < 7) wr = cmul(wr, w1); }
                wi += 1;
            }
            bfly8<INV>(v);
            if (p < P::PASSES - 1) {
                int k = t % ns, base = (t / ns) * ns * 8 + k;
#pragma unroll
                for (int r = 0; r < 8; ++r) buf[swz(base + r * ns)] = v[r];
            }
            ns *= 8;
        } else {
            {
                const cf wa1 = twid<INV>(w[wi]), wb1 = twid<INV>(w[wi + 1]);
                cf wa = wa1, wb = wb1;
#pragma unroll
                for (int r = 1; r < 4; ++r) {
                    v[2 * r] = cmul(v[2 * r], wa);
                    v[2 * r + 1] = cmul(v[2 * r + 1], wb);
                    if (r < 3) { wa = cmul(wa, wa1); wb = cmul(wb, wb1); }
                }
            }
            wi += 2;
            bfly4<INV>(v, 0);
            bfly4<INV>(v, 1);
            if (p < P::PASSES - 1) {
                int ja = t, jb = t + P::T;
                int basea = (ja / ns) * ns * 4 + ja % ns, baseb = (jb / ns) * ns * 4 + jb % ns;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    buf[swz(basea + r * ns)] = v[2 * r];
                    buf[swz(baseb + r * ns)] = v[2 * r + 1];
                }
            }
            ns *= 4;
        }
    }
}

// ---- CFO phasor exp(-j f_delta n) with the phase reduced in f64 (receiver.rs:44-50 runs in f64; an f32
// product f_delta*n would lose ~1e-5 rad by n ~ 1e3).  turns = f_delta / (2 pi).
__device__ __forceinline__ cf cfo_phasor(double turns, long long n) {
    double ph = turns * (double)n;
    ph -= rint(ph);                 // [-0.5, 0.5] turns, exact
    // the hardware sine / cosine take TURNS: 2 instructions, max abs error 1.3e-7 over the whole range on gfx950
    // (tools/lab/trig_probe.cpp; sincospif: 5e-8 for ~35 instructions)
    const float t = (float)ph;
    return make_float2(__builtin_amdgcn_cosf(t), -__builtin_amdgcn_sinf(t));
}

// ---- hard decisions (src/receiver.rs:147-190 for BPSK/QPSK; DESIGN.md 3.1 for 16/64/256-QAM)
// level index l (0 .. M - 1) of one axis -> its m stream bits: Gray code of l with the first stream bit (the Gray MSB) at bit 0
__device__ __forceinline__ unsigned axis_code(unsigned l, int m) {
    const int M = 1 << m;
    if (m <= 3) {
        // Gray code + bit reversal (first stream bit at bit 0) from a compile-time table, m bits per entry: 2 instructions
        // (shift-add, bit-field extract) instead of shift / xor / brev / shift
        unsigned lut = 0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const unsigned g = (unsigned)i ^ ((unsigned)i >> 1);
            unsigned r = 0;
#pragma unroll
            for (int b = 0; b < m; ++b) r |= ((g >> b) & 1u) << (m - 1 - b);
            lut |= r << (m * i);
        }
        return __builtin_amdgcn_ubfe(lut, l * (unsigned)m, (unsigned)m);
    }
    if (m == 4) { // 16 entries x 4 bits: two 32-bit halves of the same table
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned g = (unsigned)i ^ ((unsigned)i >> 1);
            const unsigned r = ((g & 1u) << 3) | ((g & 2u) << 1) | ((g & 4u) >> 1) | ((g & 8u) >> 3);
            if (i < 8) lo |= r << (4 * i); else hi |= r << (4 * (i - 8));
        }
        return __builtin_amdgcn_ubfe(l < 8u ? lo : hi, (l & 7u) * 4u, 4u);
    }
    const unsigned g = l ^ (l >> 1);           // Gray code, MSB = first stream bit
    return __brev(g) >> (32 - m);              // first stream bit at bit 0
}
// returns the bps-bit index: bit j = j-th bit of the point in stream order
// clamp(f, 0, top) as ONE v_med3_f32 (v_max_f32 and v_min_f32 each cost what it costs: 4 cycles per wavefront on gfx950, twice a
// v_add_f32 -- tools/lab/valu_rate.hip); a NaN gives min3 of the operands = 0, as fminf(fmaxf(NaN, 0), top) did
__device__ __forceinline__ float clamp0(float f, float top) { return __builtin_amdgcn_fmed3f(f, 0.0f, top); }
__device__ __forceinline__ unsigned axis_bits(float x, int m) {
    const int M = 1 << m;
    // l = clamp(floor(x (M-1) / 2) + M/2, 0, M-1): one fma, clamp, truncating convert (the clamped value is non-negative, so
    // truncation is the floor; NaN -> 0)
    const float f = clamp0(fmaf(x, 0.5f * (float)(M - 1), (float)(M / 2)), (float)(M - 1));
    return axis_code((unsigned)f, m);
}
__device__ __forceinline__ unsigned demap_point(cf z, int bps) {
    if (bps == 1) return z.x > 0.0f ? 1u : 0u;                       // receiver.rs:162
    if (bps == 2) {                                                   // receiver.rs:169-175, arms in order
        float re = z.x, im = z.y;
        if (re >= 0.0f && im >= 0.0f) return 3u;
        if (re >= 0.0f && im <= 0.0f) return 1u;
        if (re < 0.0f && im > 0.0f) return 2u;
        return 0u;
    }
    const int m = bps >> 1;
    if (bps == 8) {
        // 256-QAM: both axes' levels side by side (Q below I), Gray-coded together (the mask keeps the shift inside each nibble),
        // and ONE 32-bit bit reversal turns the byte round: rev4(gray(l_I)) lands in the low nibble, rev4(gray(l_Q)) in the high
        // one -- 12 instead of 17 instructions per point against the two table look-ups of axis_bits (two 32-bit halves each)
        const float fi = clamp0(fmaf(z.x, 7.5f, 8.0f), 15.0f), fq = clamp0(fmaf(z.y, 7.5f, 8.0f), 15.0f);
        const unsigned x = (unsigned)fq | ((unsigned)fi << 4);
        const unsigned g = x ^ ((x >> 1) & 0x77u);
        return __brev(g) >> 24;
    }
    return axis_bits(z.x, m) | (axis_bits(z.y, m) << m);
}
// demap_point(z * rot): the pilot-phase rotation (src/receiver.rs:106-145) folded into the demapper's scale-and-offset -- per axis
// two fused multiply-adds on (rot * scale) instead of a complex product and one more; the soft value differs from the two-step
// form by an ulp, far inside the 1e-5 band in which the parity rule excuses a decision.  BPSK / QPSK keep the product.
__device__ __forceinline__ unsigned demap_point_rot(cf z, cf rot, int bps) {
    if (bps <= 2) return demap_point(make_float2(z.x * rot.x - z.y * rot.y, z.x * rot.y + z.y * rot.x), bps);
    const int m = bps >> 1, M = 1 << m;
    const float sc = 0.5f * (float)(M - 1), half = (float)(M / 2), top = (float)(M - 1);
    const float rx = rot.x * sc, ry = rot.y * sc;       // common to the eight points of a lane: computed once
    const float fi = clamp0(fmaf(z.x, rx, fmaf(-z.y, ry, half)), top);
    const float fq = clamp0(fmaf(z.x, ry, fmaf(z.y, rx, half)), top);
    if (bps == 8) {
        const unsigned x = (unsigned)fq | ((unsigned)fi << 4);
        const unsigned g = x ^ ((x >> 1) & 0x77u);
        return __brev(g) >> 24;
    }
    return axis_code((unsigned)fi, m) | (axis_code((unsigned)fq, m) << m);
}
// map a bps-bit index to a constellation point (src/transmitter.rs:108-140; levels in [-1,1])
__device__ __forceinline__ float axis_level(unsigned bits, int m) {
    unsigned g = __brev(bits) >> (32 - m); // first stream bit -> Gray MSB
    unsigned l = g;
    l ^= l >> 1; l ^= l >> 2;              // Gray decode (m <= 4)
    const int M = 1 << m;
    // (2l - (M-1)) / (M-1), correctly rounded without an IEEE divide: one Newton step on x * RN(1/d) (checked
    // exhaustively for d = 1, 3, 7, 15 and every numerator: bit-identical to the division)
    const float d = (float)(M - 1);
    const float r = m == 1 ? 1.0f : m == 2 ? (1.0f / 3.0f) : m == 3 ? (1.0f / 7.0f) : (1.0f / 15.0f);
    const float x = (float)(2 * (int)l - (M - 1));
    const float q0 = x * r;
    return fmaf(fmaf(-q0, d, x), r, q0);
}
__device__ __forceinline__ cf map_point(unsigned idx, int bps) {
    if (bps == 1) return make_float2((idx & 1u) ? 1.0f : -1.0f, 0.0f);
    const int m = bps >> 1;
    return make_float2(axis_level(idx & ((1u << m) - 1u), m), axis_level((idx >> m) & ((1u << m) - 1u), m));
}
// One constellation point of a TX symbol (src/transmitter.rs:108-165).  boff = bit offset of the bin's field inside the symbol's
// byte window sbw (dwords, one slack dword behind it), -1 = null carrier, -2 = pilot; fields at or beyond live_bits carry no
// stream bits (zero point, transmitter.rs:158-160); ptab[idx] = map_point(idx, bps) tabulated in LDS (bit-identical to the
// staged modulate).  Two LDS instructions per point -- the two dwords that hold the field (one ds_read2_b32) and the point
// (one ds_read_b64) -- instead of two byte reads and two axis-level reads: the mapping's LDS reads bound the TX kernels.
// BRANCHFREE: dead lanes read dword 0 and discard it instead of branching around the reads (pays for k_tx4096, costs 30 % in
// the R x 64 kernels, measured).
template <bool BRANCHFREE>
__device__ __forceinline__ cf tx_point(const unsigned *sbw, const cf *ptab, int boff, int live_bits, unsigned mask) {
    const bool live = boff >= 0 && boff < live_bits;
    const float px = boff == -2 ? 1.0f : 0.0f;
    if (BRANCHFREE) {
        const int bit = live ? boff : 0;
        const unsigned lo = sbw[bit >> 5], hi = sbw[(bit >> 5) + 1];
        const cf pt = ptab[__builtin_amdgcn_alignbit(hi, lo, (unsigned)bit & 31u) & mask];
        return make_float2(live ? pt.x : px, live ? pt.y : 0.0f);
    }
    cf pt = make_float2(px, 0.0f);
    if (live) {
        const unsigned lo = sbw[boff >> 5], hi = sbw[(boff >> 5) + 1];
        pt = ptab[__builtin_amdgcn_alignbit(hi, lo, (unsigned)boff & 31u) & mask];
    }
    return pt;
}
// bps bits starting at bit `bit` of an LSB-first byte stream made of a 16-byte little-endian length header
// followed by payload[0..len)  (src/packets/mod.rs:20-32, src/transmitter.rs:37-47); bits past the end are 0
__device__ __forceinline__ unsigned stream_byte(const uint8_t *__restrict__ payload, long long len, long long by) {
    if (by < 16) return by < 8 ? (unsigned)((unsigned long long)len >> (8 * by)) & 0xFFu : 0u;
    by -= 16;
    return by < len ? (unsigned)payload[by] : 0u;
}
__device__ __forceinline__ unsigned raw_bits(const uint8_t *__restrict__ bytes, long long n_bytes, long long bit, int bps) {
    long long by = bit >> 3;
    unsigned lo = by < n_bytes ? bytes[by] : 0u, hi = (by + 1) < n_bytes ? bytes[by + 1] : 0u;
    return ((lo | (hi << 8)) >> (bit & 7)) & ((1u << bps) - 1u);
}

// A payload row's own length: callers' values outside [0, payload_bytes] are clamped (a length above the row would read the next row --
// or, behind the last row, unmapped memory -- and claim bytes the frame has no symbols for)
__device__ __forceinline__ long long row_len(long long v, int payload_bytes) {
    return v < 0 ? 0 : (v > payload_bytes ? (long long)payload_bytes : v);
}

// ---- Hamming(7,4): codeword bits [d0 d1 d2 d3 p0 p1 p2], LSB first
__device__ __forceinline__ unsigned ham_enc(unsigned d) {
    unsigned d0 = d & 1u, d1 = (d >> 1) & 1u, d2 = (d >> 2) & 1u, d3 = (d >> 3) & 1u;
    return (d & 0xFu) | ((d0 ^ d1 ^ d3) << 4) | ((d0 ^ d2 ^ d3) << 5) | ((d1 ^ d2 ^ d3) << 6);
}
__device__ __forceinline__ unsigned ham_dec(unsigned c, unsigned &fixed) {
    unsigned b0 = c & 1u, b1 = (c >> 1) & 1u, b2 = (c >> 2) & 1u, b3 = (c >> 3) & 1u;
    unsigned s0 = ((c >> 4) & 1u) ^ b0 ^ b1 ^ b3, s1 = ((c >> 5) & 1u) ^ b0 ^ b2 ^ b3, s2 = ((c >> 6) & 1u) ^ b1 ^ b2 ^ b3;
    unsigned syn = s0 | (s1 << 1) | (s2 << 2);
    // syndrome -> flipped bit position: 1:p0(4) 2:p1(5) 3:d0 4:p2(6) 5:d1 6:d2 7:d3, packed 3 bits each
    const unsigned flip = (4u << 3) | (5u << 6) | (0u << 9) | (6u << 12) | (1u << 15) | (2u << 18) | (3u << 21);
    if (syn) { c ^= 1u << ((flip >> (3 * syn)) & 7u); fixed++; }
    return c & 0xFu;
}

template <int CTRL> __device__ __forceinline__ float dpp8_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum8_lanes(float x) { // sum over the 8 lanes of a symbol group (lanes 8s .. 8s+7)
    x += dpp8_f<0xB1>(x);  // quad_perm [1,0,3,2]
    x += dpp8_f<0x4E>(x);  // quad_perm [2,3,0,1]
    x += dpp8_f<0x141>(x); // row_half_mirror
    return x;
}

// ---- FFT64 with ONE point per lane (decimation in frequency, 6 radix-2 stages across the wavefront): lane n enters with
// x[n] and leaves with X[bitrev6(n)].  ~12 VALU per stage instead of a whole 8-symbol group iteration, which is what
// makes the per-frame channel estimate cheap (one transform per frame).  tws[st] = this lane's twiddle of stage st
// (1 for the lanes that take the sum).
template <int CTRL> __device__ __forceinline__ cf dpp_cf(cf v) { return make_float2(dpp8_f<CTRL>(v.x), dpp8_f<CTRL>(v.y)); }
__device__ __forceinline__ cf lane_fft64(cf x, int lane, const cf *tws) {
#pragma unroll
    for (int st = 0; st < 6; ++st) {
        const int h = 32 >> st;
        cf xp;
        if (h == 32) xp = make_float2(__shfl_xor(x.x, 32, 64), __shfl_xor(x.y, 32, 64));
        else if (h == 16) xp = make_float2(__shfl_xor(x.x, 16, 64), __shfl_xor(x.y, 16, 64));
        else if (h == 8) xp = dpp_cf<0x128>(x);                 // row_ror:8   : lane ^ 8
        else if (h == 4) xp = dpp_cf<0x1B>(dpp_cf<0x141>(x));   // half mirror (^7) then quad reverse (^3) : lane ^ 4
        else if (h == 2) xp = dpp_cf<0x4E>(x);                  // quad_perm [2,3,0,1] : lane ^ 2
        else xp = dpp_cf<0xB1>(x);                              // quad_perm [1,0,3,2] : lane ^ 1
        const float sg = (lane & h) ? -1.f : 1.f;              // lower half: a + b; upper half: (a - b) w
        const cf y = make_float2(fmaf(sg, x.x, xp.x), fmaf(sg, x.y, xp.y));
        x = cmul(y, tws[st]);
    }
    return x;
}
__device__ __forceinline__ int bitrev6(int v) { return (int)(__brev((unsigned)v) >> 26); }

// one 7-byte block (8 codewords, LSB first) -> 4 data bytes
__device__ __forceinline__ void ham_decode_block(const uint8_t *src, uint8_t *dst, unsigned &fixed) {
    unsigned long long acc = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) acc |= (unsigned long long)src[j] << (8 * j);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned lo = ham_dec((unsigned)(acc >> (14 * j)) & 0x7Fu, fixed);
        unsigned hi = ham_dec((unsigned)(acc >> (14 * j + 7)) & 0x7Fu, fixed);
        dst[j] = (uint8_t)(lo | (hi << 4));
    }
}

// One 8-byte load of a complex sample through a pointer that came out of a select: as a struct of two floats the compiler
// splits such a load into two global_load_dword (it no longer sees that the halves are adjacent).
__device__ __forceinline__ cf ld_cf(const cf *a) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    const f2v r = *reinterpret_cast<const f2v *>(a);
    return make_float2(r.x, r.y);
}

// ---- payload dwords fetched ahead of their use
// A global load under a branch (`if (in_range) v = p[i];`, or `in_range ? p[i] : 0`) is compiled as branch + load +
// s_waitcnt vmcnt(0) at the join: the "prefetch" is synchronous (found in the round-3 ISA scan; k_sc_stream's evaluations
// ran 1.7x faster without it).  So the load is issued UNCONDITIONALLY -- from `safe`, any mapped 4-byte aligned address,
// when bytes by .. by + 3 are not wanted, not wholly inside [0, len) or not 4-byte aligned -- and the verdict is taken
// where the value is used (paydw_settle), which rebuilds the ragged dword byte by byte.
__device__ __forceinline__ bool paydw_whole(long long by, long long len, bool want, bool aligned) {
    return want && aligned && by >= 0 && by + 4 <= len;
}
__device__ __forceinline__ unsigned paydw_issue(const uint8_t *base, long long by, long long len, bool want, bool aligned, const void *safe) {
    const unsigned *src = paydw_whole(by, len, want, aligned) ? reinterpret_cast<const unsigned *>(base + by)
                                                              : reinterpret_cast<const unsigned *>(safe);
    return *src;
}
__device__ __forceinline__ unsigned paydw_settle(unsigned raw, const uint8_t *base, long long by, long long len, bool want, bool aligned) {
    if (paydw_whole(by, len, want, aligned)) return raw;
    unsigned v = 0;
    if (want) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (by + j >= 0 && by + j < len) v |= (unsigned)base[by + j] << (8 * j);
    }
    return v;
}

// ---- LDS-DMA (global_load_lds) staging helpers
// Workgroup barrier that does NOT drain the VM counter: the next tile's LDS-DMA stays in flight across it.
// (__syncthreads() would emit s_waitcnt vmcnt(0) while a global_load_lds is outstanding.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One 1-KiB LDS-DMA piece: lane i copies 16 B from sbase + voff (per lane) to lds_byte_addr + 16 i.  Issued from
// inline asm so that hipcc does not count it and drain it with vmcnt(0) before the next ds_read; the kernel waits
// for it by hand (s_waitcnt vmcnt(0) at the top of the frame loop).
__device__ __forceinline__ void glds16(const void *sbase, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(unsigned long)((const __attribute__((address_space(3))) char *)p);
}

} // namespace ofdm
