// kernels_sym.hip -- per-OFDM-symbol kernels: FFT / IFFT+CP / unprefix / RX demod / channel estimate / TX symbols.
//
// One OFDM symbol of N points is spread over T = N/8 threads, 8 points per thread.  The transform is a
// Stockham autosort FFT with radix-8 (then radix-4) passes; between passes the points go through a per-symbol
// LDS slab (XOR-swizzled, padded).  For N <= 512 a symbol lives inside ONE 64-lane wavefront, so the passes
// need no workgroup barrier at all: each wave streams 64/T symbols per iteration independently.
// The first pass reads straight from HBM in the Stockham pattern x[t + m*T] (T consecutive lanes read T
// consecutive interleaved-IQ samples), so a symbol is read exactly once and never staged twice.
//
// Roofline: HBM.  RX demod moves 8 B per input sample (cyclic prefix skipped: 8*N of every 8*(N+CP) bytes
// are actually fetched) + bps*carriers/8 output bytes per symbol; 5 N log2 N flops per symbol are ~3-6 flop/B.
#include "device_common.hpp"
#include "kernels.hpp"

extern "C" __device__ float __ocml_atan2pi_f32(float, float); // atan2(y, x) / pi (ROCm device library)

namespace ofdm {

enum { M_FFT = 0, M_IFFT = 1, M_IFFT_CP = 2, M_DEMOD = 3, M_CHEST = 4, M_TX = 5 };

template <int T> __device__ __forceinline__ float symbol_sum(float x, float *red, int slot, int t) {
    // sum over the T threads of one symbol
    constexpr int W = T < 64 ? T : 64;
#pragma unroll
    for (int m = W / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    if (T > 64) {
        constexpr int NW = T / 64;
        __syncthreads();
        if ((t & 63) == 0) red[slot * NW + (t >> 6)] = x;
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) s += red[slot * NW + i];
        x = s;
    }
    return x;
}

template <int N, int MODE>
__global__ __launch_bounds__(Plan<N>::WG) void k_sym(SymParams p) {
    typedef Plan<N> P;
    constexpr int T = P::T, G = P::G;
    constexpr bool INV = (MODE == M_IFFT || MODE == M_IFFT_CP || MODE == M_TX);
    constexpr int K = N / 64; // carrier-map tiling factor (EXT-4)

    __shared__ cf lds[G * P::LDS_SYM];
    __shared__ unsigned char idx_lds[(MODE == M_DEMOD) ? G * N : 4];
    __shared__ float red[(T > 64) ? G * (T / 64) : 1];
    __shared__ __align__(4) unsigned char txb[(MODE == M_TX) ? G * (N + 8) : 4]; // TX: the symbol's slice of the byte stream

    const int tid = threadIdx.x;
    const int t = tid % T, slot = tid / T;
    cf *buf = lds + slot * P::LDS_SYM;

    cf w[P::NTW > 0 ? P::NTW : 1];
    load_twiddles<N>(p.tw, t, w);

    const int S = p.sym_len;     // N + CP
    const int cp = S - N;
    const int c0 = t / K;        // reference carrier class of bin t + m*T is c0 + 8m
    const long long total = (MODE == M_CHEST) ? p.n_frames : p.n_frames * (long long)p.syms_per_frame;

    // symbol index -> (frame, symbol in frame); 32-bit division whenever the batch allows it
    const bool small = total < 0x7fffffffLL;
    auto split = [&](long long sg, long long &f, int &k) {
        if (small) { const unsigned q = (unsigned)sg / (unsigned)p.syms_per_frame; f = q; k = (int)((unsigned)sg - q * (unsigned)p.syms_per_frame); }
        else { f = sg / p.syms_per_frame; k = (int)(sg - f * p.syms_per_frame); }
    };
    // first-pass loads of symbol sg in the Stockham pattern x[t + m*T] (zero past frame_len / past the batch)
    auto fetch = [&](long long sg, cf *dst) {
        if (sg < total) {
            long long f; int k;
            split(sg, f, k);
            const long long off = p.offset ? p.offset[f] : 0;
            const long long n0 = (long long)(p.first_symbol + k) * p.in_sym_stride + p.in_skip;
            const cf *src = p.in + f * p.frame_stride;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                long long pos = off + n0 + t + m * T;
                dst[m] = pos < p.frame_len ? src[pos] : make_float2(0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) dst[m] = make_float2(0.f, 0.f);
        }
    };
    // software pipeline: the NEXT iteration's samples are in flight (8 x 8 B per lane) while this one runs its passes
    constexpr bool PREFETCH = (MODE != M_CHEST && MODE != M_TX);
    cf pre[8];
    if (PREFETCH) fetch((long long)blockIdx.x * G + slot, pre);

    // TX: a symbol's slice of the byte stream is at most N bytes = 2 dwords per thread; they are fetched one symbol
    // ahead (the stream = [16-byte little-endian length | payload | zeros], or the plain bytes in continuous mode)
    const int tx_nd = (MODE == M_TX) ? (p.guard ? 48 * K : N) : 0;
    const int tx_sym_bytes = tx_nd * p.bps / 8;
    const bool tx_raw = p.tx_raw_total >= 0;
    const bool tx_aligned = (MODE == M_TX) && ((reinterpret_cast<uintptr_t>(p.payload) | (uintptr_t)p.payload_stride) & 3) == 0;
    auto tx_dword = [&](const uint8_t *pay, long long len, long long by) -> unsigned { // stream bytes by .. by+3
        if (!tx_raw && by < 16) return by < 8 ? (unsigned)((unsigned long long)len >> (8 * by)) : 0u;
        const long long off = tx_raw ? by : by - 16;
        if (tx_aligned && (off & 3) == 0 && off + 4 <= len) return *reinterpret_cast<const unsigned *>(pay + off);
        unsigned v = 0;
        for (int j = 0; j < 4; ++j) if (off + j < len) v |= (unsigned)pay[off + j] << (8 * j);
        return v;
    };
    auto tx_fetch = [&](long long sg, unsigned &d0, unsigned &d1) {
        d0 = d1 = 0u;
        if (sg >= total) return;
        long long f; int k;
        split(sg, f, k);
        long long len = p.payload_len ? row_len(p.payload_len[f], p.payload_bytes) : p.payload_bytes;
        if (tx_raw) { const long long left = p.tx_raw_total - f * p.payload_stride; len = left < 0 ? 0 : (left < len ? left : len); }
        const uint8_t *pay = p.payload + f * p.payload_stride;
        const long long sb0 = (long long)k * tx_sym_bytes;
        if (4 * t < tx_sym_bytes + 4) d0 = tx_dword(pay, len, sb0 + 4 * t);
        if (4 * (t + T) < tx_sym_bytes + 4) d1 = tx_dword(pay, len, sb0 + 4 * (t + T));
    };
    unsigned txd0 = 0, txd1 = 0;
    if (MODE == M_TX) tx_fetch((long long)blockIdx.x * G + slot, txd0, txd1);

    for (long long base = (long long)blockIdx.x * G; base < total; base += (long long)gridDim.x * G) {
        const long long sigma = base + slot;
        const bool valid = sigma < total;
        long long f = 0; int k = 0;
        if (valid) {
            if (MODE == M_CHEST) { f = sigma; }
            else split(sigma, f, k);
        }
        cf v[8];

        if (MODE == M_CHEST) {
            // estimate_channel (src/receiver.rs:212-229): mean over the 5 training blocks of FFT(block)/training
            cf acc[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m] = make_float2(0.f, 0.f);
            const long long off = (valid && p.offset) ? p.offset[f] : 0;
            const double turns = (valid && p.f_delta) ? p.f_delta[f] * 0.15915494309189533577 : 0.0;
            for (int b = 0; b < 5; ++b) {
                const long long n0 = (long long)(5 + b) * S + cp; // sample id of the first FFT sample
                if (valid) {
                    const cf *src = p.in + f * p.frame_stride;
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        long long pos = off + n0 + t + m * T;
                        v[m] = pos < p.frame_len ? src[pos] : make_float2(0.f, 0.f);
                    }
                    if (p.f_delta) {
                        cf ph = cfo_phasor(turns, n0 + t), st = cfo_phasor(turns, T);
#pragma unroll
                        for (int m = 0; m < 8; ++m) { v[m] = cmul(v[m], ph); ph = cmul(ph, st); }
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < 8; ++m) v[m] = make_float2(0.f, 0.f);
                }
                // the transform is linear: the derotated blocks are summed in the time domain and transformed once
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m] = cadd(acc[m], v[m]);
            }
            fft_symbol<N, false>(acc, buf, t, w);
            group_sync<T>();
            if (valid) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    int bin = t + m * T;
                    cf h = cmul(acc[m], p.inv_training[bin]);
                    p.out[f * N + bin] = make_float2(h.x / 5.0f, h.y / 5.0f);
                }
            }
            continue;
        }

        // ------------------------------------------------------------------ load (first pass pattern)
        if (MODE == M_TX) {
            // modulate + encode_block (src/transmitter.rs:108-165): bin -> null / pilot / next data point
            const int nd = p.guard ? 48 * K : N;
            const int sym_bytes = nd * p.bps / 8;           // nd is a multiple of 8
            unsigned char *sbuf = txb + slot * (N + 8);
            const bool raw = p.tx_raw_total >= 0;
            long long len = 0, nsym = 0;
            if (valid) {
                len = p.payload_len ? row_len(p.payload_len[f], p.payload_bytes) : p.payload_bytes;
                if (raw) { // this symbol's share of the continuous byte stream (zeros once it runs dry, transmitter.rs:158-160)
                    const long long left = p.tx_raw_total - f * p.payload_stride;
                    len = left < 0 ? 0 : (left < len ? left : len);
                }
                nsym = ((raw ? len : 16 + len) * 8 + p.bps - 1) / p.bps;
            }
            {   // this symbol's dwords were fetched during the previous symbol; fetch the next one's now
                unsigned *sw = reinterpret_cast<unsigned *>(sbuf);
                if (4 * t < sym_bytes + 4) sw[t] = txd0;
                if (4 * (t + T) < sym_bytes + 4) sw[t + T] = txd1;
                tx_fetch(base + (long long)gridDim.x * G + slot, txd0, txd1);
            }
            group_sync<T>();
            if (valid) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    int c = c0 + 8 * m, cls = carrier_class64(c, p.guard);
                    cf z = make_float2(0.f, 0.f);
                    if (cls == 2) z = make_float2(1.f, 0.f);
                    else if (cls == 0) {
                        int q = p.guard ? data_classes_below64(c) * K + (t % K) : (t + m * T);
                        long long g = (long long)k * nd + q;
                        if (g < nsym) {
                            const int bit = q * p.bps; // inside this symbol's slice
                            const unsigned two = (unsigned)sbuf[bit >> 3] | ((unsigned)sbuf[(bit >> 3) + 1] << 8);
                            z = map_point((two >> (bit & 7)) & ((1u << p.bps) - 1u), p.bps);
                        }
                    }
                    v[m] = z;
                }
            } else {
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = make_float2(0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = pre[m];
            fetch(base + (long long)gridDim.x * G + slot, pre); // issue the next symbol's loads now
            if (MODE == M_DEMOD && p.f_delta && valid) {
                const long long n0 = (long long)(p.first_symbol + k) * p.in_sym_stride + p.in_skip;
                const double turns = p.f_delta[f] * 0.15915494309189533577; // 1/(2 pi)
                cf ph = cfo_phasor(turns, n0 + t), st = cfo_phasor(turns, T);
#pragma unroll
                for (int m = 0; m < 8; ++m) { v[m] = cmul(v[m], ph); ph = cmul(ph, st); }
            }
        }
        if (MODE == M_DEMOD && valid && p.nsym_frame && k >= p.nsym_frame[f]) {
            // this frame holds fewer symbols (short capture / failed sync): nothing is written for it
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = make_float2(0.f, 0.f);
        }

        fft_symbol<N, INV>(v, buf, t, w);

        // ------------------------------------------------------------------ epilogues
        if (MODE == M_FFT || MODE == M_IFFT) {
            if (valid) {
                cf *dst = p.out + sigma * N;
#pragma unroll
                for (int m = 0; m < 8; ++m) dst[t + m * T] = INV ? cscale(v[m], 1.0f / N) : v[m];
            }
        } else if (MODE == M_IFFT_CP || MODE == M_TX) {
            // prefix_block (src/transmitter.rs:168-181): out = [x[N-CP..N), x[0..N)]
            if (valid) {
                cf *dst = (MODE == M_TX) ? p.out + f * p.out_stride_s + (long long)((p.tx_raw_total >= 0 ? 0 : 10) + k) * S : p.out + sigma * S;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    int n = t + m * T;
                    cf z = cscale(v[m], 1.0f / N);
                    dst[cp + n] = z;
                    if (n >= N - cp) dst[n - (N - cp)] = z;
                }
            }
            if (MODE == M_TX) {
                // normalize, pass 1 (src/transmitter.rs:184-188): signed max over re and im, floor 0.
                // One atomic per symbol (per wave for T > 64): symbols of one wave may belong to different frames.
                float mine = 0.f;
                if (valid && p.frame_max) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) mine = fmaxf(mine, fmaxf(v[m].x, v[m].y) * (1.0f / N));
                    constexpr int W = T < 64 ? T : 64;
#pragma unroll
                    for (int s = W / 2; s >= 1; s >>= 1) mine = fmaxf(mine, __shfl_xor(mine, s, 64));
                    if ((t & (W - 1)) == 0) atomicMax(p.frame_max + f, __float_as_uint(mine));
                }
            }
        } else if (MODE == M_DEMOD) {
            const bool live = valid && !(p.nsym_frame && k >= p.nsym_frame[f]);
            const int nd = p.guard ? 48 * K : N;
            // equalise: Y[k] /= H[k] (src/receiver.rs:68-70)
            if (p.hk && live) {
                const cf *h = p.hk + f * p.hk_stride;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    cf hh = h[t + m * T];
                    const float rn = __builtin_amdgcn_rcpf(hh.x * hh.x + hh.y * hh.y); // as k_rxframe64 (1 ulp)
                    cf q = cmulc(v[m], hh);
                    v[m] = make_float2(q.x * rn, q.y * rn);
                }
            }
            // decode_block (src/receiver.rs:106-145): mean pilot angle, rotate the data points by -phase
            if (p.guard) {
                float ang = 0.f; // in units of pi: atan2pi / sincospi need no large-argument reduction
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    if (carrier_class64(c0 + 8 * m, 1) == 2) ang += __ocml_atan2pi_f32(v[m].y, v[m].x);
                ang = symbol_sum<T>(ang, red, slot, t) / (4.0f * K);
                float s, c;
                sincospif(ang, &s, &c);
                const cf rot = make_float2(c, -s);
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = cmul(v[m], rot);
            }
            // demodulate (src/receiver.rs:147-190): hard decision per data bin -> LDS (ordinal order)
            unsigned char *ib = idx_lds + slot * N;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                int c = c0 + 8 * m;
                if (carrier_class64(c, p.guard) == 0) {
                    int q = p.guard ? data_classes_below64(c) * K + (t % K) : (t + m * T);
                    ib[q] = (unsigned char)demap_point(v[m], p.bps);
                    if (p.soft && live) p.soft[(f * p.syms_per_frame + k) * (long long)nd + q] = v[m];
                }
            }
            group_sync<T>();
            // pack bps-bit indices LSB-first into bytes (src/utils.rs:30-36) and write whole dwords
            if (live) {
                const int nbytes = nd * p.bps / 8;
                unsigned char *dst = p.out_bytes + f * p.out_stride + (long long)k * nbytes;
                if ((nbytes & 3) == 0 && (p.out_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out_bytes) & 3) == 0) {
                    for (int wd = t; wd < nbytes / 4; wd += T) {
                        unsigned acc = 0;
                        int i = (32 * wd) / p.bps;
                        for (;; ++i) {
                            int sh = i * p.bps - 32 * wd;
                            if (sh >= 32) break;
                            unsigned val = ib[i];
                            acc |= sh >= 0 ? (val << sh) : (val >> (-sh));
                        }
                        reinterpret_cast<unsigned *>(dst)[wd] = acc;
                    }
                } else {
                    for (int by = t; by < nbytes; by += T) {
                        unsigned acc = 0;
                        int i = (8 * by) / p.bps;
                        for (;; ++i) {
                            int sh = i * p.bps - 8 * by;
                            if (sh >= 8) break;
                            unsigned val = ib[i];
                            acc |= sh >= 0 ? (val << sh) : (val >> (-sh));
                        }
                        dst[by] = (unsigned char)acc;
                    }
                }
            }
            group_sync<T>();
        }
    }
}

// ---------------------------------------------------------------------------------------------- launchers
template <int N, int MODE> static hipError_t launch_sym(const SymParams &p, hipStream_t st, int num_cu) {
    typedef Plan<N> P;
    const long long total = (MODE == M_CHEST) ? p.n_frames : p.n_frames * (long long)p.syms_per_frame;
    if (total <= 0) return hipSuccess;
    long long groups = (total + P::G - 1) / P::G;
    long long cap = (long long)num_cu * 8; // persistent: ~8 workgroups per CU, grid-stride over symbol groups
    int grid = (int)(groups < cap ? groups : cap);
    static const char *const names[] = {"k_sym<fft>", "k_sym<ifft>", "k_sym<ifft_cp>", "k_sym<demod>", "k_sym<chest>", "k_sym<tx>"};
    trace_add(p.trace, names[MODE]);
    hipLaunchKernelGGL((k_sym<N, MODE>), dim3(grid), dim3(P::WG), 0, st, p);
    return hipGetLastError();
}

template <int MODE> static hipError_t dispatch_n(int n, const SymParams &p, hipStream_t st, int num_cu) {
    switch (n) {
    case 64: return launch_sym<64, MODE>(p, st, num_cu);
    case 128: return launch_sym<128, MODE>(p, st, num_cu);
    case 256: return launch_sym<256, MODE>(p, st, num_cu);
    case 512: return launch_sym<512, MODE>(p, st, num_cu);
    case 1024: return launch_sym<1024, MODE>(p, st, num_cu);
    case 2048: return launch_sym<2048, MODE>(p, st, num_cu);
    case 4096: return launch_sym<4096, MODE>(p, st, num_cu);
    default: return hipErrorInvalidValue;
    }
}

hipError_t run_fft(int n, const SymParams &p, bool inverse, hipStream_t st, int cu) {
    return inverse ? dispatch_n<M_IFFT>(n, p, st, cu) : dispatch_n<M_FFT>(n, p, st, cu);
}
hipError_t run_ifft_cp(int n, const SymParams &p, hipStream_t st, int cu) { return dispatch_n<M_IFFT_CP>(n, p, st, cu); }
hipError_t run_demod(int n, const SymParams &p, hipStream_t st, int cu) { return dispatch_n<M_DEMOD>(n, p, st, cu); }
hipError_t run_chest(int n, const SymParams &p, hipStream_t st, int cu) { return dispatch_n<M_CHEST>(n, p, st, cu); }
hipError_t run_tx_symbols(int n, const SymParams &p, hipStream_t st, int cu) { return dispatch_n<M_TX>(n, p, st, cu); }

} // namespace ofdm
