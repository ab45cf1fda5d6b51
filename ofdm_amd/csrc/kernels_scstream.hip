// kernels_scstream.hip -- k_sc_stream: Schmidl-Cox timing for periods L = 160 .. 5120 (N = 128 .. 4096) in ONE streaming pass
// that stops as soon as the decision is determined.
//
// The detector is threshold-then-peak (oracle: orc_sc_sync): d1 = first lag with M >= threshold,
// d_hat = first maximum of M over [d1, d1 + W].  Nothing after lag d1 + W can change the answer, so a frame that holds a packet
// needs its samples only up to d1 + 2 W + L -- for the config-4 frame about half of the 18 176-sample slot -- and the receive
// kernel reads the rest.  Round 2 made two passes for this: k_scb_chunks summed EVERY chunk of every frame (1.21 reads per
// sample), then k_scb_fine re-read ~60 % of the frame around the packet (kernels_scbig.hip, kept for L > 1280).
//
// One wavefront per frame, persistent over the batch, no workgroup barrier anywhere:
//   * tiles of 640 samples arrive by LDS-DMA (global_load_lds_dwordx4, five 1 KiB pieces) into a ring of L + 3 tiles, two
//     tiles ahead of the one being summed;
//   * per tile, lane l sums ONE 10-sample micro-chunk of e[n] = |r[n]|^2 and one of q[n] = conj(r[n]) r[n + L] in f64 (products of
//     f32 samples are exact in f64); a DPP scan turns them into prefixes Ep[], Q[] kept in LDS rings of (W + L) / 10 and W / 10
//     entries, so that the EXACT sums at every 10th lag are prefix differences
//         P(d) = Q[d + W] - Q[d],   E(d) = Ep[d + W] - Ep[d],   R(d) = Ep[d + W + L] - Ep[d + L];
//   * each new 10-lag interval [d, d + 10) gets an upper bound of its metric,
//         |P| <= |P(d)| + 1/2 (e-sums of the four 10-sample pieces that enter / leave),  E >= E(d) - e-sum,  R likewise
//     (0.5 % slack at N = 1024).  Before the crossing only intervals whose bound reaches the threshold are evaluated lag by lag
//     (their 36 samples come back from L2); inside the peak window an interval stays on a short live list only while its bound
//     reaches the best exact value seen so far, and the list is evaluated exactly when the window closes (or the list is full);
//   * then the wavefront writes d_hat, CFO = arg P / L, M and moves on: the rest of the frame is never fetched.
// Every decision is taken on f64 sums, as in k_sc_tile / k_scb_fine: timing indices equal the f64 oracle's except on ties below
// f64 resolution.  Roofline: HBM, 8 B per sample actually needed; the bench reports achieved rates against the WHOLE capture.
#include "device_common.hpp"
#include "kernels.hpp"
#include <limits.h>

namespace ofdm {

namespace {

constexpr int ST_T = 640;            // samples per tile (64 micro-chunks of 10: one per lane)
constexpr int ST_PIECES = 5;         // 1 KiB DMA pieces per tile
constexpr int ST_LIVE = 32;          // live-list capacity (flushed by exact evaluation when full)

struct StParams {
    const float2 *in;
    long long n_frames, frame_stride, frame_len, n_lags;
    int L, W;
    double threshold;
    int ring;                        // sample ring capacity (a multiple of ST_T, >= L + 3 ST_T)
    int epn, qn;                     // prefix ring capacities (entries)
    int32_t *d_hat;
    double *f_delta;
    float *metric;
    int debug;   // profile build only (Tuning::debug_sc = 40 + i): metric[f] = s_memtime ticks of section i of the frame --
                 // producer: 0 tile waits, 1 sums + scans, 2 barrier waits; consumer: 3 bounds + lists, 4 evaluations before the
                 // crossing, 5 barrier waits, 6 closing the window; 7 the whole frame (consumer)
};

struct SSums { double pr, pi, e, r; };
struct SCand { double num, den, pr, pi; int lag; };
// first maximum wins, whatever the order of discovery: strictly greater replaces; equal replaces only from a lower lag
__device__ __forceinline__ SCand sc_pick(SCand a, SCand b) {
    const double lhs = b.num * a.den, rhs = a.num * b.den;
    return (lhs > rhs || (lhs == rhs && b.lag < a.lag)) ? b : a;
}
__device__ __forceinline__ SCand sc_shfl_xor(SCand a, int d) {
    return SCand{__shfl_xor(a.num, d, 64), __shfl_xor(a.den, d, 64), __shfl_xor(a.pr, d, 64), __shfl_xor(a.pi, d, 64),
                 __shfl_xor(a.lag, d, 64)};
}
__device__ __forceinline__ SCand sc_wave_best(SCand c) {
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) c = sc_pick(c, sc_shfl_xor(c, sft));
    return c;
}

template <int CTRL, int ROW_MASK> __device__ __forceinline__ double st_dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, true); // out-of-range / masked lanes read 0
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, true);
    return __hiloint2double(hi, lo);
}
// inclusive prefix sum over the 64 lanes of a wavefront
__device__ __forceinline__ double st_scan(double x) {
    x += st_dpp<0x111, 0xF>(x); // row_shr:1
    x += st_dpp<0x112, 0xF>(x); // row_shr:2
    x += st_dpp<0x114, 0xF>(x); // row_shr:4
    x += st_dpp<0x118, 0xF>(x); // row_shr:8
    x += st_dpp<0x142, 0xA>(x); // row_bcast:15 into rows 1 and 3
    x += st_dpp<0x143, 0xC>(x); // row_bcast:31 into rows 2 and 3
    return x;
}
__device__ __forceinline__ double st_readlane(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float st_dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xF, true));
}
// maximum of a non-negative float over the wavefront (DPP only: no LDS round trips), returned wave-uniform
__device__ __forceinline__ float st_wave_max(float x) {
    x = fmaxf(x, st_dpp_f<0x111, 0xF>(x));
    x = fmaxf(x, st_dpp_f<0x112, 0xF>(x));
    x = fmaxf(x, st_dpp_f<0x114, 0xF>(x));
    x = fmaxf(x, st_dpp_f<0x118, 0xF>(x));
    x = fmaxf(x, st_dpp_f<0x142, 0xA>(x));
    x = fmaxf(x, st_dpp_f<0x143, 0xC>(x));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
__device__ __forceinline__ SCand sc_readlane(SCand c, int l) {
    auto rd = [&](double v) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
    return SCand{rd(c.num), rd(c.den), rd(c.pr), rd(c.pi), __builtin_amdgcn_readlane(c.lag, l)};
}
template <int N> __device__ __forceinline__ void st_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int st_wrap(int x, int m) { return x >= m ? x - m : x; }   // ring index of x in [0, 2 m): no integer division in the loop
// values that ARE wave-uniform but derive from cross-lane operations (which the compiler's divergence analysis cannot see through)
__device__ __forceinline__ int st_uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ const char *st_uni_ptr(const char *q) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<const char *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void st_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

} // namespace

// Two wavefronts per frame.  Wavefront 1, the PRODUCER, owns the DMA ring and does the micro-chunk sums and scans of tile i;
// wavefront 0, the CONSUMER, does the bounds, the lists and the lag-by-lag evaluations of step i - 1 at the same time; one
// workgroup barrier per tile separates them (the prefix rings hold 64 entries more than one step needs).  The consumer's
// global-memory round trips (evaluations) stall only itself and, through the barrier, at most the tile in progress.
// DELAY = L / 640 when that is 1 or 2: the partner micro-chunk L samples earlier is the one the producer's lane read DELAY tiles
// ago and stays in its registers (the LDS ring then holds only the tile being read and two in flight); 0: re-read from the ring.
template <int DELAY>
__global__ __launch_bounds__(128, 3) void k_sc_stream(StParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    cf *ring = reinterpret_cast<cf *>(smem);                                  // [p.ring] samples: tile s in slot s % ring_tiles
    double *Ep = reinterpret_cast<double *>(ring + p.ring);                   // [epn] exclusive prefix of e over micro-chunks, index j % epn
    double *Qr = Ep + p.epn;                                                  // [qn]  ... of q.re
    double *Qi = Qr + p.qn;                                                   // [qn]  ... of q.im
    double *lsum = Qi + p.qn;                                                 // [ST_LIVE][4] boundary sums of the live intervals
    int *lk = reinterpret_cast<int *>(lsum + 4 * ST_LIVE);                    // [ST_LIVE] their interval index
    float *lub = reinterpret_cast<float *>(lk + ST_LIVE);                     // [ST_LIVE] their bound
    long long *ptk = reinterpret_cast<long long *>(lub + ST_LIVE + 8);        // [3] the producer's section ticks (profile build)
    int *ctl = reinterpret_cast<int *>(lub + ST_LIVE);                        // [2][2], slot = iteration parity: {the frame is decided, tiles needed (s_stop)}

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L = p.L, W = p.W, cL = L / 10, cW = W / 10, cWL = cW + cL;
    const long long n = p.n_lags;
    const double thr = p.threshold;
    const float thr_f = (float)thr * 0.999999f;
    const unsigned ring_lds = lds_addr(ring);
    const int ring_tiles = p.ring / ST_T;
    // tiles the search can need: the last lag's window plus the one extra micro-chunk of its interval's bound
    const long long last_sample = n - 1 + W + L + 20;
    const int s_max = (int)((last_sample + ST_T - 1) / ST_T);                 // tiles 0 .. s_max - 1
    // The consumer has nothing to judge before step i0 - 1 (no interval index k = 64 s - cWL + lane is >= 0 yet), so neither role
    // takes part in a barrier before iteration i0: the producer sums the first tiles of a frame while the consumer is still
    // closing the previous frame's window (list + global memory only: the rings belong to the producer until barrier i0).
    const int i0 = cWL > 63 ? (cWL - 63 + 63) / 64 : 0;
    const bool prof = kProfile && p.debug >= 40;
    auto now = [&]() -> long long { return prof ? (long long)__builtin_amdgcn_s_memtime() : 0; };

    // The two roles are two separate frame loops that execute the SAME sequence of barriers (written as one loop with the roles as
    // branches the register allocator carried both roles' state through both bodies: 70 spilled registers, 1.8x slower, measured;
    // with one frame loop around two role loops it still kept 35 registers more than the larger role needs).
    if (wave == 1) {
    for (long long f = blockIdx.x; f < p.n_frames; f += gridDim.x) {
        const cf *frame = p.in + f * p.frame_stride;
        // Stage tile s into ring slot s % ring_tiles (producer).  Whole tiles inside the capture go by LDS-DMA; a tile that touches
        // the end of the capture is loaded with bounds checks and stored through registers (zeros past frame_len).
        auto issue = [&](int s, int slot_idx) -> bool {   // slot_idx = s % ring_tiles, kept incrementally by the caller
            const long long t0 = (long long)s * ST_T;
            const unsigned slot = (unsigned)slot_idx * (unsigned)(ST_T * sizeof(cf));
            if (t0 + ST_T <= p.frame_len) {
                const char *sb = st_uni_ptr(reinterpret_cast<const char *>(frame + t0));
                const unsigned l0 = (unsigned)st_uni((int)(ring_lds + slot));
#pragma unroll
                for (int j = 0; j < ST_PIECES; ++j) glds16(sb, (unsigned)j * 1024u + (unsigned)lane * 16u, l0 + (unsigned)j * 1024u);
                return true;
            }
            cf *dst = ring + slot_idx * ST_T;
#pragma unroll
            for (int j = 0; j < ST_T / 64; ++j) {
                const long long i = t0 + lane + 64 * j;
                dst[lane + 64 * j] = i < p.frame_len ? frame[i] : make_float2(0.f, 0.f);
            }
            return false;
        };
        // ---- producer state (wavefront 1)
        double run_e = 0.0, run_qr = 0.0, run_qi = 0.0;   // totals so far = Ep / Q at the next index
        int issued = 0;                                   // tiles issued so far
        bool dma0 = false, dma1 = false;                  // whether tiles s / s + 1 went by DMA
        float4 xo[DELAY > 0 ? DELAY : 1][5];              // DELAY > 0: this lane's micro-chunks of the last DELAY tiles
#pragma unroll
        for (int dly = 0; dly < (DELAY > 0 ? DELAY : 1); ++dly)
#pragma unroll
            for (int i = 0; i < 5; ++i) xo[dly][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        int s_stop = s_max;                               // tiles beyond this one are not needed (the consumer learns it with d1)
        // ring positions, advanced by one tile / 64 entries per step with a conditional wrap (no integer division in the loop)
        int slot_s = 0;                                   // i % ring_tiles (producer)
        int slot_i = 2 % ring_tiles;                      // (i + 2) % ring_tiles
        int e_new = 0;                                    // (64 i) % epn: Ep entry ke + 1 goes to e_new + lane + 1
        int q_new = p.qn - (cL % p.qn);                   // (64 i - cL) mod qn
        if (q_new == p.qn) q_new = 0;
        {
            // the first two tiles go out before the frame's barrier: the ring belongs to the producer alone, and the consumer may
            // still be closing the previous frame's window
            st_wait_vm<0>();                              // (tiles of the previous frame issued beyond its last step)
            if (s_max > 0) { dma0 = issue(0, 0); issued = 1; }
            if (s_max > 1) { dma1 = issue(1, 1 % ring_tiles); issued = 2; }
            if (lane == 0) { Ep[0] = 0.0; Qr[0] = 0.0; Qi[0] = 0.0; }   // (the consumer left the rings at the last barrier of the previous frame)
            st_fence();
            long long pt0 = 0, pt1 = 0, pt2 = 0;
#pragma clang loop unroll(disable)
            for (int i = 0;; ++i) {
                const long long ta = now();
                long long tb = ta;
                if (i < s_stop) {
                // ---- PRODUCER, tile i: it has landed (this wavefront issued every piece of it); the next one stays in flight
                if (dma0) { if (issued > i + 1 && dma1) st_wait_vm<ST_PIECES>(); else st_wait_vm<0>(); }
                st_fence();
                tb = now();
                // Tile i + 2 is requested two tiles ahead.  DELAY == 0: into a slot of the L + 3-tile ring that nobody reads any more.
                // DELAY > 0: the ring has TWO slots (25 instead of 30 KB of LDS per frame: six frames per CU), so it goes into tile i's own
                // slot as soon as this lane's samples of tile i are in registers (below).
                bool dma2 = false;
                if (DELAY == 0 && issued == i + 2 && issued < s_stop) { dma2 = issue(issued, slot_i); ++issued; }
                // micro-chunk sums: e over samples [10 ke, 10 ke + 10) of this tile, q over [10 kq, ..) with partners L later (= this tile)
                const int ke = 64 * i + lane, kq = ke - cL;
                double se = 0.0, sqr = 0.0, sqi = 0.0;
                {
                    const int ie = slot_s * ST_T + 10 * lane;
                    const float4 *pe = reinterpret_cast<const float4 *>(ring + ie);
                    float4 x[5], y[5];
#pragma unroll
                    for (int j = 0; j < 5; ++j) x[j] = pe[j];
                    if (DELAY > 0) {
#pragma unroll
                        for (int j = 0; j < 5; ++j) y[j] = xo[DELAY - 1][j];
                    } else if (kq >= 0) {
                        const float4 *pq = reinterpret_cast<const float4 *>(ring + st_wrap(ie - L + p.ring, p.ring));   // L < ring
#pragma unroll
                        for (int j = 0; j < 5; ++j) y[j] = pq[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 5; ++j) y[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    if (DELAY > 0) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every lane's reads of tile i are done (one wavefront owns the ring)
                        if (issued == i + 2 && issued < s_stop) { dma2 = issue(issued, slot_i); ++issued; }
                    }
                    dma0 = dma1; dma1 = dma2;
#pragma unroll
                    for (int j = 0; j < 5; ++j) {   // y = r[n], r[n + 1]; x = r[n + L], r[n + L + 1]
                        const double br = x[j].x, bi = x[j].y, dr = x[j].z, di = x[j].w;
                        const double ar = y[j].x, ai = y[j].y, cr = y[j].z, ci = y[j].w;
                        se += br * br + bi * bi; se += dr * dr + di * di;
                        sqr += ar * br + ai * bi; sqi += ar * bi - ai * br;
                        sqr += cr * dr + ci * di; sqi += cr * di - ci * dr;
                    }
                    if (DELAY > 0) {
#pragma unroll
                        for (int dly = DELAY - 1; dly > 0; --dly)
#pragma unroll
                            for (int j = 0; j < 5; ++j) xo[dly][j] = xo[dly - 1][j];
#pragma unroll
                        for (int j = 0; j < 5; ++j) xo[0][j] = x[j];
                    }
                }
                // inclusive scans -> prefix entries ke + 1 (Ep) and kq + 1 (Q)
                const double pe_ = st_scan(se) + run_e, pqr = st_scan(sqr) + run_qr, pqi = st_scan(sqi) + run_qi;
                Ep[st_wrap(e_new + lane + 1, p.epn)] = pe_;
                if (kq >= 0) { const int j = st_wrap(q_new + lane + 1, p.qn); Qr[j] = pqr; Qi[j] = pqi; }
                run_e = st_readlane(pe_, 63);
                if (64 * i + 63 - cL >= 0) { run_qr = st_readlane(pqr, 63); run_qi = st_readlane(pqi, 63); }
                slot_s = st_wrap(slot_s + 1, ring_tiles); slot_i = st_wrap(slot_i + 1, ring_tiles);
                e_new = st_wrap(e_new + 64, p.epn); q_new = st_wrap(q_new + 64, p.qn);
            }
                const long long tc = now();
                if (i < i0) { if (prof) { pt0 += tb - ta; pt1 += tc - tb; } continue; }   // (i0 < s_max always: W + L + 20 samples take more than i0 tiles)
                lds_barrier();   // tile i is summed, step i - 1 is judged
                if (prof) { const long long td = now(); pt0 += tb - ta; pt1 += tc - tb; pt2 += td - tc; }
                const int stop = __builtin_amdgcn_readfirstlane(ctl[2 * (i & 1)]);
                s_stop = __builtin_amdgcn_readfirstlane(ctl[2 * (i & 1) + 1]);
                if (stop || i >= s_stop) break;
            }
            if (prof && lane == 0) { ptk[0] = pt0; ptk[1] = pt1; ptk[2] = pt2; }
        }
        st_fence();
    }
    st_wait_vm<0>();
    } else {
    for (long long f = blockIdx.x; f < p.n_frames; f += gridDim.x) {
        const cf *frame = p.in + f * p.frame_stride;
        // Lag-by-lag evaluation of interval k (lags 10 k .. 10 k + 9) from its boundary sums (consumer); the 4 x 9 samples of the slide
        // come from global memory (L2 / MALL: the producer streamed them moments ago).  Over lags in [lo, hi]: the first lag with
        // M >= threshold (INT_MAX: none) and the first maximum.  from_cross: the maximum only counts lags from this interval's own
        // first crossing on (the search before the crossing: the lane that holds the wavefront's lowest crossing then already has
        // the candidates of the peak window's first interval).
        auto eval_interval = [&](bool active, int k, SSums x, long long lo, long long hi, bool from_cross, int &cross, SCand &best) {
            cross = INT_MAX;
            best = SCand{-1.0, 1.0, 0.0, 0.0, INT_MAX};
            const long long d0 = 10LL * k;
            // The loads are UNCONDITIONAL, from an address that is always valid (the sample itself, or sample 0 of the frame for a
            // lane / an index that must read as zero), and the zeros are selected afterwards: written as `cond ? frame[i] : 0` the
            // compiler branches around every load and waits for it inside the branch -- 36 serialized round trips per evaluation
            // (5-16 thousand cycles, measured) instead of one.  They go in THREE batches of 12 (three slide steps each): all 36 at
            // once cost 72 VGPRs and held the kernel at two waves per SIMD; three round trips per evaluation buy a third wave
            // (six frames in flight per CU instead of four).
#pragma clang loop unroll(disable)
            for (int b3 = 0; b3 < 3; ++b3) {
                cf s0[3], s1[3], s2[3], s3[3];
                bool ok0[3], ok1[3], ok2[3], ok3[3];
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    const long long i0 = d0 + 3 * b3 + jj, i1 = i0 + L, i2 = i0 + W, i3 = i2 + L;
                    ok0[jj] = active && i0 < p.frame_len; ok1[jj] = active && i1 < p.frame_len;
                    ok2[jj] = active && i2 < p.frame_len; ok3[jj] = active && i3 < p.frame_len;
                    s0[jj] = ld_cf(frame + (ok0[jj] ? i0 : 0)); s1[jj] = ld_cf(frame + (ok1[jj] ? i1 : 0));
                    s2[jj] = ld_cf(frame + (ok2[jj] ? i2 : 0)); s3[jj] = ld_cf(frame + (ok3[jj] ? i3 : 0));
                }
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    if (!ok0[jj]) s0[jj] = make_float2(0.f, 0.f);
                    if (!ok1[jj]) s1[jj] = make_float2(0.f, 0.f);
                    if (!ok2[jj]) s2[jj] = make_float2(0.f, 0.f);
                    if (!ok3[jj]) s3[jj] = make_float2(0.f, 0.f);
                }
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    const long long lag = d0 + 3 * b3 + jj;
                    const double num = x.pr * x.pr + x.pi * x.pi, den = x.e * x.r;
                    if (active && lag >= lo && lag <= hi && den > 0.0) {
                        if (cross == INT_MAX && num >= thr * den) cross = (int)lag;
                        if (!from_cross || cross != INT_MAX) best = sc_pick(best, SCand{num, den, x.pr, x.pi, (int)lag});
                    }
                    const double r0 = s0[jj].x, i0 = s0[jj].y, r1 = s1[jj].x, i1 = s1[jj].y, r2 = s2[jj].x, i2 = s2[jj].y, r3 = s3[jj].x, i3 = s3[jj].y;
                    x.pr += (r2 * r3 + i2 * i3) - (r0 * r1 + i0 * i1);
                    x.pi += (r2 * i3 - i2 * r3) - (r0 * i1 - i0 * r1);
                    x.e += (r2 * r2 + i2 * i2) - (r0 * r0 + i0 * i0);
                    x.r += (r3 * r3 + i3 * i3) - (r1 * r1 + i1 * i1);
                }
            }
            {   // the tenth lag: after the ninth slide step
                const long long lag = d0 + 9;
                const double num = x.pr * x.pr + x.pi * x.pi, den = x.e * x.r;
                if (active && lag >= lo && lag <= hi && den > 0.0) {
                    if (cross == INT_MAX && num >= thr * den) cross = (int)lag;
                    if (!from_cross || cross != INT_MAX) best = sc_pick(best, SCand{num, den, x.pr, x.pi, (int)lag});
                }
            }
        };

        // ---- consumer state (wavefront 0)
        long long d1 = -1, hi = n - 1;                    // first crossing; last lag of the peak window
        int k1 = 0, kE = INT_MAX;                         // interval of d1; interval of hi
        int n_live = 0;
        SCand best = SCand{-1.0, 1.0, 0.0, 0.0, INT_MAX};     // the crossing's interval and every flushed list
        SCand mybest = SCand{-1.0, 1.0, 0.0, 0.0, INT_MAX};   // per lane: the exact boundary candidates this lane has met (merged once, at the end)
        float best_lo = 0.f;                                  // LOWER bound of the best metric met so far (f32, rounded down): prunes the list
        bool done = false;
        // evaluate every interval on the live list exactly, fold the result into `best`, empty the list
        auto flush_live = [&]() {
            st_fence();
            const bool act = lane < n_live;
            const int k = act ? lk[lane] : 0;
            const SSums x = act ? SSums{lsum[4 * lane], lsum[4 * lane + 1], lsum[4 * lane + 2], lsum[4 * lane + 3]} : SSums{0, 0, 0, 0};
            int cr; SCand c;
            eval_interval(act, k, x, d1, hi, false, cr, c);
            best = sc_pick(best, sc_wave_best(c));
            if (best.lag != INT_MAX) best_lo = fmaxf(best_lo, (float)(best.num / best.den) * 0.99999f);
            n_live = 0;
            st_fence();
        };

        int s_stop = s_max;                               // tiles beyond this one are not needed (the consumer learns it with d1)
        int e_k = p.epn - (cWL % p.epn);                  // (64 s - cWL) mod epn: interval k = 64 s - cWL + lane of the consumer's step s
        if (e_k == p.epn) e_k = 0;
        int q_k = p.qn - (cWL % p.qn);                    // (64 s - cWL) mod qn
        if (q_k == p.qn) q_k = 0;
        {
            const long long t_frame = now();
            long long ct3 = 0, ct4 = 0, ct5 = 0;
            // ring positions of the consumer's first step, i0 - 1
            for (int j = 0; j < i0 - 1; ++j) { e_k = st_wrap(e_k + 64, p.epn); q_k = st_wrap(q_k + 64, p.qn); }
#pragma clang loop unroll(disable)
            for (int i = i0;; ++i) {
                const long long ta = now();
                long long tev = 0;
                if (i >= 1 && !done) {

                // ---- CONSUMER, step s = i - 1: the 64 intervals whose bound became computable, k = 64 s - cWL + lane
                const int s = i - 1;
                const int k = 64 * s - cWL + lane;
                const bool valid = k >= 0 && 10LL * k < n;
                SSums b = SSums{0, 0, 0, 0};
                float ub = 0.f;
                if (valid) {
                    // ring positions of k, k + cL, k + cW, k + cWL (each offset is below the ring size: one conditional wrap each)
                    const int e0 = st_wrap(e_k + lane, p.epn), eL = st_wrap(e0 + cL, p.epn), eW = st_wrap(e0 + cW, p.epn), eWL = st_wrap(eW + cL, p.epn);
                    const int q0 = st_wrap(q_k + lane, p.qn), qW = st_wrap(q0 + cW, p.qn);
                    const double E0 = Ep[e0], EL = Ep[eL], EW = Ep[eW], EWL = Ep[eWL];
                    b = SSums{Qr[qW] - Qr[q0], Qi[qW] - Qi[q0], EW - E0, EWL - EL};   // exact sums at lag 10 k
                    const double te0 = Ep[st_wrap(e0 + 1, p.epn)] - E0, teL = Ep[st_wrap(eL + 1, p.epn)] - EL,
                                 teW = Ep[st_wrap(eW + 1, p.epn)] - EW, teWL = Ep[st_wrap(eWL + 1, p.epn)] - EWL;
                    // The bound only has to be an UPPER bound: the exact f64 sums are rounded to f32 (6e-8 each), the square root, the
                    // reciprocal and the products are f32 (1 ulp each), and one factor 1.00001 covers all of it -- no f64 sqrt / divide
                    const float pm = __builtin_amdgcn_sqrtf((float)(b.pr * b.pr + b.pi * b.pi));
                    const float u = pm + 0.5f * (float)(te0 + teL + teW + teWL);
                    const float elo = (float)(b.e - te0), rlo = (float)(b.r - teL);
                    ub = 3.0e38f;
                    if (elo > 0.f && rlo > 0.f) { const float q = u * u * __builtin_amdgcn_rcpf(elo * rlo) * 1.00001f; ub = q < 3.0e38f ? q : 3.0e38f; }
                    if (!(u > 0.f)) ub = 0.f;   // an all-zero neighbourhood: no lag of this interval has a defined metric
                }
                const int k_hi = 64 * s - cWL + 63;               // the highest interval of this step
                if (d1 < 0) {
                    // ---- before the crossing: intervals whose bound reaches the threshold are evaluated lag by lag
                    const bool flag = valid && ub >= thr_f;
                    if (__builtin_amdgcn_ballot_w64(flag)) {
                        int cr; SCand c;
                        const long long te_ = now();
                        eval_interval(flag, k, b, 0, n - 1, true, cr, c);
                        if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tev = now() - te_; }
                        int cmin = cr;
#pragma unroll
                        for (int sft = 32; sft >= 1; sft >>= 1) { const int o = __shfl_xor(cmin, sft, 64); cmin = o < cmin ? o : cmin; }
                        cmin = st_uni(cmin);
                        if (cmin != INT_MAX) {
                            d1 = cmin; k1 = (int)(d1 / 10);
                            hi = d1 + W < n - 1 ? d1 + W : n - 1;
                            kE = (int)(hi / 10);
                            // the stream can stop once interval kE has had its bound: tiles up to sample 10 (kE + cWL + 2)
                            const long long need = (10LL * (kE + cWL + 2) + ST_T - 1) / ST_T;
                            if (need < s_stop) s_stop = need > s + 1 ? (int)need : s + 1;
                            // the crossing's own interval, from the crossing on: held by the one lane whose first crossing is d1
                            // (hi >= d1 + 9 unless the search ends inside this interval, where eval_interval's own range [0, n - 1] agrees)
                            const int src = __builtin_ctzll(__builtin_amdgcn_ballot_w64(cr == cmin));
                            best = sc_readlane(c, src);
                            best_lo = (float)(best.num / best.den) * 0.99999f;
                        }
                    }
                }
                if (d1 >= 0) {
                    // ---- inside the peak window: boundaries are exact candidates (kept per lane, merged once when the window closes); an
                    //      interval stays alive while its bound can beat a lower bound of the best metric met so far
                    const bool inwin = valid && k > k1 && 10LL * k <= hi;
                    float mf = 0.f;
                    if (inwin) {
                        const double num = b.pr * b.pr + b.pi * b.pi, den = b.e * b.r;
                        if (den > 0.0) { mybest = sc_pick(mybest, SCand{num, den, b.pr, b.pi, 10 * k}); mf = (float)num * __builtin_amdgcn_rcpf((float)den); }
                    }
                    best_lo = fmaxf(best_lo, st_wave_max(mf) * 0.99999f);
                    const float need = best_lo;
                    if (n_live > 0) {   // old entries that can still matter, compacted in place
                        st_fence();
                        const bool old = lane < n_live;
                        const int ok_ = old ? lk[lane] : 0;
                        const float oub = old ? lub[lane] : 0.f;
                        SSums os = SSums{0, 0, 0, 0};
                        if (old) os = SSums{lsum[4 * lane], lsum[4 * lane + 1], lsum[4 * lane + 2], lsum[4 * lane + 3]};
                        const bool keep = old && oub >= need;
                        const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
                        if (__builtin_popcountll(km) != n_live) {
                            const int pos = __builtin_popcountll(km & ((1ull << lane) - 1ull));
                            st_fence();
                            if (keep) { lk[pos] = ok_; lub[pos] = oub; lsum[4 * pos] = os.pr; lsum[4 * pos + 1] = os.pi; lsum[4 * pos + 2] = os.e; lsum[4 * pos + 3] = os.r; }
                            n_live = __builtin_popcountll(km);
                            st_fence();
                        }
                    }
                    // new intervals (their interior lags 10 k + 1 .. 10 k + 9, as far as they lie in the window)
                    const bool add = inwin && ub >= need;
                    unsigned long long am = __builtin_amdgcn_ballot_w64(add);
                    while (am) {   // in batches that fit the list; a full list is evaluated exactly and emptied
                        const int room = ST_LIVE - n_live;
                        if (room == 0) { flush_live(); continue; }
                        const int rank = __builtin_popcountll(am & ((1ull << lane) - 1ull));
                        const bool take = add && ((am >> lane) & 1ull) && rank < room;
                        if (take) {
                            const int pos = n_live + rank;
                            lk[pos] = k; lub[pos] = ub; lsum[4 * pos] = b.pr; lsum[4 * pos + 1] = b.pi; lsum[4 * pos + 2] = b.e; lsum[4 * pos + 3] = b.r;
                        }
                        const unsigned long long tm = __builtin_amdgcn_ballot_w64(take);
                        n_live += __builtin_popcountll(tm);
                        am &= ~tm;
                        st_fence();
                    }
                    if (k_hi >= kE) done = true;   // every interval up to kE has been seen: the producer may stop and start the next frame
                }
                e_k = st_wrap(e_k + 64, p.epn); q_k = st_wrap(q_k + 64, p.qn);
                }
                // the verdict of this iteration goes into the control slot of the iteration's parity: the slot is rewritten two
                // iterations later, after a further barrier, so a wavefront that is late reading it can never see the next verdict
                if (lane == 0) { ctl[2 * (i & 1)] = done ? 1 : 0; ctl[2 * (i & 1) + 1] = s_stop; }
                const long long tc = now();
                lds_barrier();   // tile i is summed, step i - 1 is judged
                if (prof) { const long long td = now(); ct3 += tc - ta - tev; ct4 += tev; ct5 += td - tc; }
                const int stop = __builtin_amdgcn_readfirstlane(ctl[2 * (i & 1)]);
                s_stop = __builtin_amdgcn_readfirstlane(ctl[2 * (i & 1) + 1]);
                if (stop || i >= s_stop) break;
            }
            // ---- close the window: what is left on the list is evaluated lag by lag (the producer is already fetching the next frame)
            const long long tcl = now();
            if (d1 >= 0) { flush_live(); best = sc_pick(best, sc_wave_best(mybest)); }
            const long long tend = now();
            if (lane == 0) {
                if (best.lag == INT_MAX || d1 < 0) { p.d_hat[f] = -1; if (p.f_delta) p.f_delta[f] = 0.0; if (p.metric) p.metric[f] = 0.f; }
                else {
                    p.d_hat[f] = best.lag;
                    if (p.f_delta) p.f_delta[f] = atan2(best.pi, best.pr) / (double)L;
                    if (p.metric) p.metric[f] = (float)(best.num / best.den);
                }
                if (prof && p.metric) {   // (the producer left its loop, and wrote its ticks, before this wavefront's last barrier released)
                    const long long v[8] = {ptk[0], ptk[1], ptk[2], ct3, ct4, ct5, tend - tcl, tend - t_frame};
                    p.metric[f] = (float)v[(p.debug - 40) & 7];
                }
            }
        }
        st_fence();
    }
    }
}

// LDS of one frame's workgroup: sample ring (the partner micro-chunk lives in registers when L is one or two tiles), prefix rings,
// live list, control slots, profile ticks
static size_t sc_stream_lds(int L, int W) {
    const int delay = (L % ST_T == 0 && L / ST_T <= 2) ? L / ST_T : 0;
    const int ring = (delay ? 2 : (L + ST_T - 1) / ST_T + 3) * ST_T;
    const int epn = (W + L) / 10 + 136, qn = W / 10 + 136;
    return (size_t)ring * sizeof(float2) + (size_t)(epn + 2 * qn) * sizeof(double) + (size_t)ST_LIVE * (4 * sizeof(double) + 8) + 32 + 32 + 64;
}
// one tile-aligned streaming pass: L = 160 .. 5120 (N = 128 .. 4096) with 80 | L, lags that fit
// 32-bit intervals, rings that fit one CU's LDS (N = 4096: 100 KB, one frame per CU)
bool sc_stream_ok(const ScParams &p) {
    if (p.mode != 0 || p.L % 80 != 0 || p.L < 160 || p.L > 5120 || p.W % p.L != 0 || p.W / p.L > 3) return false;
    if ((reinterpret_cast<uintptr_t>(p.in) & 7) != 0) return false;   // LDS-DMA takes any 4-byte aligned source (tools/lab/glds_align.hip): odd strides are fine
    if (sc_stream_lds(p.L, p.W) > 150 * 1024) return false;
    return p.n_lags > 0 && p.n_lags + p.W + p.L < (1LL << 30);
}

hipError_t run_sc_stream(const ScParams &p, int num_cu, hipStream_t st) {
    if (p.n_frames <= 0) return hipSuccess;
    StParams q;
    q.in = p.in; q.n_frames = p.n_frames; q.frame_stride = p.frame_stride; q.frame_len = p.frame_len; q.n_lags = p.n_lags;
    q.L = p.L; q.W = p.W; q.threshold = p.threshold;
    const int delay = (p.L % ST_T == 0 && p.L / ST_T <= 2) ? p.L / ST_T : 0;
    // samples kept for the partner micro-chunk + the tile being summed + two in flight; with the partner in the producer's registers: two
    // slots (the tile in flight, and the tile being summed whose slot is refilled as soon as its samples are in registers)
    q.ring = (delay ? 2 : (p.L + ST_T - 1) / ST_T + 3) * ST_T;
    q.epn = (p.W + p.L) / 10 + 136; q.qn = p.W / 10 + 136;   // one step of history more than a step needs: producer and consumer overlap
    q.d_hat = p.d_hat; q.f_delta = p.f_delta; q.metric = p.metric;
    q.debug = kProfile ? tuning_or_default(p.tune).debug_sc : 0;
    const size_t lds = sc_stream_lds(p.L, p.W);
    if (lds > 48 * 1024) { // > 64 KB of dynamic LDS needs the attribute; per device, so set on every such call
        const void *fn = delay == 2 ? reinterpret_cast<const void *>(k_sc_stream<2>) : delay == 1 ? reinterpret_cast<const void *>(k_sc_stream<1>)
                                                                                                  : reinterpret_cast<const void *>(k_sc_stream<0>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    long long per_cu = (long long)(160 * 1024) / (long long)lds;
    if (per_cu > 6) per_cu = 6;   // two wavefronts per frame, built for three wavefronts per SIMD
    if (per_cu < 1) per_cu = 1;
    long long grid = (long long)num_cu * per_cu;
    const Tuning &tu = tuning_or_default(p.tune);
    if (tu.grid_cap > 0 && grid > tu.grid_cap) grid = tu.grid_cap;
    if (grid > p.n_frames) grid = p.n_frames;
    trace_add(p.trace, delay ? "k_sc_stream<regs>" : "k_sc_stream");
    if (delay == 2) hipLaunchKernelGGL(k_sc_stream<2>, dim3((unsigned)grid), dim3(128), lds, st, q);
    else if (delay == 1) hipLaunchKernelGGL(k_sc_stream<1>, dim3((unsigned)grid), dim3(128), lds, st, q);
    else hipLaunchKernelGGL(k_sc_stream<0>, dim3((unsigned)grid), dim3(128), lds, st, q);
    return hipGetLastError();
}

} // namespace ofdm
