// kernels_sync.hip -- Schmidl-Cox sliding autocorrelation, reference CFO estimate, CFO derotation.
//
// k_sc_tile: one 256-thread workgroup per (frame, tile of CH = 2560 lags).  The tile's CH + W + L samples are
// staged once from HBM into LDS with coalesced interleaved-IQ loads (16 B per lane when aligned); every
// thread then owns C = 10 consecutive lags and advances the three sliding sums by the exact update
//     P(d+1) = P(d) + conj(r[d+W]) r[d+W+L] - conj(r[d]) r[d+L]     (same for E, R)
// A wavefront-shuffle scan over the per-thread totals (plus a 4-entry LDS step across the waves) turns the
// local prefixes into the sums at every lag.  All sums are f64: products of f32 samples are exact in f64, so
// the integer outputs (first threshold crossing, argmax) agree bit-for-bit with the f64 CPU oracle.
//
// Roofline: HBM.  Algorithmic traffic is 8 B per input sample + 16 B per frame of results; the halo of W + L
// samples per tile is re-read from L2.  ~45 f64-rate VALU ops per lag (MI355X: f64 vector = 1/2 f32 rate).
#include "device_common.hpp"
#include "kernels.hpp"
#include <limits.h>
#include <stdlib.h>

namespace ofdm {

constexpr int SC_C = 10;            // lags per thread
constexpr int SC_WG = 256;
constexpr int SC_CH = SC_C * SC_WG; // lags per tile

struct Sums { double pr, pi, e, r; };
__device__ __forceinline__ Sums s_add(Sums a, Sums b) { return Sums{a.pr + b.pr, a.pi + b.pi, a.e + b.e, a.r + b.r}; }
__device__ __forceinline__ Sums s_shfl_up(Sums a, int d) {
    return Sums{__shfl_up(a.pr, d, 64), __shfl_up(a.pi, d, 64), __shfl_up(a.e, d, 64), __shfl_up(a.r, d, 64)};
}
__device__ __forceinline__ Sums s_shfl_xor(Sums a, int d) {
    return Sums{__shfl_xor(a.pr, d, 64), __shfl_xor(a.pi, d, 64), __shfl_xor(a.e, d, 64), __shfl_xor(a.r, d, 64)};
}

struct Cand { double num, den, pr, pi; int lag; };
// a covers lower lags than b: b wins only if strictly greater (first maximum wins)
__device__ __forceinline__ Cand c_pick(Cand a, Cand b) { return (b.num * a.den > a.num * b.den) ? b : a; }
__device__ __forceinline__ Cand c_shfl_down(Cand a, int d) {
    return Cand{__shfl_down(a.num, d, 64), __shfl_down(a.den, d, 64), __shfl_down(a.pr, d, 64),
                __shfl_down(a.pi, d, 64), __shfl_down(a.lag, d, 64)};
}

__global__ __launch_bounds__(SC_WG) void k_sc_tile(ScParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int span = SC_CH + p.W + p.L;
    cf *raw = reinterpret_cast<cf *>(smem);
    Sums *wsum = reinterpret_cast<Sums *>(smem + (size_t)span * sizeof(cf)); // [4] wave totals
    Cand *wcand = reinterpret_cast<Cand *>(wsum + 4);                          // [4]
    int *wmin = reinterpret_cast<int *>(wcand + 4);                            // [4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = p.L, W = p.W;
    // slow-list mode: persistent workgroups redo the listed frames (tiles_per_frame tiles each: more than one only in mode 1, the
    // first-crossing pass of a search over more lags than a tile holds); otherwise one (frame, tile) per block
    const long long n_items = p.slow_list ? (long long)*p.slow_count * p.tiles_per_frame : 1;
  for (long long item = p.slow_list ? (long long)blockIdx.x : 0; item < n_items; item += p.slow_list ? (long long)gridDim.x : 1) {
    if (p.slow_list) __syncthreads();
    const long long unit = p.slow_list ? item : (long long)blockIdx.x;
    const long long fi = unit / p.tiles_per_frame;
    const long long f = p.slow_list ? (long long)p.slow_list[fi] : fi;
    const int tile = (int)(unit - fi * p.tiles_per_frame);

    long long d0;
    if (p.mode == 2) {
        int lb = p.lag_base[f];
        if (lb < 0) { // no crossing anywhere in this frame
            if (tid == 0) { p.d_hat[f] = -1; if (p.f_delta) p.f_delta[f] = 0.0; if (p.metric) p.metric[f] = 0.f; }
            continue;
        }
        d0 = lb;
    } else d0 = (long long)tile * SC_CH;
    long long nl = p.n_lags - d0;
    int n = (int)(nl < SC_CH ? nl : SC_CH); // lags handled here
    if (p.mode == 2 && n > W + 1) n = W + 1;

    // ---- stage CH + W + L samples (zero beyond the frame) : coalesced interleaved-IQ loads
    {
        const cf *src = p.in + f * p.frame_stride + d0;
        const long long avail = p.frame_len - d0; // > 0
        const bool aligned = ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
        if (aligned) {
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            float4 *r4 = reinterpret_cast<float4 *>(raw);
            for (int i = tid; i < span / 2; i += SC_WG) {
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (2 * i + 1 < avail) x = s4[i];
                else if (2 * i < avail) { cf y = src[2 * i]; x.x = y.x; x.y = y.y; }
                r4[i] = x;
            }
        } else {
            for (int i = tid; i < span; i += SC_WG) raw[i] = i < avail ? src[i] : make_float2(0.f, 0.f);
        }
    }
    __syncthreads();

    // ---- sums at the tile's first lag: WG reduction over the W products
    Sums x0 = Sums{0, 0, 0, 0};
    for (int m = tid; m < W; m += SC_WG) {
        cf a = raw[m], b = raw[m + L];
        double ar = a.x, ai = a.y, br = b.x, bi = b.y;
        x0.pr += ar * br + ai * bi;
        x0.pi += ar * bi - ai * br;
        x0.e += ar * ar + ai * ai;
        x0.r += br * br + bi * bi;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) x0 = s_add(x0, s_shfl_xor(x0, s));
    if (lane == 0) wsum[wave] = x0;
    __syncthreads();
    x0 = s_add(s_add(wsum[0], wsum[1]), s_add(wsum[2], wsum[3]));
    __syncthreads();

    // ---- per-thread updates for its C lags, local inclusive prefix
    const int a0 = tid * SC_C;
    Sums pre[SC_C];
    {
        Sums run = Sums{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < SC_C; ++j) {
            cf s0 = raw[a0 + j], s1 = raw[a0 + j + L], s2 = raw[a0 + j + W], s3 = raw[a0 + j + W + L];
            double r0 = s0.x, i0 = s0.y, r1 = s1.x, i1 = s1.y, r2 = s2.x, i2 = s2.y, r3 = s3.x, i3 = s3.y;
            double e0 = r0 * r0 + i0 * i0, e1 = r1 * r1 + i1 * i1, e2 = r2 * r2 + i2 * i2, e3 = r3 * r3 + i3 * i3;
            run.pr += (r2 * r3 + i2 * i3) - (r0 * r1 + i0 * i1);
            run.pi += (r2 * i3 - i2 * r3) - (r0 * i1 - i0 * r1);
            run.e += e2 - e0;
            run.r += e3 - e1;
            pre[j] = run;
        }
    }
    // ---- exclusive scan of the thread totals over the workgroup (wave shuffles + 4 wave totals in LDS)
    Sums inc = pre[SC_C - 1];
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        Sums o = s_shfl_up(inc, s);
        if (lane >= s) inc = s_add(inc, o);
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    Sums base = x0;
    for (int wv = 0; wv < wave; ++wv) base = s_add(base, wsum[wv]);
    {
        Sums excl = s_shfl_up(inc, 1);
        if (lane > 0) base = s_add(base, excl);
    }
    // base = sums at lag a0

    const double thr = p.threshold;
    // ---- (A) packet detect: first lag with M >= threshold
    int d1;
    if (p.mode == 2) d1 = 0;
    else {
        int mine = INT_MAX;
#pragma unroll
        for (int j = SC_C - 1; j >= 0; --j) {
            Sums x = j ? s_add(base, pre[j - 1]) : base;
            double num = x.pr * x.pr + x.pi * x.pi, den = x.e * x.r;
            if (a0 + j < n && den > 0.0 && num >= thr * den) mine = a0 + j;
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) { int o = __shfl_xor(mine, s, 64); mine = o < mine ? o : mine; }
        if (lane == 0) wmin[wave] = mine;
        __syncthreads();
        int m01 = wmin[0] < wmin[1] ? wmin[0] : wmin[1], m23 = wmin[2] < wmin[3] ? wmin[2] : wmin[3];
        d1 = m01 < m23 ? m01 : m23;
        if (p.mode == 1) {
            if (tid == 0) p.cross[f * p.tiles_per_frame + tile] = d1 == INT_MAX ? LLONG_MAX : d0 + d1;
            continue;
        }
        if (d1 == INT_MAX) {
            if (tid == 0) { p.d_hat[f] = -1; if (p.f_delta) p.f_delta[f] = 0.0; if (p.metric) p.metric[f] = 0.f; }
            continue;
        }
    }
    // ---- (B) first maximum of M over [d1, d1 + W]
    Cand best = Cand{-1.0, 1.0, 0.0, 0.0, -1};
#pragma unroll
    for (int j = 0; j < SC_C; ++j) {
        const int lag = a0 + j;
        Sums x = j ? s_add(base, pre[j - 1]) : base;
        double num = x.pr * x.pr + x.pi * x.pi, den = x.e * x.r;
        if (lag < n && lag >= d1 && lag <= d1 + W && den > 0.0) best = c_pick(best, Cand{num, den, x.pr, x.pi, lag});
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        Cand o = c_shfl_down(best, s);
        if (lane + s < 64) best = c_pick(best, o);
    }
    if (lane == 0) wcand[wave] = best;
    __syncthreads();
    if (tid == 0) {
        Cand b = c_pick(c_pick(wcand[0], wcand[1]), c_pick(wcand[2], wcand[3]));
        if (b.lag < 0) { p.d_hat[f] = -1; if (p.f_delta) p.f_delta[f] = 0.0; if (p.metric) p.metric[f] = 0.f; }
        else {
            p.d_hat[f] = (int32_t)(d0 + b.lag);
            if (p.f_delta) p.f_delta[f] = atan2(b.pi, b.pr) / (double)L;
            if (p.metric) p.metric[f] = (float)(b.num / b.den);
        }
    }
  } // item loop
}

size_t sc_lds_bytes(const ScParams &p) {
    return (size_t)(SC_CH + p.W + p.L) * sizeof(float2) + 4 * sizeof(Sums) + 4 * sizeof(Cand) + 4 * sizeof(int) + 64;
}

hipError_t run_sc(const ScParams &p, hipStream_t st) {
    if (p.n_frames <= 0) return hipSuccess;
    size_t lds = sc_lds_bytes(p);
    if (lds > 48 * 1024) { // per-device attribute; cheap enough to set on every launch that needs it
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_sc_tile),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    long long blocks = p.n_frames * (long long)p.tiles_per_frame;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    trace_add(p.trace, p.mode == 0 ? "k_sc_tile" : p.mode == 1 ? "k_sc_tile<cross>" : "k_sc_tile<peak>");
    hipLaunchKernelGGL(k_sc_tile, dim3((unsigned)blocks), dim3(SC_WG), lds, st, p);
    return hipGetLastError();
}
int sc_tile_lags() { return SC_CH; }


// ---------------------------------------------------------------------------------------------------------------
// Fast Schmidl-Cox path for one-tile frames with a short period (L = 80: N = 64) and 16-byte aligned frame bases:
// "filter in f32, decide in f64", and only look closely where a packet can be.
//   * every product q[n] = conj(r[n]) r[n+L] and energy e[n] = |r[n]|^2 is formed ONCE, in f32, and summed over
//     10-sample chunks; a DPP wavefront scan (row shifts / broadcasts) plus the wave totals give exclusive chunk
//     prefixes, so the sums at a chunk's first lag are prefix differences:
//         P = Bq[c + W/10] - Bq[c],   E = Be[c + W/10] - Be[c],   R = Be[c + (W+L)/10] - Be[c + L/10];
//   * a bound on the metric over each chunk's 10 lags discards the chunks that cannot reach the threshold (k_sc_cf);
//   * decisions are EXACT: the f32 metric (absolute error <= ~5e-6 x prefix-energy / window-energy) only filters.
//     The first crossing is accepted from f32 when "M >= thr(1-EPS)" and "M >= thr(1+EPS)" first hold at the same lag;
//     the peak is re-evaluated in f64 (products of f32 are exact in f64) at every lag within 2 EPS of the f32 window
//     maximum.  Lags whose error bound is not small (prefix energy > 20x window energy) are never trusted.  Frames the
//     filter cannot settle (ambiguous crossing, > 4 peak candidates) go to a device-side list and are redone by the
//     all-f64 kernel k_sc_tile.  Outputs equal the f64 oracle's.
constexpr float SC_EPS = 1e-3f;         // relative guard band of the f32 filter around the threshold / the maximum
constexpr float SC_UNSAFE_RATIO = 20.f; // prefix energy / window energy above which a lag is never trusted
constexpr float SC_PREFIX_ERR = 1.6e-5f; // bound on |f32 window sum - exact| / prefix magnitude when PROVING a lag below the threshold (the scans and slides stay below 4e-6; x 4 for margin)
constexpr int SC_MAXCAND = 4;

template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_d(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, true); // out-of-range / masked lanes read 0
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, true);
    return __hiloint2double(hi, lo);
}
// inclusive prefix sum over the 64 lanes of a wavefront (lane 63 ends up with the wave total)
__device__ __forceinline__ double wave_scan(double x) {
    x += dpp_d<0x111, 0xF>(x); // row_shr:1
    x += dpp_d<0x112, 0xF>(x); // row_shr:2
    x += dpp_d<0x114, 0xF>(x); // row_shr:4
    x += dpp_d<0x118, 0xF>(x); // row_shr:8
    x += dpp_d<0x142, 0xA>(x); // row_bcast:15 into rows 1 and 3
    x += dpp_d<0x143, 0xC>(x); // row_bcast:31 into rows 2 and 3
    return x;
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_s(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ float wave_scan_f(float x) {
    x += dpp_s<0x111, 0xF>(x);
    x += dpp_s<0x112, 0xF>(x);
    x += dpp_s<0x114, 0xF>(x);
    x += dpp_s<0x118, 0xF>(x);
    x += dpp_s<0x142, 0xA>(x);
    x += dpp_s<0x143, 0xC>(x);
    return x;
}
__device__ __forceinline__ float readlane_f(float x, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}
__device__ __forceinline__ double readlane_d(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

// Totals of FOUR f64 values over the wavefront, every lane receiving all four -- a transpose-reduce instead of four scans: the first
// two steps trade values between lane pairs (lane ^ 1: even lanes keep a and c, odd lanes b and d; lane ^ 2: one value per lane,
// quantity = lane & 3), two row rotations add the four lanes of a row that hold the same quantity, and gfx950's
// v_permlane16_swap / v_permlane32_swap add the rows.  Eight 64-bit additions and 14 lane movements instead of 24 and 48.
template <int CTRL> __device__ __forceinline__ double dpp_q(double x) { // quad / row permutation of a double, every lane valid
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rows_sum16(double x) { // [r0, r1, r2, r3] -> [r0 + r1, r0 + r1, r2 + r3, r2 + r3], lane for lane
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(x), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(x), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double halves_sum32(double x) { // [lower, upper] -> lower + upper in both halves, lane for lane
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(x), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(x), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void wave_sum4(double &a, double &b, double &c, double &d, int lane) {
    const bool o1 = (lane & 1) != 0, o2 = (lane & 2) != 0;
    double k0 = o1 ? b : a, k1 = o1 ? d : c;            // kept; the partner receives the other two
    k0 += dpp_q<0xB1>(o1 ? a : b);                       // quad_perm [1,0,3,2]: lane ^ 1
    k1 += dpp_q<0xB1>(o1 ? c : d);
    double k = o2 ? k1 : k0;
    k += dpp_q<0x4E>(o2 ? k0 : k1);                      // quad_perm [2,3,0,1]: lane ^ 2 -> quantity (lane & 3) summed over the quad
    k += dpp_q<0x124>(k);                                // row_ror:4
    k += dpp_q<0x128>(k);                                // row_ror:8 -> ... over the row
    k = halves_sum32(rows_sum16(k));                     // ... over the wavefront
    a = readlane_d(k, 0); b = readlane_d(k, 1); c = readlane_d(k, 2); d = readlane_d(k, 3);
}

struct ScFastParams {
    const float2 *in;
    long long n_frames, frame_stride;
    int n16;        // 16-byte pieces to stage per frame = min(10 NCH, frame_len) / 2
    int n_lags, L, W;
    int debug;      // profiling aid (Tuning::debug_sc, profile build only): 2 / 3 / 5 / 4 stop after phase 1 / coarse / slide / select; 10..17 section times
    float thr_lo, thr_hi;
    int32_t *d_hat;
    int32_t *slow_list; // frames the filter could not settle
    int32_t *slow_count;
    struct ScExact *exact; // exact sums at the chosen lag, per frame
    // Two-phase search (run_sc_fast): the first launch looks at the first n_lags lags only and puts every frame whose result those
    // lags do not DETERMINE -- no crossing among them, or a peak window that reaches beyond them -- on the redo list; the second
    // launch runs the whole search for the frames of that list (frame_list / frame_count).
    int defer;                    // 1: undetermined frames go to redo_list instead of getting a result
    int32_t *redo_list, *redo_count;
    const int32_t *frame_list;    // optional: only these frames (count on the device)
    const int32_t *frame_count;
};

// ---------------------------------------------------------------------------------------------------------------
// k_sc_cf: coarse-then-fine filter.  The per-lag metric is only ever needed around the packet: at the first threshold
// crossing and over the W + 1 lags after it.  One workgroup per frame (persistent over the frame list), the frame
// staged once from HBM into LDS by LDS-DMA (global_load_lds_dwordx4), then
//   phase 1  (all threads)  chunk totals of q = conj(r[n]) r[n+L] and e = |r[n]|^2 over 10-sample chunks, one
//            workgroup scan -> exclusive chunk prefixes Bq / Be and the chunk energies Te in LDS (4 KB);
//   coarse   (all threads)  P, E, R at the chunk's first lag are prefix differences.  Over the chunk's 10 lags
//                |P(d)| <= |P0| + 1/2 (Te[c] + Te[c+L] + Te[c+W] + Te[c+W+L])      (|q[n]| <= (e[n] + e[n+L]) / 2)
//                E(d) >= E0 - Te[c],   R(d) >= R0 - Te[c+L]
//            so M(d) <= ub^2 / (Elo Rlo): chunks whose bound stays below the threshold (with the f32 error of the
//            prefix differences, <= 4e-6 x the largest prefix, charged against them) CANNOT hold a crossing and are
//            never looked at again.  On noise and on data symbols the bound is ~0.03, so only the chunks where the
//            periodic header enters the window are flagged;
//   fine     (ONE wavefront, rotating per frame so the SIMDs share the work)  the 32 chunks (320 lags) from the first
//            flagged one on: each lane slides the window over 5 lags from a chunk boundary, recomputing the entering and
//            leaving products from the raw samples in LDS; first crossing by ballot, window maximum by a DPP max-scan,
//            <= 4 candidates re-evaluated in f64 from the same LDS samples.  The exact sums go to k_sc_post, which
//            turns them into CFO and metric one LANE per frame (an f64 atan2 costs the same for 64 frames as for one).
// Anything the filter cannot settle (ambiguous crossing, > 4 candidates) goes to the slow list.
// LDS: raw samples (18 KB for a 2176-sample frame) + 4.3 KB -> seven workgroups per CU.  Measured (OFDM_SC_DEBUG=10..17,
// tools/lab/sc_sections.py): per frame the DMA wait, phase 1, bases, coarse and fine sections take about 1.0 / 2.0 / 0.7 /
// 1.3 / 6.0 thousand s_memtime ticks; the fine wavefront is the critical path, which is why its LDS reads are issued
// up front (one round trip per stage) and its reductions never touch LDS.
//
// maximum of non-negative values over the wavefront, DPP only (no LDS round trips): a max-scan leaves it in lane 63
__device__ __forceinline__ float wave_max_nonneg(float x) {
    x = fmaxf(x, dpp_s<0x111, 0xF>(x));
    x = fmaxf(x, dpp_s<0x112, 0xF>(x));
    x = fmaxf(x, dpp_s<0x114, 0xF>(x));
    x = fmaxf(x, dpp_s<0x118, 0xF>(x));
    x = fmaxf(x, dpp_s<0x142, 0xA>(x));
    x = fmaxf(x, dpp_s<0x143, 0xC>(x));
    return readlane_f(x, 63);
}

// Exact decision among <= 4 candidate lags of one frame (one wavefront; samples in LDS): f64 sums over the window,
// first maximum wins.  All of a candidate's LDS reads are issued together and the four sums are reduced with DPP
// scans, so a candidate costs one LDS round trip.  The result is wave-uniform (lag == INT_MAX: none).
__device__ __forceinline__ Cand sc_exact_pick(const cf *raw, int c0, int c1, int c2, int c3, int cnt, int L, int W, int lane) {
    Cand best = Cand{-1.0, 1.0, 0.0, 0.0, INT_MAX};
    for (int i = 0; i < cnt; ++i) {
        const int d = i == 0 ? c0 : i == 1 ? c1 : i == 2 ? c2 : c3;
        cf sa[4], sb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { // W <= 256 in one go (lanes past the window read sample d and are zeroed)
            const int m = lane + 64 * t < W ? lane + 64 * t : 0;
            sa[t] = raw[d + m]; sb[t] = raw[d + m + L];
        }
        double xr = 0, xi = 0, xe = 0, xq = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool v = lane + 64 * t < W;
            const double ar = v ? sa[t].x : 0.f, ai = v ? sa[t].y : 0.f, br = v ? sb[t].x : 0.f, bi = v ? sb[t].y : 0.f;
            xr += ar * br + ai * bi;
            xi += ar * bi - ai * br;
            xe += ar * ar + ai * ai;
            xq += br * br + bi * bi;
        }
        for (int m = lane + 256; m < W; m += 64) {
            const cf a = raw[d + m], b = raw[d + m + L];
            const double ar = a.x, ai = a.y, br = b.x, bi = b.y;
            xr += ar * br + ai * bi;
            xi += ar * bi - ai * br;
            xe += ar * ar + ai * ai;
            xq += br * br + bi * bi;
        }
        wave_sum4(xr, xi, xe, xq, lane);
        const double xn = xr * xr + xi * xi, xd = xe * xq;
        if (xd > 0.0) { // first maximum wins: strictly greater replaces, ties go to the lower lag
            const double lhs = xn * best.den, rhs = best.num * xd;
            if (lhs > rhs || (lhs == rhs && d < best.lag)) best = Cand{xn, xd, xr, xi, d};
        }
    }
    return best;
}
// CFO and metric of every frame the filter settled, one LANE per frame (the f64 atan2 and division cost the same for
// 64 frames as for one).  Frames without a packet and frames on the slow list (rewritten by k_sc_tile) get zeros.
__global__ __launch_bounds__(256) void k_sc_post(const int32_t *d_hat, const ScExact *ex, long long n_frames, int L, double *f_delta,
                                                 float *metric) {
    const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
    if (f >= n_frames) return;
    const bool found = d_hat[f] >= 0;
    ScExact e = ScExact{1.0, 0.0, 0.0, 1.0};
    if (found) e = ex[f];
    if (f_delta) f_delta[f] = found ? atan2(e.pi, e.pr) / (double)L : 0.0;
    if (metric) metric[f] = found ? (float)(e.num / e.den) : 0.f;
}

// NCH chunks per frame (128 / 256), CPT chunks per thread: thread t owns chunks t, t + WG, ... so that every
// (wavefront, u) pair is a run of 64 consecutive chunks -- a "virtual wavefront" for the scan and the flag masks.
// Fewer, fatter threads leave fewer wavefronts idle while one of them does the fine pass.
template <int NCH, int CPT, int OCC>
__global__ __launch_bounds__(NCH / CPT, OCC) void k_sc_cf(ScFastParams p) {
    constexpr int C = 10, WG = NCH / CPT, NW = WG / 64, VW = NCH / 64;
    extern __shared__ __align__(16) unsigned char smem[];
    const int L = p.L, W = p.W, n = p.n_lags;
    const int dbg = kProfile ? p.debug : 0; // the early exits / section ticks exist in the profile build only
    const int nstaged = 2 * p.n16;
    const int ns = (nstaged + C - 1) / C * C;               // whole 10-sample chunks
    cf *raw = reinterpret_cast<cf *>(smem);                 // [ns + L]; entries past the staged samples stay 0
    float2 *bq = reinterpret_cast<float2 *>(raw + ns + L);  // [NCH + 2] exclusive chunk prefix of q ([NCH] = total)
    float *be = reinterpret_cast<float *>(bq + NCH + 2);    // [NCH + 2] ... of e
    float *tes = be + NCH + 2;                              // [NCH] chunk energies
    float *wtot = tes + NCH;                                // [VW][4] virtual-wave totals
    unsigned long long *flg = reinterpret_cast<unsigned long long *>(wtot + 4 * VW); // [VW] flagged chunks

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cL = L / C, cW = W / C;
    const unsigned raw_lds = lds_addr(raw);
    const float thr_c = p.thr_lo * (1.f - SC_EPS);

    auto stage = [&](long long fr) { // this wave's 1-KiB pieces of frame fr -> raw
        const char *sbase = reinterpret_cast<const char *>(p.in + fr * p.frame_stride);
        for (int piece = wave; piece * 64 < p.n16; piece += NW) {
            const int i = piece * 64 + lane;
            if (i < p.n16) glds16(sbase, (unsigned)i * 16u, raw_lds + (unsigned)piece * 1024u);
        }
    };
    auto to_slow = [&](long long fr) { p.d_hat[fr] = -1; p.slow_list[atomicAdd(p.slow_count, 1)] = (int32_t)fr; };

    for (int i = nstaged + tid; i < ns + L; i += WG) raw[i] = make_float2(0.f, 0.f); // never written by the DMA
    if (tid == 0) { bq[NCH] = make_float2(0.f, 0.f); be[NCH] = 0.f; } // chunk NCH: local prefix 0 of the wavefront after the last
    // Frames the first lags do not determine go to the device-side redo list in BATCHES: with one atomicAdd per frame on the one
    // counter, a batch in which every frame defers (late packets, empty slots) spent 1.5 ms per 131 072 frames waiting for that
    // address; a workgroup now collects up to 16 of its frames in LDS and reserves their list slots with one atomic.  Writers are
    // thread 0 (no flagged chunk) or lane 0 of the fine wavefront, in different phases of a frame, always barriers apart.
    constexpr int RB = 16;
    int *rbuf = reinterpret_cast<int *>(flg + VW);          // [0] pending, [1 .. RB] frames
    auto redo_flush = [&]() {
        const int m = rbuf[0];
        if (m > 0) {
            const int base = atomicAdd(p.redo_count, m);
            for (int i = 0; i < m; ++i) p.redo_list[base + i] = rbuf[1 + i];
            rbuf[0] = 0;
        }
    };
    auto to_redo = [&](long long fr) {
        const int m = rbuf[0];
        rbuf[1 + m] = (int32_t)fr;
        rbuf[0] = m + 1;
        if (m + 1 == RB) redo_flush();
    };
    if (tid == 0) rbuf[0] = 0;
    const bool listed = p.frame_list != nullptr;
    const long long n_items = listed ? (long long)*p.frame_count : p.n_frames;
    auto frame_of = [&](long long item) -> long long { return listed ? (long long)p.frame_list[item] : item; };
    long long item = blockIdx.x;
    const long long fstep = gridDim.x;
    if (item < n_items) stage(frame_of(item));
    int it = blockIdx.x;

    for (; item < n_items; item += fstep, ++it) {
        const long long f = frame_of(item);
        const long long t0 = (dbg >= 10) ? (long long)__builtin_amdgcn_s_memtime() : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's pieces of the CURRENT frame have landed
        lds_barrier();                                   // B0: ... and everyone else's
        const long long t1 = (dbg >= 10) ? (long long)__builtin_amdgcn_s_memtime() : 0;
        const bool more = item + fstep < n_items;
        const long long f_next = more ? frame_of(item + fstep) : 0;
        // ---- phase 1 (f32): chunk totals of q and e; exclusive prefixes WITHIN each virtual wavefront (64 chunks) go to
        //      LDS together with the wavefront totals.  A prefix difference that crosses into the next virtual wavefront
        //      just adds the first one's total, so no second scan across wavefronts (and no barrier for it) is needed.
        float te[CPT], lqr[CPT], lqi[CPT], le[CPT], Tqr[CPT], Tqi[CPT], Te[CPT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int vt = u * WG + tid, n0 = vt * C;
            float tqr = 0.f, tqi = 0.f;
            te[u] = 0.f;
            if (n0 < ns) {
                const float4 *pa = reinterpret_cast<const float4 *>(raw + n0), *pb = reinterpret_cast<const float4 *>(raw + n0 + L);
#pragma unroll
                for (int i = 0; i < C / 2; ++i) {
                    const float4 x = pa[i], y = pb[i];
                    tqr += x.x * y.x + x.y * y.y; tqi += x.x * y.y - x.y * y.x; te[u] += x.x * x.x + x.y * x.y;
                    tqr += x.z * y.z + x.w * y.w; tqi += x.z * y.w - x.w * y.z; te[u] += x.z * x.z + x.w * x.w;
                }
            }
            const float iqr = wave_scan_f(tqr), iqi = wave_scan_f(tqi), ie = wave_scan_f(te[u]);
            lqr[u] = iqr - tqr; lqi[u] = iqi - tqi; le[u] = ie - te[u];
            Tqr[u] = readlane_f(iqr, 63); Tqi[u] = readlane_f(iqi, 63); Te[u] = readlane_f(ie, 63);
            tes[vt] = te[u];
            bq[vt] = make_float2(lqr[u], lqi[u]);
            be[vt] = le[u];
            if (lane == 63) { const int vw = u * NW + wave; wtot[vw * 4 + 0] = iqr; wtot[vw * 4 + 1] = iqi; wtot[vw * 4 + 2] = ie; }
        }
        lds_barrier(); // B3: chunk prefixes, energies and wavefront totals visible
        const long long t2 = t1, t3 = (dbg >= 10) ? (long long)__builtin_amdgcn_s_memtime() : 0;
        if (dbg == 2) { lds_barrier(); if (more) stage(f_next); continue; }
        float etot = 0.f; // frame energy: bounds every prefix (error margin of the coarse bound)
#pragma unroll
        for (int wv = 0; wv < VW; ++wv) etot += wtot[wv * 4 + 2];

        // ---- coarse pass: can any of a chunk's 10 lags reach the threshold?
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int vt = u * WG + tid;
            bool flag = false;
            if (vt * C < n) { // owns at least one searched lag (implies vt + cW + cL < NCH)
                const bool xW = lane + cW >= 64, xL = lane + cL >= 64, xWL = lane + cW + cL >= 64; // leaves this virtual wavefront
                const float2 b1 = bq[vt + cW];
                const float Pr = (b1.x - lqr[u]) + (xW ? Tqr[u] : 0.f), Pi = (b1.y - lqi[u]) + (xW ? Tqi[u] : 0.f);
                const float eL = tes[vt + cL], eW = tes[vt + cW], eWL = tes[vt + cW + cL];
                const float E0 = (be[vt + cW] - le[u]) + (xW ? Te[u] : 0.f);
                const float R0 = (be[vt + cW + cL] - be[vt + cL]) + ((xWL ? Te[u] : 0.f) - (xL ? Te[u] : 0.f));
                const float dlt = etot * 4e-6f;                                // f32 error of a prefix difference
                const float ub = __builtin_sqrtf(Pr * Pr + Pi * Pi) * 1.000001f + 0.5f * (te[u] + eL + eW + eWL) + 2.f * dlt;
                const float Elo = E0 - te[u] - dlt, Rlo = R0 - eL - dlt;
                flag = ub > 0.f && !(Elo > 0.f && Rlo > 0.f && ub * ub < thr_c * Elo * Rlo);
            }
            const unsigned long long fm = __ballot(flag);
            if (lane == 0) flg[u * NW + wave] = fm;
        }
        lds_barrier(); // B4: flags visible
        const long long t4 = (dbg >= 10) ? (long long)__builtin_amdgcn_s_memtime() : 0;
        unsigned long long fmask[VW];
#pragma unroll
        for (int wv = 0; wv < VW; ++wv) {
            const unsigned long long v = flg[wv];
            fmask[wv] = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                        (unsigned)__builtin_amdgcn_readfirstlane((int)v);
        }
        auto next_flag = [&](int pos) -> int { // first flagged chunk >= pos, or -1 (wave-uniform)
#pragma unroll
            for (int wv = 0; wv < VW; ++wv) {
                if (pos >= (wv + 1) * 64) continue;
                unsigned long long mm = fmask[wv];
                if (pos > wv * 64) mm &= ~0ull << (pos - wv * 64);
                if (mm) return wv * 64 + __ffsll((long long)mm) - 1;
            }
            return -1;
        };
        int gs = next_flag(0);
        if (gs < 0) { // no chunk can reach the threshold: no packet
            if (tid == 0) {
                if (p.defer) to_redo(f);   // nothing in the first lags: the whole search has to look
                else p.d_hat[f] = -1;
            }
            if (more) stage(f_next);   // every wave is done with the raw samples (phase 1 ended before B3)
            continue;
        }
        if (dbg == 3) { lds_barrier(); if (more) stage(f_next); continue; }

        // ---- fine pass: one wavefront, the 32 chunks (320 lags) from the first flagged one; lane 2k slides forward from
        //      chunk boundary k over lags +0..+4, lane 2k+1 slides BACKWARD from boundary k+1 over lags +9..+5
        if (wave == (it & (NW - 1))) {
            const int hh = lane & 1;
            const float sgn = hh ? -1.f : 1.f;
            long long u0 = (dbg >= 15) ? (long long)__builtin_amdgcn_s_memtime() : 0, u1 = 0, u2 = 0;
            int res_d = -1;          // wave-uniform outcome: lag >= 0, -1 no packet, -2 slow list, -3 debug exit, -4 redo list
            Cand best = Cand{0.0, 1.0, 1.0, 0.0, INT_MAX};
            for (;;) {
                const int c = gs + (lane >> 1), m0 = c * C;
                const bool lv = m0 + 5 * hh < n; // implies c + hh + cW + cL <= NCH
                float mp[5];                      // metric of round k: lag m0 + k (forward) or m0 + 9 - k (backward)
                bool unsafe_t = false;
                int lo = INT_MAX, hi = INT_MAX;
#pragma unroll
                for (int k = 0; k < 5; ++k) mp[k] = -1.f;
                if (lv) {
                    const int cb = c + hh;
                    // every LDS read of the slide up front: one round trip
                    const float2 q0 = bq[cb], q1 = bq[cb + cW];
                    const float e0 = be[cb], e1 = be[cb + cW], e2 = be[cb + cL], e3 = be[cb + cW + cL];
                    // wave-local prefixes: a difference that leaves its virtual wavefront adds that wavefront's total
                    const int v0 = cb >> 6, v2 = (cb + cL) >> 6;
                    const bool xP = ((cb + cW) >> 6) != v0, xR = ((cb + cW + cL) >> 6) != v2;
                    const float t0r = wtot[v0 * 4 + 0], t0i = wtot[v0 * 4 + 1], t0e = wtot[v0 * 4 + 2], t2e = wtot[v2 * 4 + 2];
                    // magnitude of the prefixes behind each sum (what its f32 cancellation error scales with): a wave-local prefix holds
                    // the energy from its virtual wavefront's first chunk up to the entry, plus that wavefront's total when the difference
                    // leaves it.  (Round 3 charged e3 + t0e + t2e against every sum: in front of a LATE packet, where the window is noise
                    // and the totals are packet energy, every lag then looked untrusted and the whole frame went to the all-f64 kernel.)
                    const float magE = fmaxf(e0, e1) + (xP ? t0e : 0.f), magR = fmaxf(e2, e3) + (xR ? t2e : 0.f);
                    const float magP = fmaxf(magE, magR) + (v2 != v0 ? t0e : 0.f); // |q[n]| <= (e[n] + e[n+L]) / 2
                    cf sa[5], sb[5], sc[5], sd[5];
#pragma unroll
                    for (int k = 0; k < 5; ++k) { // low sample x of the step to round k's lag (forward k = 0 sits on the boundary: no step)
                        const int x = hh ? m0 + 9 - k : m0 + (k > 0 ? k - 1 : 0);
                        sa[k] = raw[x]; sb[k] = raw[x + L]; sc[k] = raw[x + W]; sd[k] = raw[x + W + L];
                    }
                    float Pr = (q1.x - q0.x) + (xP ? t0r : 0.f), Pi = (q1.y - q0.y) + (xP ? t0i : 0.f);
                    float E = (e1 - e0) + (xP ? t0e : 0.f), R = (e3 - e2) + (xR ? t2e : 0.f);
                    float emE = 3.0e38f, emR = 3.0e38f, mmax = -1.f;
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        const float wk = (hh || k > 0) ? sgn : 0.f;
                        const cf a = sa[k], b = sb[k], cc = sc[k], d = sd[k];
                        Pr += wk * (cc.x * d.x + cc.y * d.y - a.x * b.x - a.y * b.y);
                        Pi += wk * (cc.x * d.y - cc.y * d.x - a.x * b.y + a.y * b.x);
                        E += wk * (cc.x * cc.x + cc.y * cc.y - a.x * a.x - a.y * a.y);
                        R += wk * (d.x * d.x + d.y * d.y - b.x * b.x - b.y * b.y);
                        const int lag = hh ? m0 + 9 - k : m0 + k;
                        const float den = E * R;
                        const bool ok = den > 0.f && lag < n;
                        mp[k] = ok ? (Pr * Pr + Pi * Pi) * __builtin_amdgcn_rcpf(den) : -1.f;
                        emE = fminf(emE, ok ? E : 3.0e38f);
                        emR = fminf(emR, ok ? R : 3.0e38f);
                        mmax = fmaxf(mmax, mp[k]);
                    }
                    // trusted to within SC_EPS: every sum's error (<= 4e-6 x its prefix magnitude) is below 1e-4 of the sum itself --
                    // E and R against themselves, |P| against sqrt(E R) (M near the threshold means |P| ~ 0.7 sqrt(E R))
                    const float gmin = __builtin_amdgcn_sqrtf(emE) * __builtin_amdgcn_sqrtf(emR); // the minima stay huge when no lag has energy
                    unsafe_t = magE > SC_UNSAFE_RATIO * emE || magR > SC_UNSAFE_RATIO * emR || magP > SC_UNSAFE_RATIO * gmin;
                    // An untrusted lag is not yet a possible crossing: sums too coarse to place M within SC_EPS can still prove it far
                    // below the threshold.  With every sum off by at most SC_PREFIX_ERR x its prefix magnitude,
                    //     sqrt(M) <= (sqrt(M_f32) + 1.5 dP / sqrt(E R)) / sqrt((1 - dE / E)(1 - dR / R))
                    // over the lane's lags (worst E, R and M of the five); lags that stay below the threshold under it cannot cross.
                    bool maybe = unsafe_t;
                    if (unsafe_t && mmax >= 0.f) {
                        const float erE = SC_PREFIX_ERR * magE * __builtin_amdgcn_rcpf(emE), erR = SC_PREFIX_ERR * magR * __builtin_amdgcn_rcpf(emR);
                        const float up = __builtin_amdgcn_sqrtf(mmax) + 1.5f * SC_PREFIX_ERR * magP * __builtin_amdgcn_rcpf(gmin);
                        maybe = !(erE < 0.25f && erR < 0.25f && up * up < p.thr_lo * (1.f - erE) * (1.f - erR));
                    }
                    if (maybe || mmax >= p.thr_lo) {
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            const int kk = hh ? k : 4 - k;           // visit this lane's lags in decreasing order
                            const int lag = hh ? m0 + 9 - kk : m0 + kk;
                            const float m = mp[kk];
                            if (m >= 0.f && (maybe || m >= p.thr_lo)) lo = lag;
                            if (m >= 0.f && !unsafe_t && m >= p.thr_hi) hi = lag;
                        }
                    }
                }
                if (dbg == 5) { res_d = -3; break; }
                if (dbg >= 15) u1 = (long long)__builtin_amdgcn_s_memtime();
                const unsigned long long lo_m = __ballot(lo != INT_MAX), hi_m = __ballot(hi != INT_MAX);
                if (lo_m == 0) { // nothing crosses in these 32 chunks: on to the next flagged chunk
                    gs = next_flag(gs + 32);
                    if (gs < 0) { res_d = -1; break; }
                    continue;
                }
                const int c_lo = __builtin_amdgcn_readlane(lo, __ffsll((long long)lo_m) - 1);
                const int c_hi = hi_m ? __builtin_amdgcn_readlane(hi, __ffsll((long long)hi_m) - 1) : INT_MAX;
                if (c_lo != c_hi) { res_d = -2; break; }                  // ambiguous crossing: redo in f64
                const int d1 = c_lo;
                if (p.defer && d1 + W > n - 1) { res_d = -4; break; } // the peak window reaches beyond the first lags: whole search
                const int dend = d1 + W < n - 1 ? d1 + W : n - 1;        // last lag of the peak window
                if (dend >= (gs + 32) * C) { gs = d1 / C; continue; }    // window not covered: restart at the crossing's chunk
                // window maximum over the trusted lags, then the candidates within 2 EPS of it (+ the untrusted ones)
                float lmax = 0.f;
                bool inw[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const int lag = hh ? m0 + 9 - k : m0 + k;
                    inw[k] = lag >= d1 && lag <= dend && mp[k] >= 0.f;
                    if (inw[k] && !unsafe_t) lmax = fmaxf(lmax, mp[k]);
                }
                const float mcut = wave_max_nonneg(lmax) * (1.f - 2.f * SC_EPS);
                unsigned cm = 0; // this lane's candidate rounds
#pragma unroll
                for (int k = 0; k < 5; ++k) cm |= (inw[k] && (unsafe_t || mp[k] >= mcut)) ? 1u << k : 0u;
                int cand[SC_MAXCAND] = {0, 0, 0, 0}; // only ever indexed statically
                int cnt = 0;
                unsigned long long any = __ballot(cm != 0);
                while (any && cnt <= SC_MAXCAND) { // wave-uniform: one candidate per trip (typically one trip)
                    const int l = __ffsll((long long)any) - 1;
                    const unsigned mk = (unsigned)__builtin_amdgcn_readlane((int)cm, l);
                    const int k = __ffs((int)mk) - 1;
                    const int lag = (gs + (l >> 1)) * C + ((l & 1) ? 9 - k : k);
#pragma unroll
                    for (int q = 0; q < SC_MAXCAND; ++q) cand[q] = cnt == q ? lag : cand[q];
                    ++cnt;
                    if (lane == l) cm &= cm - 1;
                    any = __ballot(cm != 0);
                }
                if (cnt == 0) { res_d = -1; break; }
                if (cnt > SC_MAXCAND) { res_d = -2; break; }
                if (dbg == 4) { res_d = -3; break; }
                if (dbg >= 15) u2 = (long long)__builtin_amdgcn_s_memtime();
                best = sc_exact_pick(raw, cand[0], cand[1], cand[2], cand[3], cnt, L, W, lane);
                res_d = best.lag != INT_MAX ? best.lag : -1;
                break;
            }
            if (lane == 0 && res_d != -3) {
                if (res_d == -2) to_slow(f);
                else if (p.defer && (res_d == -4 || res_d == -1)) to_redo(f); // undetermined by the first lags
                else {
                    p.d_hat[f] = res_d;
                    if (res_d >= 0) p.exact[f] = ScExact{best.pr, best.pi, best.num, best.den};
                    else p.exact[f] = ScExact{0.0, 0.0, 0.0, 1.0};
                }
            }
            if (dbg >= 15 && lane == 0) { // profiling aid: slide / select / exact time of the fine wavefront
                const long long u3 = (long long)__builtin_amdgcn_s_memtime();
                p.d_hat[f] = (int32_t)(dbg == 15 ? u1 - u0 : dbg == 16 ? u2 - u1 : u3 - u2);
            }
        }
        lds_barrier(); // B5: the fine wavefront is done with the raw samples
        if (dbg >= 10 && dbg < 15 && tid == 0) { // profiling aid: per-frame section time (s_memtime ticks) instead of the timing result
            const long long t5 = (long long)__builtin_amdgcn_s_memtime();
            const long long dt[5] = {t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4};
            p.d_hat[f] = (int32_t)dt[dbg - 10 < 5 ? dbg - 10 : 4];
        }
        if (more) stage(f_next);
        continue;
    }
    if (p.defer) { // the frames still waiting in this workgroup's LDS batch
        lds_barrier();
        if (tid == 0) redo_flush();
    }
}
static size_t sc_cf_lds_bytes(int L, int nch, long long frame_len) { // raw samples + chunk prefixes, energies, flags
    long long ns = ((frame_len + 1) / 2 * 2 + 9) / 10 * 10;
    if (ns > (long long)nch * 10) ns = (long long)nch * 10;
    return (size_t)(ns + L) * sizeof(float2) + (size_t)(nch + 2) * (sizeof(float2) + sizeof(float)) + (size_t)nch * sizeof(float) +
           16 * sizeof(float) + 64 + 16 + 80 /* the redo batch: count + 16 frames */;
}

// chunks per frame (128 / 256, 10 samples each): the smallest tile that covers the searched lags plus the window
static int sc_fast_pick_nch(const ScParams &p) {
    for (int nch = 128; nch <= 256; nch *= 2)
        if ((long long)nch * 10 - p.W - p.L >= p.n_lags) return nch;
    return 0;
}
bool sc_fast_ok(const ScParams &p) {
    // one tile per frame, 10 | L, 10 | W, 16-byte aligned even-length frames, and a peak window (W + 1 lags) that the
    // fine pass (320 lags from the first flagged chunk) can cover
    return p.mode == 0 && sc_fast_pick_nch(p) != 0 && p.L % 10 == 0 && p.W % 10 == 0 && p.W + 20 <= 320 &&
           (reinterpret_cast<uintptr_t>(p.in) & 15) == 0 && (p.frame_stride & 1) == 0 && (p.frame_len & 1) == 0;
}
// earliest crossing per frame over the tiles of a multi-tile search, for a device-side list of frames
__global__ void k_sc_min_cross_list(const long long *cross, int tiles, const int32_t *list, const int32_t *count, int32_t *d1) {
    const long long n = *count;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long f = list[i];
        long long m = LLONG_MAX;
        for (int t = 0; t < tiles; ++t) { long long c = cross[f * tiles + t]; m = c < m ? c : m; }
        d1[f] = m == LLONG_MAX ? -1 : (int32_t)m;
    }
}
// exact sums + slow list + redo list (+ their counters); a search over more lags than one k_sc_tile tile holds (k_sc80 on long slots)
// also keeps the first crossing per (frame, tile) and per frame for the slow list's three-launch redo
static long long sc_fast_tiles(long long n_lags) { return (n_lags + SC_CH - 1) / SC_CH; }
size_t sc_fast_workspace_bytes(long long n_frames, long long n_lags) {
    const long long tiles = sc_fast_tiles(n_lags);
    size_t b = (size_t)n_frames * (sizeof(ScExact) + 2 * sizeof(int32_t)) + 128;
    if (tiles > 1) b += (size_t)n_frames * (size_t)tiles * sizeof(long long) + (size_t)n_frames * sizeof(int32_t) + 64;
    return b;
}

// p.mode == 0, p.tiles_per_frame == 1.  workspace: sc_fast_workspace_bytes(n_frames, n_lags) bytes of device memory.
// Callers: sc_fast_ok(p) (the filter pair can take the frame) or sc80_wanted(p) (k_sc80: any slot length).
hipError_t run_sc_fast(const ScParams &p, void *workspace, int num_cu, hipStream_t st) {
    if (p.n_frames <= 0) return hipSuccess;
    const Tuning &tu = tuning_or_default(p.tune);
    const bool use80 = !tu.no_sc80 && sc80_wanted(p);
    const int nch = sc_fast_pick_nch(p);
    if (!use80 && nch == 0) return hipErrorNotSupported;
    ScExact *exact = reinterpret_cast<ScExact *>(workspace);
    int32_t *slow_count = reinterpret_cast<int32_t *>(exact + p.n_frames);
    int32_t *slow_list = slow_count + 4;
    int32_t *redo_count = slow_list + p.n_frames;
    int32_t *redo_list = redo_count + 4;
    bool used_redo = false;
    hipError_t e = hipMemsetAsync(slow_count, 0, 16, st);
    if (e != hipSuccess) return e;
    ScFastParams q;
    q.defer = 0; q.redo_list = redo_list; q.redo_count = redo_count; q.frame_list = nullptr; q.frame_count = nullptr;
    q.in = p.in; q.n_frames = p.n_frames; q.frame_stride = p.frame_stride;
    const long long tile_n = (long long)(nch ? nch : 128) * 10;
    long long stage = p.frame_len < tile_n ? p.frame_len : tile_n;
    const long long needed = (p.n_lags + p.W + p.L + 10 + 1) & ~1LL; // the last searched lag's window, plus the slide's reach
    if (needed < stage) stage = needed;                        // bounded searches stage (and sum) only what they use
    q.n16 = (int)(stage / 2);
    q.n_lags = (int)p.n_lags; q.L = p.L; q.W = p.W;
    q.debug = kProfile ? tu.debug_sc : 0;
    q.thr_lo = (float)(p.threshold * (1.0 - (double)SC_EPS));
    q.thr_hi = (float)(p.threshold * (1.0 + (double)SC_EPS));
    q.d_hat = p.d_hat; q.slow_list = slow_list; q.slow_count = slow_count; q.exact = exact;
    // persistent over the frames; workgroups per CU bounded by LDS (22.5 KB for a 2176-sample frame -> 7)
    const int per_cu_cap = tu.sc_wg_per_cu > 0 ? tu.sc_wg_per_cu : 7; // tuning knob
    const size_t lds = sc_cf_lds_bytes(p.L, nch ? nch : 128, stage);
    long long per_cu = (long long)(160 * 1024) / (long long)lds;
    if (per_cu > per_cu_cap) per_cu = per_cu_cap;
    if (per_cu < 1) per_cu = 1;
    long long grid = (long long)num_cu * per_cu;
    const long long gcap = tu.grid_cap > 0 ? tu.grid_cap : (1LL << 40);
    if (grid > gcap) grid = gcap;
    if (grid > p.n_frames) grid = p.n_frames;
    if (use80) {
        // N = 64 (L = 80, W = 240): every lag exactly, one streaming pass that stops when the peak window has closed (kernels_sc80.hip)
        if ((e = launch_sc80(p, exact, slow_list, slow_count, num_cu, st)) != hipSuccess) return e;
    } else if (nch == 256) {
        // Two-phase search.  The detector is threshold-then-peak: once the first crossing d1 is known nothing after lag d1 + W can
        // change the answer, so the first `first` lags DETERMINE the result of every frame whose crossing and whole peak window lie
        // among them (a packet near the start of its slot: the usual case).  Launch 1 searches those lags only -- it stages and sums
        // first + W + L samples instead of the slot -- and lists the frames they do not determine (no crossing there, or a window
        // that reaches beyond); launch 2 is the whole search over that list.  Results are those of the whole search on every frame.
        const int first = tu.sc_first_lags;
        // (the first launch's tile is 128 chunks = 1280 samples: first + W + L + 11 samples must fit it, or the kernel's arrays behind the
        // samples would leave the LDS allocation)
        bool two_phase = first > 0 && p.n_lags >= 2LL * first && (long long)128 * 10 - p.W - p.L - 12 >= first && first > p.W + 1;
        if (two_phase) {
            if ((e = hipMemsetAsync(redo_count, 0, 16, st)) != hipSuccess) return e;
            used_redo = true;
            ScFastParams q1 = q;
            q1.n_lags = first; q1.defer = 1;
            long long stage1 = (first + p.W + p.L + 10 + 1) & ~1LL;
            if (stage1 > p.frame_len) stage1 = p.frame_len & ~1LL;
            if (stage1 > 1280) stage1 = 1280;
            q1.n16 = (int)(stage1 / 2);
            const size_t lds1 = sc_cf_lds_bytes(p.L, 128, stage1);
            long long pc = (long long)(160 * 1024) / (long long)lds1;
            if (pc > 10) pc = 10;
            long long g1 = (long long)num_cu * pc;
            if (g1 > gcap) g1 = gcap;
            if (g1 > p.n_frames) g1 = p.n_frames;
            trace_add(p.trace, "k_sc_cf<128,first>");
            if (tu.sc128_one_wave) { // one wavefront per frame (two chunks per lane): no idle second wavefront during the fine pass, LDS-bound 15 frames per CU
                long long pc1 = (long long)(160 * 1024) / (long long)lds1;
                if (pc1 > 16) pc1 = 16;
                long long g1w = (long long)num_cu * pc1;
                if (g1w > gcap) g1w = gcap;
                if (g1w > p.n_frames) g1w = p.n_frames;
                hipLaunchKernelGGL((k_sc_cf<128, 2, 4>), dim3((unsigned)g1w), dim3(64), lds1, st, q1);
            } else
            hipLaunchKernelGGL((k_sc_cf<128, 1, 5>), dim3((unsigned)g1), dim3(128), lds1, st, q1);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            q.frame_list = redo_list; q.frame_count = redo_count;
            trace_add(p.trace, "k_sc_cf<256,list>");
        } else trace_add(p.trace, "k_sc_cf<256>");
        hipLaunchKernelGGL((k_sc_cf<256, 2, 4>), dim3((unsigned)grid), dim3(128), lds, st, q); // <= 7 x 2 waves per CU
    } else { // 128-chunk tile (bounded searches): 128 threads, one chunk each, up to 10 workgroups per CU (measured best)
        per_cu = (long long)(160 * 1024) / (long long)lds;
        if (per_cu > 10) per_cu = 10;
        grid = (long long)num_cu * per_cu;
        if (grid > gcap) grid = gcap;
        if (grid > p.n_frames) grid = p.n_frames;
        trace_add(p.trace, "k_sc_cf<128>");
        if (tu.sc128_one_wave) {
            per_cu = (long long)(160 * 1024) / (long long)lds;
            if (per_cu > 16) per_cu = 16;
            grid = (long long)num_cu * per_cu;
            if (grid > gcap) grid = gcap;
            if (grid > p.n_frames) grid = p.n_frames;
            hipLaunchKernelGGL((k_sc_cf<128, 2, 4>), dim3((unsigned)grid), dim3(64), lds, st, q);
        } else
        hipLaunchKernelGGL((k_sc_cf<128, 1, 5>), dim3((unsigned)grid), dim3(128), lds, st, q);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(k_sc_post, dim3((unsigned)((p.n_frames + 255) / 256)), dim3(256), 0, st, p.d_hat, exact, p.n_frames, p.L,
                       p.f_delta, p.metric);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // frames the filter could not settle: all-f64 kernel over the device-side list (usually empty)
    ScParams s = p;
    s.slow_list = slow_list; s.slow_count = slow_count; s.tiles_per_frame = 1; s.mode = 0;
    const size_t lds2 = sc_lds_bytes(s);
    if (lds2 > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_sc_tile), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e != hipSuccess) return e;
    }
    long long g2 = p.n_frames < 512 ? p.n_frames : 512;
    const long long tiles = sc_fast_tiles(p.n_lags);
    if (tiles <= 1) {
        trace_add(p.trace, "k_sc_tile<list>");
        hipLaunchKernelGGL(k_sc_tile, dim3((unsigned)g2), dim3(SC_WG), lds2, st, s);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    } else {
        // more lags than one tile (k_sc80 on slots longer than 2 560 + W + L samples): first crossing per (listed frame, tile), the
        // earliest per frame, then the peak window from there -- the three launches of the general multi-tile search, over the list
        if (tiles > 0x7fffffff) return hipErrorInvalidValue;
        long long *cross = reinterpret_cast<long long *>(reinterpret_cast<unsigned char *>(workspace) +
                                                         (((size_t)p.n_frames * (sizeof(ScExact) + 2 * sizeof(int32_t)) + 128 + 7) & ~(size_t)7));
        int32_t *d1 = reinterpret_cast<int32_t *>(cross + p.n_frames * tiles);
        trace_add(p.trace, "k_sc_tile<list,cross>+k_sc_tile<list,peak>");
        s.tiles_per_frame = (int)tiles; s.mode = 1; s.cross = cross;
        hipLaunchKernelGGL(k_sc_tile, dim3((unsigned)g2), dim3(SC_WG), lds2, st, s);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        hipLaunchKernelGGL(k_sc_min_cross_list, dim3(64), dim3(256), 0, st, cross, (int)tiles, slow_list, slow_count, d1);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        s.tiles_per_frame = 1; s.mode = 2; s.lag_base = d1;
        hipLaunchKernelGGL(k_sc_tile, dim3((unsigned)g2), dim3(SC_WG), lds2, st, s);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (p.stats && p.stats->dev) { // the counters leave the workspace: it may be regrown or reused before anyone asks for them
        if ((e = hipMemcpyAsync(p.stats->dev, slow_count, sizeof(int32_t), hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
        p.stats->has_slow = true;
        if (used_redo) {
            if ((e = hipMemcpyAsync(p.stats->dev + 1, redo_count, sizeof(int32_t), hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
            p.stats->has_redo = true;
        }
    }
    return hipSuccess;
}

__global__ void k_sc_min_cross(const long long *cross, int tiles, long long n_frames, int32_t *d1) {
    long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    long long m = LLONG_MAX;
    for (int t = 0; t < tiles; ++t) { long long c = cross[f * tiles + t]; m = c < m ? c : m; }
    d1[f] = m == LLONG_MAX ? -1 : (int32_t)m;
}
hipError_t run_sc_min_cross(const long long *cross, int tiles, long long n_frames, int32_t *d1, hipStream_t st) {
    if (n_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_sc_min_cross, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, st, cross, tiles,
                       n_frames, d1);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// xcorr_fft (src/signals/mod.rs:186-217): the reference's timing detector -- full cross-correlation of a capture a (N samples)
// with a short signal b (the 80-sample locking ramp in decode, src/receiver.rs:20), zero-padded to 2N - 1, fft_shifted, and
// the FIRST index of the largest |.|^2 (strictly greater replaces; start value 0).  In exact arithmetic
//     out[i] = c[i - (N - 1)],   c[k] = sum_n a[n + k] conj(b[n])      (zero lag at index N - 1; a = 0 outside [0, N))
// and c[k] = 0 for k < -(nb - 1): those indices can never win.  The reference evaluates this with three FFTs of the odd length
// 2N - 1; here every lag is the direct nb-term sum, accumulated in f64 (products of f32 samples are exact in f64), which
// is the same number to ~1e-15 and needs no odd-length transform: one workgroup per (frame, tile of 2048 lags), the tile's
// 2048 + nb samples and b staged in LDS, 8 lags per thread, taps consumed 8 at a time from a 15-sample register window.
struct XcorrBest { double val; long long idx; };
__global__ __launch_bounds__(256) void k_xcorr(const cf *a, long long n_frames, long long stride, long long N, const cf *b, int nb,
                                               int tiles, XcorrBest *tile_best, cf *out, long long out_stride) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int nbp = (nb + 7) / 8 * 8;                 // taps padded with zeros to a multiple of 8
    cf *sa = reinterpret_cast<cf *>(smem);            // [2048 + nbp + 8]
    cf *sb = sa + 2048 + nbp + 8;                     // [nbp]
    __shared__ XcorrBest wbest[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long items = n_frames * (long long)tiles;
    for (long long it = blockIdx.x; it < items; it += gridDim.x) {
        const long long f = it / tiles;
        const int tile = (int)(it - f * tiles);
        const cf *af = a + f * stride;
        const long long k0 = -(long long)(nb - 1) + (long long)tile * 2048; // first lag of the tile
        __syncthreads();
        for (int i = tid; i < 2048 + nbp + 8; i += 256) { const long long j = k0 + i; sa[i] = (j >= 0 && j < N) ? af[j] : make_float2(0.f, 0.f); }
        for (int i = tid; i < nbp; i += 256) sb[i] = i < nb ? b[i] : make_float2(0.f, 0.f);
        __syncthreads();
        const int l0 = tid * 8;
        double cr[8], ci[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { cr[j] = 0.0; ci[j] = 0.0; }
        for (int n0 = 0; n0 < nbp; n0 += 8) {
            double wr[15], wi[15];
#pragma unroll
            for (int j = 0; j < 15; ++j) { const cf v = sa[l0 + n0 + j]; wr[j] = v.x; wi[j] = v.y; }
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const cf t = sb[n0 + n];
                const double br = t.x, bi = t.y; // a * conj(b) = (ar br + ai bi) + j (ai br - ar bi)
#pragma unroll
                for (int j = 0; j < 8; ++j) { cr[j] += wr[j + n] * br + wi[j + n] * bi; ci[j] += wi[j + n] * br - wr[j + n] * bi; }
            }
        }
        XcorrBest mine = XcorrBest{0.0, LLONG_MAX};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long long k = k0 + l0 + j;         // lag; output index i = k + N - 1
            if (k <= N - 1) {
                const double v = cr[j] * cr[j] + ci[j] * ci[j];
                if (v > mine.val) mine = XcorrBest{v, k + N - 1};
                if (out) out[f * out_stride + (k + N - 1)] = make_float2((float)cr[j], (float)ci[j]);
            }
        }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const double ov = __shfl_xor(mine.val, sft, 64);
            const long long oi = __shfl_xor(mine.idx, sft, 64);
            if (ov > mine.val || (ov == mine.val && oi < mine.idx)) mine = XcorrBest{ov, oi};
        }
        if (lane == 0) wbest[wave] = mine;
        __syncthreads();
        if (tid == 0) {
            XcorrBest m = wbest[0];
            for (int wv = 1; wv < 4; ++wv) if (wbest[wv].val > m.val || (wbest[wv].val == m.val && wbest[wv].idx < m.idx)) m = wbest[wv];
            tile_best[it] = m;
        }
        if (out && tile == 0) // indices below N - nb are exact zeros of the correlation
            for (long long i = tid; i < N - nb; i += 256) out[f * out_stride + i] = make_float2(0.f, 0.f);
    }
}
__global__ __launch_bounds__(256) void k_xcorr_pick(const XcorrBest *tile_best, long long n_frames, int tiles, int32_t *idx_max, float *peak) {
    const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
    if (f >= n_frames) return;
    XcorrBest m = XcorrBest{0.0, 0}; // `max = Complex64::default(); idx_max = 0` (signals/mod.rs:206-207)
    for (int t = 0; t < tiles; ++t) { const XcorrBest c = tile_best[f * tiles + t]; if (c.val > m.val) m = c; } // ascending lags: first maximum wins
    idx_max[f] = (int32_t)m.idx;
    if (peak) peak[f] = (float)sqrt(m.val);
}
size_t xcorr_workspace_bytes(long long n_frames, long long N, int nb) {
    const long long tiles = (N + nb - 1 + 2047) / 2048;
    return (size_t)(n_frames * tiles) * sizeof(XcorrBest);
}
hipError_t run_xcorr(const float2 *a, long long n_frames, long long stride, long long N, const float2 *b, int nb, void *workspace,
                     int32_t *idx_max, float *peak, float2 *out, long long out_stride, int num_cu, hipStream_t st) {
    if (n_frames <= 0) return hipSuccess;
    const long long tiles = (N + nb - 1 + 2047) / 2048; // lags -(nb - 1) .. N - 1
    if (tiles > 0x7fffffff) return hipErrorInvalidValue;
    const int nbp = (nb + 7) / 8 * 8;
    const size_t lds = (size_t)(2048 + 2 * nbp + 8) * sizeof(float2);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_xcorr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    long long grid = (long long)num_cu * 4, items = n_frames * tiles;
    if (grid > items) grid = items;
    hipLaunchKernelGGL(k_xcorr, dim3((unsigned)grid), dim3(256), lds, st, a, n_frames, stride, N, b, nb, (int)tiles,
                       reinterpret_cast<XcorrBest *>(workspace), out, out_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_xcorr_pick, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const XcorrBest *>(workspace), n_frames, (int)tiles, idx_max, peak);
    return hipGetLastError();
}

// frequency_correction (src/receiver.rs:231-240): one wavefront per (left, right) pair
__global__ __launch_bounds__(256) void k_freq_corr(const cf *in, long long n_pairs, long long stride,
                                                   long long right_offset, int L, double *f_delta, const int32_t *base,
                                                   const int32_t *status) {
    const int lane = threadIdx.x & 63;
    long long pidx = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pidx >= n_pairs) return;
    if (status && status[pidx] != 0) { if (lane == 0) f_delta[pidx] = 0.0; return; } // no valid block pair in this capture
    const cf *l = in + pidx * stride + (base ? base[pidx] : 0), *r = l + right_offset;
    double sum = 0.0;
    for (int m = lane; m < L; m += 64) {    // all f64 like the reference (receiver.rs:231-240): products of f32 are exact
        const double lr = l[m].x, li = l[m].y, rr = r[m].x, ri = r[m].y;
        const double ns = lr * lr + li * li; // the reference divides first (num-complex Div): r / l = r conj(l) / |l|^2
        sum += atan2((ri * lr - rr * li) / ns, (rr * lr + ri * li) / ns);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) sum += __shfl_xor(sum, s, 64);
    if (lane == 0) f_delta[pidx] = fabs((sum / (double)L) / (double)L);
}
hipError_t run_freq_correction(const float2 *in, long long n_pairs, long long stride, long long right_offset, int L,
                               double *f_delta, hipStream_t st, const int32_t *base, const int32_t *status) {
    if (n_pairs <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_freq_corr, dim3((unsigned)((n_pairs + 3) / 4)), dim3(256), 0, st, in, n_pairs, stride,
                       right_offset, L, f_delta, base, status);
    return hipGetLastError();
}

// CFO derotation (src/receiver.rs:44-50)
__global__ __launch_bounds__(256) void k_cfo_rotate(cf *x, long long n_frames, long long frame_stride,
                                                    long long frame_len, const double *f_delta,
                                                    const int32_t *first_index, int chunks) {
    const long long f = blockIdx.x / chunks;
    const int c = (int)(blockIdx.x - f * chunks);
    const double turns = f_delta[f] * 0.15915494309189533577;
    const long long first = first_index ? first_index[f] : 0;
    cf *row = x + f * frame_stride;
    for (long long n = (long long)c * 1024 + threadIdx.x; n < frame_len && n < (long long)(c + 1) * 1024; n += 256)
        row[n] = cmul(row[n], cfo_phasor(turns, first + n));
}
hipError_t run_cfo_rotate(float2 *x, long long n_frames, long long frame_stride, long long frame_len,
                          const double *f_delta, const int32_t *first_index, hipStream_t st) {
    if (n_frames <= 0 || frame_len <= 0) return hipSuccess;
    long long chunks = (frame_len + 1023) / 1024;
    long long blocks = chunks * n_frames;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_cfo_rotate, dim3((unsigned)blocks), dim3(256), 0, st, x, n_frames, frame_stride, frame_len,
                       f_delta, first_index, (int)chunks);
    return hipGetLastError();
}

} // namespace ofdm
