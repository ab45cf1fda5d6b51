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
    __shared__ int s_d1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long f = blockIdx.x / p.tiles_per_frame;
    const int tile = (int)(blockIdx.x - f * p.tiles_per_frame);
    const int L = p.L, W = p.W;

    long long d0;
    if (p.mode == 2) {
        int lb = p.lag_base[f];
        if (lb < 0) { // no crossing anywhere in this frame
            if (tid == 0) { p.d_hat[f] = -1; if (p.f_delta) p.f_delta[f] = 0.0; if (p.metric) p.metric[f] = 0.f; }
            return;
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
            if (tid == 0) p.cross[blockIdx.x] = d1 == INT_MAX ? LLONG_MAX : d0 + d1;
            return;
        }
        if (d1 == INT_MAX) {
            if (tid == 0) { p.d_hat[f] = -1; if (p.f_delta) p.f_delta[f] = 0.0; if (p.metric) p.metric[f] = 0.f; }
            return;
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
    hipLaunchKernelGGL(k_sc_tile, dim3((unsigned)blocks), dim3(SC_WG), lds, st, p);
    return hipGetLastError();
}
int sc_tile_lags() { return SC_CH; }

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

// frequency_correction (src/receiver.rs:231-240): one wavefront per (left, right) pair
__global__ __launch_bounds__(256) void k_freq_corr(const cf *in, long long n_pairs, long long stride,
                                                   long long right_offset, int L, double *f_delta) {
    const int lane = threadIdx.x & 63;
    long long pidx = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pidx >= n_pairs) return;
    const cf *l = in + pidx * stride, *r = l + right_offset;
    double sum = 0.0;
    for (int m = lane; m < L; m += 64) {
        cf q = cmulc(r[m], l[m]);            // angle(r / l) == angle(r * conj(l))
        cf a = l[m];
        float ns = a.x * a.x + a.y * a.y;    // the reference divides first (num-complex Div)
        sum += (double)atan2f(q.y / ns, q.x / ns);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) sum += __shfl_xor(sum, s, 64);
    if (lane == 0) f_delta[pidx] = fabs((sum / (double)L) / (double)L);
}
hipError_t run_freq_correction(const float2 *in, long long n_pairs, long long stride, long long right_offset, int L,
                               double *f_delta, hipStream_t st) {
    if (n_pairs <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_freq_corr, dim3((unsigned)((n_pairs + 3) / 4)), dim3(256), 0, st, in, n_pairs, stride,
                       right_offset, L, f_delta);
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
