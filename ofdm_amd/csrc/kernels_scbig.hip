// kernels_scbig.hip -- Schmidl-Cox timing for LONG periods (L = S >= 160: N >= 128), where neither the frame nor even one
// correlation window (W + L samples: 164 KB at N = 4096) fits in LDS.
//
// The sliding sums are prefix differences:
//     P(d) = Q[d + W] - Q[d],   E(d) = Ep[d + W] - Ep[d],   R(d) = Ep[d + W + L] - Ep[d + L],
//     Q[n] = sum_{m<n} conj(r[m]) r[m+L],  Ep[n] = sum_{m<n} |r[m]|^2.
// so the capture is read ONCE to form chunk sums of q and e (chunks of C = L/8 samples, f64: products of f32 samples are
// exact in f64), and the per-lag metric is only evaluated where it can matter:
//   k_scb_chunks   one workgroup per (frame, tile of 2560 samples): the tile and its partner tile L samples later are staged
//                  coalesced into LDS, every thread sums one 10-sample micro-chunk in f64, C/10 neighbouring lanes are
//                  combined with wave shuffles, chunk totals go to a small per-frame workspace (24 B per chunk: 2 % of the
//                  capture).  HBM roofline: 8 B/sample read once (the partner tile is an L2 / Infinity-Cache re-read).
//   k_scb_fine     one workgroup per frame: f64 prefix sums over the chunk totals, exact sums at every chunk boundary, and
//                  for every chunk an upper bound of the metric over its C lags,
//                      |P(d)| <= |P0| + 1/2 (Te[c] + Te[c+L] + Te[c+W] + Te[c+W+L]),  E(d) >= E0 - Te[c],  R(d) >= R0 - Te[c+L],
//                  so only chunks whose bound reaches the threshold are searched for the first crossing d1, and only chunks
//                  whose bound reaches the best exact value found so far are searched for the peak over [d1, d1 + W].
//                  A search evaluates a tile of 320 (N <= 2048) or 640 lags exactly: four segments of that many samples (at d0,
//                  d0+L, d0+W, d0+W+L) are staged into LDS, each thread slides the three sums over its 5 / 10 lags in f64 from the
//                  tile's boundary sums
//                  (prefix differences), a workgroup scan chains the threads.
// Every decision is taken on f64 sums, as k_sc_tile does: timing indices equal the f64 oracle's except on ties below f64
// resolution.  Works for any N in 128..4096 (the LDS footprint does not depend on L) and any capture of up to 2047 chunks.
#include "device_common.hpp"
#include "kernels.hpp"
#include <limits.h>

namespace ofdm {

namespace {

constexpr int B_TILE = 2560;   // samples per chunk-sum tile (256 threads x 10)
constexpr int F_WG = 64;        // one wavefront per frame; tiles of 5 or 10 lags per thread (run_sc_big): 11 / 21 KB of LDS, 12 / 7 frames
                                // in flight per CU (128 threads / 1280 lags: 3 per CU, 25 % slower)

struct BSums { double pr, pi, e, r; };
__device__ __forceinline__ BSums bs_add(BSums a, BSums b) { return BSums{a.pr + b.pr, a.pi + b.pi, a.e + b.e, a.r + b.r}; }
struct BCand { double num, den, pr, pi; int lag; };
// first maximum wins, whatever the order of discovery: strictly greater replaces; equal replaces only from a lower lag
__device__ __forceinline__ BCand bc_pick(BCand a, BCand b) {
    const double lhs = b.num * a.den, rhs = a.num * b.den;
    return (lhs > rhs || (lhs == rhs && b.lag < a.lag)) ? b : a;
}
__device__ __forceinline__ BCand bc_shfl_xor(BCand a, int d) {
    return BCand{__shfl_xor(a.num, d, 64), __shfl_xor(a.den, d, 64), __shfl_xor(a.pr, d, 64), __shfl_xor(a.pi, d, 64),
                 __shfl_xor(a.lag, d, 64)};
}

// Coalesced staging of NSEG segments of PER * nthr * 2 samples each into LDS, zero past the capture.  All loads of all
// segments are issued before the first LDS store (a load -> store -> load chain would expose the memory latency once per
// 16 bytes: measured 20 us per 40 KB tile).
template <int NSEG, int PER>
__device__ __forceinline__ void stage_segments(cf *const *dst, const cf *frame, const long long *first, long long frame_len, int tid, int nthr) {
    float4 x[NSEG][PER];
#pragma unroll
    for (int g = 0; g < NSEG; ++g) {
        const cf *src = frame + first[g];
        const long long avail = frame_len - first[g]; // may be <= 0
        const bool al = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid + j * nthr;             // float4 index inside the segment
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (2 * i + 1 < avail) {
                if (al) v = reinterpret_cast<const float4 *>(src)[i];
                else { const cf a = src[2 * i], b = src[2 * i + 1]; v = make_float4(a.x, a.y, b.x, b.y); }
            } else if (2 * i < avail) { const cf a = src[2 * i]; v.x = a.x; v.y = a.y; }
            x[g][j] = v;
        }
    }
#pragma unroll
    for (int g = 0; g < NSEG; ++g)
#pragma unroll
        for (int j = 0; j < PER; ++j) reinterpret_cast<float4 *>(dst[g])[tid + j * nthr] = x[g][j];
}

} // namespace

// ---- chunk totals: ws[f][0 .. nch) = sum q.re, [nch_pad ..) = sum q.im, [2 nch_pad ..) = sum e   (doubles)
// Two stagings: L <= 1280 (N <= 1024): ONE contiguous span of 5120 + L samples per item (the partner sample n + L lives in the
// same LDS image: 1.25 global reads per sample instead of 2; two micro-chunks per thread); longer periods: the tile of 2560
// samples and its partner tile L samples later as two separate segments (LDS footprint independent of L).
__device__ __forceinline__ void chunk_sums(const cf *a, const cf *b, double &qr, double &qi, double &e) {
    const float4 *pa = reinterpret_cast<const float4 *>(a), *pb = reinterpret_cast<const float4 *>(b);
    qr = 0.0; qi = 0.0; e = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float4 x = pa[i], y = pb[i];
        const double ar = x.x, ai = x.y, br = y.x, bi = y.y, cr = x.z, ci = x.w, dr = y.z, di = y.w;
        qr += ar * br + ai * bi; qi += ar * bi - ai * br; e += ar * ar + ai * ai;
        qr += cr * dr + ci * di; qi += cr * di - ci * dr; e += cr * cr + ci * ci;
    }
}
template <bool CONTIG>
__global__ __launch_bounds__(256) void k_scb_chunks(ScBigParams p) {
    extern __shared__ __align__(16) unsigned char smem_c[];
    cf *seg_a = reinterpret_cast<cf *>(smem_c);       // CONTIG: [2 B_TILE + L]; else [B_TILE]
    cf *seg_b = seg_a + B_TILE;                       // !CONTIG: [B_TILE] the partner tile
    constexpr int TILE = CONTIG ? 2 * B_TILE : B_TILE;
    const int tid = threadIdx.x;
    const int k = p.C / 10;                  // micro-chunks per chunk: N / 64, a power of two <= 64
    const long long items = p.n_frames * (long long)p.tiles_per_frame;
    for (long long it = blockIdx.x; it < items; it += gridDim.x) {
        const long long f = it / p.tiles_per_frame;
        const int tile = (int)(it - f * p.tiles_per_frame);
        const cf *frame = p.in + f * p.frame_stride;
        const long long t0 = (long long)tile * TILE;
        __syncthreads(); // the previous item's readers are done
        if (CONTIG) {
            // (2 B_TILE + L) / 2 float4 per item, L <= 1280: 13 loads per thread at most, all issued before the first LDS store
            const int n4 = (TILE + p.L) / 2;
            const long long avail = p.frame_len - t0;
            const bool al = (reinterpret_cast<uintptr_t>(frame + t0) & 15) == 0;
            float4 x[13];
#pragma unroll
            for (int j = 0; j < 13; ++j) {
                const int i = tid + j * 256;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < n4) {
                    if (2 * i + 1 < avail) {
                        if (al) v = reinterpret_cast<const float4 *>(frame + t0)[i];
                        else { const cf a = frame[t0 + 2 * i], b = frame[t0 + 2 * i + 1]; v = make_float4(a.x, a.y, b.x, b.y); }
                    } else if (2 * i < avail) { const cf a = frame[t0 + 2 * i]; v.x = a.x; v.y = a.y; }
                }
                x[j] = v;
            }
#pragma unroll
            for (int j = 0; j < 13; ++j) { const int i = tid + j * 256; if (i < n4) reinterpret_cast<float4 *>(seg_a)[i] = x[j]; }
        } else {
            cf *const dsts[2] = {seg_a, seg_b};
            const long long firsts[2] = {t0, t0 + p.L};
            stage_segments<2, B_TILE / 2 / 256>(dsts, frame, firsts, p.frame_len, tid, 256);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < (CONTIG ? 2 : 1); ++u) {
            const int mc = tid + 256 * u;        // micro-chunk of 10 samples inside the tile
            double qr, qi, e;
            chunk_sums(seg_a + mc * 10, CONTIG ? seg_a + mc * 10 + p.L : seg_b + mc * 10, qr, qi, e);
            for (int sft = 1; sft < k; sft <<= 1) { // k consecutive lanes -> one chunk (k <= 64: inside the wavefront)
                qr += __shfl_xor(qr, sft, 64); qi += __shfl_xor(qi, sft, 64); e += __shfl_xor(e, sft, 64);
            }
            if ((tid & (k - 1)) == 0) {
                const int c = (int)(t0 / p.C) + mc / k;
                if (c < p.nch) {
                    double *w = p.ws + f * 3LL * p.nch_pad;
                    w[c] = qr; w[p.nch_pad + c] = qi; w[2 * p.nch_pad + c] = e;
                }
            }
        }
    }
}

// ---- prefix sums, chunk bounds, first crossing, peak
template <int LPT>
__global__ __launch_bounds__(F_WG, LPT == 10 ? 2 : 3) void k_scb_fine(ScBigParams p) {
    constexpr int F_TILE = LPT * F_WG;   // lags per fine tile
    extern __shared__ __align__(16) unsigned char smem[];
    cf *s0 = reinterpret_cast<cf *>(smem);       // [F_TILE + 16] samples at d0 ..
    cf *s1 = s0 + F_TILE + 16;                   // ... at d0 + L
    cf *s2 = s1 + F_TILE + 16;                   // ... at d0 + W
    cf *s3 = s2 + F_TILE + 16;                   // ... at d0 + W + L
    float *mub = reinterpret_cast<float *>(s3 + F_TILE + 16);    // [nch_pad] upper bound of M over the chunk's lags (rounded up)
    BSums *wsum = reinterpret_cast<BSums *>(mub + p.nch_pad);    // [2] wave totals
    BCand *wcand = reinterpret_cast<BCand *>(wsum + 2);          // [2]
    int *wmin = reinterpret_cast<int *>(wcand + 2);              // [2]
    int *shi = wmin + 2;                                         // [2] scratch

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = p.L, W = p.W, C = p.C, nch = p.nch, cL = L / C, cW = W / C;
    const long long n = p.n_lags;
    const int n_bound = (int)((n + C - 1) / C);  // chunks that own at least one searched lag
    const double thr = p.threshold;

    for (long long f = blockIdx.x; f < p.n_frames; f += gridDim.x) {
        const cf *frame = p.in + f * p.frame_stride;
        double *Qr = p.ws + f * 3LL * p.nch_pad, *Qi = Qr + p.nch_pad, *Ep = Qi + p.nch_pad;
        __syncthreads();
        // ---- exclusive prefix sums over the chunk totals, in place (wave 0; 64 chunks per step, carry in registers)
        if (wave == 0) {
            double cr = 0.0, ci = 0.0, ce = 0.0;
            for (int base = 0; base < nch; base += 64) {
                const int c = base + lane;
                double a = c < nch ? Qr[c] : 0.0, b = c < nch ? Qi[c] : 0.0, e = c < nch ? Ep[c] : 0.0;
                double ia = a, ib = b, ie = e;
#pragma unroll
                for (int sft = 1; sft < 64; sft <<= 1) {
                    const double oa = __shfl_up(ia, sft, 64), ob = __shfl_up(ib, sft, 64), oe = __shfl_up(ie, sft, 64);
                    if (lane >= sft) { ia += oa; ib += ob; ie += oe; }
                }
                if (c < nch) { Qr[c] = cr + (ia - a); Qi[c] = ci + (ib - b); Ep[c] = ce + (ie - e); }
                cr += __shfl(ia, 63, 64); ci += __shfl(ib, 63, 64); ce += __shfl(ie, 63, 64);
            }
            if (lane == 0) { Qr[nch] = cr; Qi[nch] = ci; Ep[nch] = ce; } // entry nch = the total
        }
        __threadfence_block();
        __syncthreads();
        // exact sums at chunk boundary c (lag c C)
        auto boundary = [&](int c) -> BSums {
            return BSums{Qr[c + cW] - Qr[c], Qi[c + cW] - Qi[c], Ep[c + cW] - Ep[c], Ep[c + cW + cL] - Ep[c + cL]};
        };
        // ---- upper bound of the metric over each chunk's lags
        for (int c = tid; c < n_bound; c += F_WG) {
            const BSums b = boundary(c);
            const double te0 = Ep[c + 1] - Ep[c], teL = Ep[c + cL + 1] - Ep[c + cL], teW = Ep[c + cW + 1] - Ep[c + cW],
                         teWL = Ep[c + cW + cL + 1] - Ep[c + cW + cL];
            const double ub = sqrt(b.pr * b.pr + b.pi * b.pi) * (1.0 + 1e-12) + 0.5 * (te0 + teL + teW + teWL);
            const double elo = b.e - te0, rlo = b.r - teL;
            float m = 3.0e38f;
            if (elo > 0.0 && rlo > 0.0) { const double q = ub * ub / (elo * rlo) * (1.0 + 1e-9); m = q < 3.0e38 ? (float)q * 1.000001f : 3.0e38f; }
            if (!(ub > 0.0)) m = 0.f; // an all-zero neighbourhood: no lag of this chunk has a defined metric
            mub[c] = m;
        }
        __syncthreads();

        // Exact evaluation of the F_TILE lags from d0 (a chunk boundary): first crossing in [lo, hi] (if want_cross), then the
        // first maximum over [max(lo, crossing), hi].  Returns the crossing (INT_MAX: none) through *cross_out.
        auto eval_tile = [&](long long d0, long long lo, long long hi, bool want_cross, long long &cross_out, BCand &best) {
            __syncthreads();
            {
                cf *const dsts[4] = {s0, s1, s2, s3};
                const long long firsts[4] = {d0, d0 + L, d0 + W, d0 + W + L};
                if (LPT % 2 == 0) stage_segments<4, (LPT % 2 == 0 ? LPT / 2 : 1)>(dsts, frame, firsts, p.frame_len, tid, F_WG);
                else { // odd lags per thread: 8-byte loads, all issued before the first LDS store
                    cf x[4][LPT];
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int j = 0; j < LPT; ++j) {
                            const long long n = firsts[g] + tid + j * F_WG;
                            x[g][j] = n < p.frame_len ? frame[n] : make_float2(0.f, 0.f);
                        }
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int j = 0; j < LPT; ++j) dsts[g][tid + j * F_WG] = x[g][j];
                }
            }
            __syncthreads();
            const int a0 = tid * LPT;
            BSums pre[LPT];
            {
                BSums run = BSums{0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < LPT; ++j) {
                    const cf x0 = s0[a0 + j], x1 = s1[a0 + j], x2 = s2[a0 + j], x3 = s3[a0 + j];
                    const double r0 = x0.x, i0 = x0.y, r1 = x1.x, i1 = x1.y, r2 = x2.x, i2 = x2.y, r3 = x3.x, i3 = x3.y;
                    run.pr += (r2 * r3 + i2 * i3) - (r0 * r1 + i0 * i1);
                    run.pi += (r2 * i3 - i2 * r3) - (r0 * i1 - i0 * r1);
                    run.e += (r2 * r2 + i2 * i2) - (r0 * r0 + i0 * i0);
                    run.r += (r3 * r3 + i3 * i3) - (r1 * r1 + i1 * i1);
                    pre[j] = run;
                }
            }
            BSums inc = pre[LPT - 1];
#pragma unroll
            for (int sft = 1; sft < 64; sft <<= 1) {
                const BSums o = BSums{__shfl_up(inc.pr, sft, 64), __shfl_up(inc.pi, sft, 64), __shfl_up(inc.e, sft, 64), __shfl_up(inc.r, sft, 64)};
                if (lane >= sft) inc = bs_add(inc, o);
            }
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            BSums base = boundary((int)(d0 / C));
            for (int wv = 0; wv < wave; ++wv) base = bs_add(base, wsum[wv]);
            {
                const BSums ex = BSums{__shfl_up(inc.pr, 1, 64), __shfl_up(inc.pi, 1, 64), __shfl_up(inc.e, 1, 64), __shfl_up(inc.r, 1, 64)};
                if (lane > 0) base = bs_add(base, ex);
            }
            // base = sums at lag d0 + a0
            long long cross = LLONG_MAX;
            if (want_cross) {
                int mine = INT_MAX;
#pragma unroll
                for (int j = LPT - 1; j >= 0; --j) {
                    const BSums x = j ? bs_add(base, pre[j - 1]) : base;
                    const double num = x.pr * x.pr + x.pi * x.pi, den = x.e * x.r;
                    const long long lag = d0 + a0 + j;
                    if (lag >= lo && lag <= hi && den > 0.0 && num >= thr * den) mine = a0 + j;
                }
#pragma unroll
                for (int sft = 32; sft >= 1; sft >>= 1) { const int o = __shfl_xor(mine, sft, 64); mine = o < mine ? o : mine; }
                if (lane == 0) wmin[wave] = mine;
                __syncthreads();
                int m = wmin[0];
                for (int wv = 1; wv < F_WG / 64; ++wv) m = wmin[wv] < m ? wmin[wv] : m;
                if (m != INT_MAX) { cross = d0 + m; lo = cross; if (hi > cross + W) hi = cross + W; }
                cross_out = cross;
                if (m == INT_MAX) return; // wave-uniform: no crossing in this tile, nothing to maximise yet
            }
            BCand mineb = BCand{-1.0, 1.0, 0.0, 0.0, INT_MAX};
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                const BSums x = j ? bs_add(base, pre[j - 1]) : base;
                const double num = x.pr * x.pr + x.pi * x.pi, den = x.e * x.r;
                const long long lag = d0 + a0 + j;
                if (lag >= lo && lag <= hi && den > 0.0) mineb = bc_pick(mineb, BCand{num, den, x.pr, x.pi, (int)lag});
            }
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) mineb = bc_pick(mineb, bc_shfl_xor(mineb, sft));
            if (lane == 0) wcand[wave] = mineb;
            __syncthreads();
            for (int wv = 0; wv < F_WG / 64; ++wv) best = bc_pick(best, wcand[wv]);
        };

        // ---- first crossing: chunks whose bound reaches the threshold, in lag order
        long long d1 = LLONG_MAX, tile0 = 0;
        BCand best = BCand{-1.0, 1.0, 0.0, 0.0, INT_MAX};
        const float thr_f = (float)thr * 0.999999f;
        int c = 0;
        for (;;) {
            // next flagged chunk >= c (all threads scan the same LDS table: wave-uniform result)
            int cn = INT_MAX;
            for (int i = c + tid; i < n_bound; i += F_WG) if (mub[i] >= thr_f) { cn = i; break; }
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) { const int o = __shfl_xor(cn, sft, 64); cn = o < cn ? o : cn; }
            if (lane == 0) shi[wave] = cn;
            __syncthreads();
            cn = shi[0];
            for (int wv = 1; wv < F_WG / 64; ++wv) cn = shi[wv] < cn ? shi[wv] : cn;
            __syncthreads();
            if (cn == INT_MAX) break;
            tile0 = (long long)cn * C;
            long long cross;
            eval_tile(tile0, tile0, n - 1, true, cross, best);
            if (cross != LLONG_MAX) { d1 = cross; break; }
            c = cn + F_TILE / C;
        }
        if (d1 == LLONG_MAX) {
            if (tid == 0) { p.d_hat[f] = -1; if (p.f_delta) p.f_delta[f] = 0.0; if (p.metric) p.metric[f] = 0.f; }
            continue;
        }
        // ---- peak over [d1, dend]: the crossing's tile is already in `best`; chunk boundaries inside the window are exact
        //      candidates for free; a further tile is evaluated only if one of its chunks can still beat the best
        const long long dend = d1 + W < n - 1 ? d1 + W : n - 1;
        {
            __syncthreads(); // every thread has merged the crossing tile's candidates out of wcand
            BCand bb = BCand{-1.0, 1.0, 0.0, 0.0, INT_MAX};
            for (int cb = (int)((d1 + C - 1) / C) + tid; (long long)cb * C <= dend; cb += F_WG) {
                const BSums x = boundary(cb);
                const double num = x.pr * x.pr + x.pi * x.pi, den = x.e * x.r;
                if (den > 0.0) bb = bc_pick(bb, BCand{num, den, x.pr, x.pi, cb * C});
            }
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) bb = bc_pick(bb, bc_shfl_xor(bb, sft));
            if (lane == 0) wcand[wave] = bb;
            __syncthreads();
            for (int wv = 0; wv < F_WG / 64; ++wv) best = bc_pick(best, wcand[wv]);
            __syncthreads();
        }
        for (long long t = tile0 + F_TILE; t <= dend; t += F_TILE) {
            const float need = (float)(best.num / best.den) * 0.999999f; // a chunk must be able to reach this to matter
            bool any = false;
            for (int i = (int)(t / C); i < n_bound && (long long)i * C <= dend && (long long)i * C < t + F_TILE; ++i) any |= mub[i] >= need;
            if (!any) continue; // wave- and workgroup-uniform (same table, same best)
            long long dummy;
            eval_tile(t, d1, dend, false, dummy, best);
        }
        if (tid == 0) {
            if (best.lag == INT_MAX) { p.d_hat[f] = -1; if (p.f_delta) p.f_delta[f] = 0.0; if (p.metric) p.metric[f] = 0.f; }
            else {
                p.d_hat[f] = best.lag;
                if (p.f_delta) p.f_delta[f] = atan2(best.pi, best.pr) / (double)L;
                if (p.metric) p.metric[f] = (float)(best.num / best.den);
            }
        }
    }
}

// L = 8 C, C = 10 k with k = N / 64 a power of two <= 64; every searched lag's window inside nch chunks
bool sc_big_ok(const ScParams &p) {
    if (p.mode != 0 || p.L % 80 != 0 || p.W % p.L != 0) return false;
    const int k = p.L / 80;
    if (k < 2 || k > 64 || (k & (k - 1))) return false;
    const int C = p.L / 8;
    const long long nch = (p.n_lags + p.W + p.L + C - 1) / C + 1;
    return nch <= 4096;
}
size_t sc_big_workspace_bytes(const ScParams &p) {
    const int C = p.L / 8;
    const long long nch = (p.n_lags + p.W + p.L + C - 1) / C + 1, pad = (nch + 1 + 7) / 8 * 8;
    return (size_t)p.n_frames * 3 * (size_t)pad * sizeof(double);
}
hipError_t run_sc_big(const ScParams &p, void *workspace, int num_cu, hipStream_t st) {
    if (p.n_frames <= 0) return hipSuccess;
    ScBigParams q;
    q.in = p.in; q.n_frames = p.n_frames; q.frame_stride = p.frame_stride; q.frame_len = p.frame_len; q.n_lags = p.n_lags;
    q.L = p.L; q.W = p.W; q.C = p.L / 8; q.threshold = p.threshold;
    q.nch = (int)((p.n_lags + p.W + p.L + q.C - 1) / q.C + 1);
    q.nch_pad = (q.nch + 1 + 7) / 8 * 8;
    const Tuning &tu = tuning_or_default(p.tune);
    const bool contig = p.L <= 1280 && !tu.scb_two_segments; // the partner sample fits the same LDS image
    const int tile = contig ? 2 * B_TILE : B_TILE;
    q.tiles_per_frame = (int)(((long long)q.nch * q.C + tile - 1) / tile);
    q.ws = reinterpret_cast<double *>(workspace);
    q.d_hat = p.d_hat; q.f_delta = p.f_delta; q.metric = p.metric;
    const long long items = p.n_frames * (long long)q.tiles_per_frame;
    const size_t lds1 = contig ? (size_t)(2 * B_TILE + p.L) * sizeof(float2) : (size_t)2 * B_TILE * sizeof(float2);
    long long g1 = (long long)num_cu * ((long long)(160 * 1024) / (long long)lds1);
    if (tu.grid_cap > 0 && g1 > tu.grid_cap) g1 = tu.grid_cap;
    if (g1 > items) g1 = items;
    hipError_t e;
    trace_add(p.trace, contig ? "k_scb_chunks<contig>" : "k_scb_chunks");
    if (contig) {
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scb_chunks<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_scb_chunks<true>, dim3((unsigned)g1), dim3(256), lds1, st, q);
    } else hipLaunchKernelGGL(k_scb_chunks<false>, dim3((unsigned)g1), dim3(256), lds1, st, q);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // Tiles start at chunk boundaries, so a tile is a whole number of chunks: 320-lag tiles (5 lags per thread) where C <= 320
    // (N <= 2048), 640-lag tiles for N = 4096.  The small tile halves the LDS and the registers of a frame (12 instead of 7
    // frames in flight per CU) at twice the number of tiles.
    const bool small = q.C <= 320 && (320 % q.C) == 0 && !tu.scb_big_tiles;
    const int f_tile = small ? 5 * F_WG : 10 * F_WG;
    const size_t lds = (size_t)4 * (f_tile + 16) * sizeof(float2) + (size_t)q.nch_pad * sizeof(float) + 2 * sizeof(BSums) +
                       2 * sizeof(BCand) + 4 * sizeof(int) + 64;
    long long per_cu = (long long)(160 * 1024) / (long long)lds;
    const long long cap = small ? 12 : 8; // 3 / 2 waves per SIMD (163 / 236 VGPRs; 4 waves spill 22 registers for +2 %)
    if (per_cu > cap) per_cu = cap;
    long long g2 = (long long)num_cu * per_cu;
    if (tu.grid_cap > 0 && g2 > tu.grid_cap) g2 = tu.grid_cap;
    if (g2 > p.n_frames) g2 = p.n_frames;
    trace_add(p.trace, small ? "k_scb_fine<5>" : "k_scb_fine<10>");
    if (small) hipLaunchKernelGGL(k_scb_fine<5>, dim3((unsigned)g2), dim3(F_WG), lds, st, q);
    else hipLaunchKernelGGL(k_scb_fine<10>, dim3((unsigned)g2), dim3(F_WG), lds, st, q);
    return hipGetLastError();
}

} // namespace ofdm
