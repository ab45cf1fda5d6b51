// kernels_sc80.hip -- k_sc80: Schmidl-Cox timing for L = 80, W = 240 (N = 64; EXT-3, oracle: orc_sc_sync) with EVERY lag
// evaluated on f64 sums, in one streaming pass over the slot that stops when the peak window has closed -- wherever in
// its slot the packet sits.  Replaces the f32 coarse / fine filter (k_sc_cf<128,first> + k_sc_cf<256,list>, kernels_sync.hip)
// as the product's search for N = 64; that pair stays behind tuning "no_sc80" for the A/B.
//
// Why every lag can be exact and still cheap: L = 80 and W = 3 L, so everything the detector needs lines up modulo 80.
//   SE(n) = sum_{m < n} |r[m]|^2,   SQ(n) = sum_{m < n} conj(r[m]) r[m + 80]           (running prefixes, f64;
//   E(d) = SE(d + 240) - SE(d),  R(d) = E(d + 80),  P(d) = SQ(d + 240) - SQ(d)           products of f32 are exact in f64)
// A frame is handled by ONE ROW of 16 lanes; lane l' owns the five residues 5 l' .. 5 l' + 4 modulo 80.  The four prefix
// values a lag needs -- SE / SQ at d, d + 80, d + 240, d + 320 -- then all live in the SAME lane, three and four
// period-steps apart: no sample, product or prefix ever crosses a lane except through the one 16-lane scan per period
// (row_shr DPP) that turns the lanes' period sums into prefixes.  Four frames per wavefront, persistent over the batch.
// Per period-step (80 samples of each of the 4 frames) a wavefront
//   * reads its 5 samples per lane from the LDS ring the DMA fills (below),
//   * forms e, q for them, the in-lane partial sums, the row scan, the new prefixes SE_t, SQ_{t-1},
//   * E(lags of period t - 3) = SE_t - SE_{t-3}: that is R of the lags of period t - 4, whose E is last step's value and whose
//     P = SQ_{t-1} - SQ_{t-4}.  M >= thr and "M > best" are the oracle's own expressions on those sums;
//   * rotates three-deep rings of SE and SQ that live in registers (the loop body is unrolled over six steps, so ring slots
//     are register names).
// The samples arrive by LDS-DMA (global_load_lds_dwordx4): a DMA instruction moves 256 contiguous bytes of each of the four
// frames (16 lanes x 16 B); the ring holds 15 such pieces = 6 periods per frame, 7 pieces are kept in flight
// (s_waitcnt vmcnt(7), constant along the schedule), and once a row knows its crossing nothing beyond lag d1 + W's window
// is requested any more (the addresses are clamped, so the instruction count the vmcnt logic relies on never changes).
// Unit u of a 256-byte piece lands at unit u ^ 8 in the odd rows: the two rows that share a ds_read_b64 pass then hit
// disjoint banks (5 l' mod 32 and 5 l' + 16 mod 32 are complementary).
//
// Exactness.  Decisions are taken on prefix DIFFERENCES, whose absolute error is ~1e-15 of the prefix magnitude.  A window
// whose energy is below 2^-20 of the energy seen so far is therefore not trusted (unless nothing but zeros came before it,
// where the difference is exact): if such a lag could matter, the frame goes to the slow list and k_sc_tile redoes it with
// direct f64 sums.  Never seen on captures with less than 60 dB of dynamic range; all-zero lead-ins stay on this path.
// Roofline: HBM, 8 B per sample actually needed (the slot up to d1 + 2 W + L, rounded up to pieces).
#include "device_common.hpp"
#include "kernels.hpp"
#include <limits.h>

namespace ofdm {

namespace {

constexpr int S8_PIECES = 15;   // ring: 15 pieces of 1 KiB (4 rows x 256 B) = 6 periods of 640 B per row

struct S80Params {
    const float2 *in;
    long long n_frames, frame_stride, frame_len;
    int n_lags;
    double threshold;
    int32_t *d_hat;
    ScExact *exact;
    int32_t *slow_list, *slow_count;
};

template <int CTRL> __device__ __forceinline__ double s8_dpp(double x) {   // lanes without a source lane read 0
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ int s8_dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }
// inclusive prefix sum over the 16 lanes of a row
__device__ __forceinline__ double s8_row_scan(double x) {
    x += s8_dpp<0x111>(x);   // row_shr:1
    x += s8_dpp<0x112>(x);   // row_shr:2
    x += s8_dpp<0x114>(x);   // row_shr:4
    x += s8_dpp<0x118>(x);   // row_shr:8
    return x;
}
__device__ __forceinline__ double s8_row_last(double x) { return s8_dpp<0x15F>(x); }   // row_newbcast:15
template <int N> __device__ __forceinline__ void s8_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void s8_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void s8_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

struct S8Cand { double num, den, pr, pi; int lag; };
// first maximum wins, whatever the order of discovery: strictly greater replaces; equal replaces only from a lower lag
__device__ __forceinline__ S8Cand s8_pick(S8Cand a, S8Cand b) {
    const double lhs = b.num * a.den, rhs = a.num * b.den;
    return (lhs > rhs || (lhs == rhs && b.lag < a.lag)) ? b : a;
}
template <int CTRL> __device__ __forceinline__ S8Cand s8_cand_dpp(S8Cand c) {
    return S8Cand{s8_dpp<CTRL>(c.num), s8_dpp<CTRL>(c.den), s8_dpp<CTRL>(c.pr), s8_dpp<CTRL>(c.pi), s8_dpp_i<CTRL>(c.lag)};
}
// best candidate of the row, in every lane of the row (xor 1, xor 2, the other quad of the half, the other half)
__device__ __forceinline__ S8Cand s8_row_best(S8Cand c) {
    c = s8_pick(c, s8_cand_dpp<0xB1>(c));    // quad_perm [1,0,3,2]
    c = s8_pick(c, s8_cand_dpp<0x4E>(c));    // quad_perm [2,3,0,1]
    c = s8_pick(c, s8_cand_dpp<0x141>(c));   // row_half_mirror
    c = s8_pick(c, s8_cand_dpp<0x140>(c));   // row_mirror
    return c;
}
__device__ __forceinline__ int s8_row_min(int x) {
    x = min(x, s8_dpp_i<0xB1>(x));
    x = min(x, s8_dpp_i<0x4E>(x));
    x = min(x, s8_dpp_i<0x141>(x));
    x = min(x, s8_dpp_i<0x140>(x));
    return x;
}

// Everything a row carries from one period-step to the next.  Arrays are indexed by compile-time constants only (the step is a
// template on its position in the six-step body), so they are registers.
struct S8State {
    double SE[3][5];              // SE_{t-3}, SE_{t-2}, SE_{t-1} at this lane's five residues; slot = period % 3
    double SQr[3][5], SQi[3][5];  // SQ_{t-4} .. SQ_{t-2}; slot = period % 3
    double Ep[5];                 // E of the lags of period t - 4 (last step's differences)
    cf xp[5];                     // the samples of period t - 1
    double base_e, base_qr, base_qi;   // SE / SQ at the first sample of the next period
    bool okp;                     // last step's differences were trusted in this lane
    int d1, hi;                   // first crossing (-1: none yet), last lag of the peak window
    bool done, amb;
    S8Cand best;                  // this lane's first maximum among its own lags of the window
    unsigned limit;               // the last 16-byte unit of the slot this row still needs (byte offset)
};

} // namespace

// D = steps between the last read of a ring piece and its refill: 1 keeps 9 - 10 pieces in flight, 2 keeps 7
template <int D>
__global__ __launch_bounds__(64, 2) void k_sc80(S80Params p) {
    __shared__ __align__(16) unsigned char ring[S8_PIECES * 1024];
    const int lane = threadIdx.x & 63, row = lane >> 4, lp = lane & 15, sw = (row & 1) * 8;
    const unsigned ring_lds = lds_addr(ring);
    // LDS byte offsets of this lane's five samples in a period at an even / odd position of the ring (two periods = five
    // pieces: later pairs are +5120 bytes, an immediate)
    unsigned ra[2][5];
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const unsigned b = 640u * par + 40u * lp + 8u * j;
            ra[par][j] = ((b >> 8) << 10) + 256u * row + ((b & 255u) ^ (16u * sw));
        }
    const unsigned unit_off = 16u * (unsigned)(lp ^ sw);   // which 16 bytes of a 256-byte piece this lane fetches
    const int n = p.n_lags;
    const double thr = p.threshold;
    const double thr_f = thr * (1.0 - 8.8817841970012523e-16);   // the filter's threshold: a hair lower, so that the rounding of thr * den can never hide a crossing from it
    const long long groups = (p.n_frames + 3) >> 2;

    // ring pieces whose last byte is read by step 0 .. 5 of the six-step body (first piece, count), and the pieces issued before a
    // group's first step: the whole first round except what steps 0 .. D - 1 refill
    constexpr int F0[6] = {0, 2, 5, 7, 10, 12}, FN[6] = {2, 3, 2, 3, 2, 3};
    constexpr int AHEAD = D == 1 ? 12 : 10;
    const unsigned limit0 = ((unsigned)(n + 318) * 8u) & ~15u;   // the unit that holds the last sample the last lag reads (< frame_len)
    // one piece: 256 bytes of each of the four frames, from stream byte `byte` on, into ring slot k; nothing beyond `limit` is
    // requested (the address is clamped instead: the instruction count the vmcnt logic relies on never changes)
    auto issue_at = [&](const char *sb_, unsigned rowoff_, unsigned limit_, int k, unsigned byte) {
        const unsigned want = byte + unit_off;
        glds16(sb_, rowoff_ + (want < limit_ ? want : limit_), ring_lds + 1024u * (unsigned)k);
    };
    auto row_offset = [&](long long gg) -> unsigned {
        return 4 * gg + row < p.n_frames ? (unsigned)((unsigned long long)row * (unsigned long long)p.frame_stride * 8ull) : 0u;
    };

    long long g = blockIdx.x;   // (the grid never exceeds the number of groups)
    const char *sb = reinterpret_cast<const char *>(p.in + 4 * g * p.frame_stride);   // wave-uniform
    unsigned rowoff = row_offset(g);
#pragma unroll
    for (int k = 0; k < AHEAD; ++k) issue_at(sb, rowoff, limit0, k, 256u * k);

    for (;;) {
        const long long f = 4 * g + row;
        const bool live = f < p.n_frames;

        S8State s;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 5; ++j) { s.SE[k][j] = 0.0; s.SQr[k][j] = 0.0; s.SQi[k][j] = 0.0; }
#pragma unroll
        for (int j = 0; j < 5; ++j) { s.Ep[j] = 0.0; s.xp[j] = make_float2(0.f, 0.f); }
        s.base_e = 0.0; s.base_qr = 0.0; s.base_qi = 0.0;
        s.okp = false;
        s.d1 = -1; s.hi = n - 1;
        s.done = !live; s.amb = false;
        s.best = S8Cand{-1.0, 1.0, 0.0, 0.0, INT_MAX};
        s.limit = limit0;

        unsigned round_byte = 0;   // stream byte of ring slot 0 in this round
        int t0 = 0;                // period of the body's first step
        bool all_done = false;

        // ---- one period-step; S = position in the six-step body, t = t0 + S the period whose samples it reads
        auto step = [&](auto S_) {
            constexpr int S = decltype(S_)::value;
            const int t = t0 + S;
            // this period's samples have landed; the younger pieces may still be in flight (their number is fixed by the schedule)
            s8_wait_vm<D == 1 ? 9 + (S & 1) : 7>();
            s8_fence();
            cf x[5];
#pragma unroll
            for (int j = 0; j < 5; ++j)
                x[j] = *reinterpret_cast<const cf *>(ring + ra[S & 1][j] + 5120u * (S >> 1));
            s8_wait_lgkm();
            s8_fence();
            // refill the pieces whose last byte was read D steps ago (a step of the previous round: with this round's stream)
            {
                constexpr int SF = (S - D + 6) % 6;
                const unsigned base = round_byte + (S >= D ? 3840u : 0u);
#pragma unroll
                for (int k = F0[SF]; k < F0[SF] + FN[SF]; ++k) issue_at(sb, rowoff, s.limit, k, base + 256u * k);
            }

            // e of period t, q of period t - 1, their in-lane exclusive partial sums
            double br[5], bi[5], xe[5], xr[5], xi[5];
            double te = 0.0, tr = 0.0, ti = 0.0;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                br[j] = (double)x[j].x; bi[j] = (double)x[j].y;
                const double ar = (double)s.xp[j].x, ai = (double)s.xp[j].y;
                xe[j] = te; xr[j] = tr; xi[j] = ti;
                te = fma(bi[j], bi[j], fma(br[j], br[j], te));     // exact products, one rounding per accumulation
                tr = fma(ai, bi[j], fma(ar, br[j], tr));
                ti = fma(-ai, br[j], fma(ar, bi[j], ti));
                s.xp[j] = x[j];
            }
            const double ie = s8_row_scan(te), ir = s8_row_scan(tr), ii = s8_row_scan(ti);
            const double oe = s.base_e + (ie - te), orr = s.base_qr + (ir - tr), oi = s.base_qi + (ii - ti);
            s.base_e += s8_row_last(ie); s.base_qr += s8_row_last(ir); s.base_qi += s8_row_last(ii);
            constexpr int KE = S % 3, KQ = (S + 2) % 3;   // ring slots of SE_{t-3} and SQ_{t-4} (overwritten by SE_t, SQ_{t-1})
            double En[5], Pr[5], Pi[5];
            // A difference is trusted when it is not small against the prefixes it was taken from (2^-20 of the energy seen so far:
            // 1e-8 relative on the difference), or when nothing but zeros came before the lane's lags (then it is exact).  Per lane.
            const bool zero_before = s.SE[KE][4] == 0.0;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const double se = j ? oe + xe[j] : oe, sr = j ? orr + xr[j] : orr, si = j ? oi + xi[j] : oi;
                En[j] = se - s.SE[KE][j];
                Pr[j] = sr - s.SQr[KQ][j];
                Pi[j] = si - s.SQi[KQ][j];
                s.SE[KE][j] = se; s.SQr[KQ][j] = sr; s.SQi[KQ][j] = si;
            }
            // (smallest difference by its high word: doubles of either sign order like their high words as signed integers
            // down to 2^-20 relative, which is all a trust threshold needs)
            const int ehi = min(min(min(__double2hiint(En[0]), __double2hiint(En[1])), min(__double2hiint(En[2]), __double2hiint(En[3]))), __double2hiint(En[4]));
            const bool okn = zero_before || ehi >= __double2hiint(s.base_e * 9.5367431640625e-7);

            if (t >= 4) {   // (wave-uniform) the lags of period t - 4: E = last step's differences, R = this step's
                const int d0 = 80 * (t - 4) + 5 * lp;
                // this lane's lags that can still matter are d0 .. d0 + cnt - 1 (lags <= hi; hi = n - 1 until the crossing is known)
                int cnt = s.hi + 1 - d0;
                cnt = s.done ? 0 : (cnt < 0 ? 0 : (cnt > 5 ? 5 : cnt));
                const bool ok = s.okp && okn;
                double num[5], den[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) { num[j] = Pr[j] * Pr[j] + Pi[j] * Pi[j]; den[j] = s.Ep[j] * En[j]; }
                if (__builtin_amdgcn_ballot_w64(cnt > 0 && !ok)) {
                    // (rare) a lane whose sums are small against the prefixes: its lags are safe to pass over only if, with the
                    // prefixes' worst-case rounding error dl on every sum, they still cannot cross the threshold / beat this lane's best
                    const double dl = s.base_e * (double)(t + 32) * 2.220446049250313e-16;
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const double el = s.Ep[j] - dl, rl = En[j] - dl;
                        const double nub = num[j] + 4.0 * dl * (fabs(Pr[j]) + fabs(Pi[j]) + dl);
                        const bool safe = el > 0.0 && rl > 0.0 && (s.d1 < 0 ? nub < thr * (el * rl) : nub * s.best.den < s.best.num * (el * rl));
                        if (j < cnt && !ok && !safe) s.amb = true;
                    }
                }
                if (__builtin_amdgcn_ballot_w64(cnt > 0 && s.d1 < 0)) {   // some row is still looking for its crossing
                    // filter: num - thr den >= 0 at any of the five lags (sign bits of five fused multiply-adds); the block below decides
                    // exactly, also about lags beyond cnt and all-zero windows (0 >= 0), which the filter lets through
                    int sg = -1;
#pragma unroll
                    for (int j = 0; j < 5; ++j) sg &= __double2hiint(fma(-thr_f, den[j], num[j]));
                    const bool any = sg >= 0 && ok && s.d1 < 0 && cnt > 0;
                    if (__builtin_amdgcn_ballot_w64(any)) {   // (about once per frame) a row's first crossing may be in this step
                        int mine = INT_MAX;
#pragma unroll
                        for (int j = 4; j >= 0; --j) {
                            const double td = thr * den[j];
                            if (j < cnt && num[j] >= td && td > 0.0) mine = d0 + j;
                        }
                        const int c = s8_row_min(any ? mine : INT_MAX);
                        if (s.d1 < 0 && c != INT_MAX) {
                            s.d1 = c;
                            s.hi = c + 240 < n - 1 ? c + 240 : n - 1;
                            s.limit = ((unsigned)(s.hi + 319) * 8u) & ~15u;   // nothing beyond lag hi's window is needed any more
                        }
                    }
                }
                if (__builtin_amdgcn_ballot_w64(cnt > 0 && s.d1 >= 0)) {   // some row is inside its peak window
                    // this lane's lags of the window: from the crossing on (only the crossing's own step has lo > 0)
                    int lo = s.d1 - d0;
                    lo = lo < 0 ? 0 : lo;
                    const bool live_lane = ok && s.d1 >= 0;
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const bool beats = num[j] * s.best.den > s.best.num * den[j];
                        if (live_lane && j < cnt && j >= lo && beats) s.best = S8Cand{num[j], den[j], Pr[j], Pi[j], d0 + j};
                    }
                }
                const int last = 80 * (t - 4) + 79;   // every lag up to here has been judged
                if (last >= s.hi) s.done = true;
                all_done = __builtin_amdgcn_ballot_w64(!s.done) == 0ull;
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) s.Ep[j] = En[j];
            s.okp = okn;
        };

#pragma clang loop unroll(disable)
        for (;;) {
            step(std::integral_constant<int, 0>{}); if (all_done) break;
            step(std::integral_constant<int, 1>{}); if (all_done) break;
            step(std::integral_constant<int, 2>{}); if (all_done) break;
            step(std::integral_constant<int, 3>{}); if (all_done) break;
            step(std::integral_constant<int, 4>{}); if (all_done) break;
            step(std::integral_constant<int, 5>{}); if (all_done) break;
            t0 += 6; round_byte += 3840u;
        }

        // ---- the next group's first pieces go out before this group's verdict is reduced and stored.  What this group still has
        //      in flight lands first (clamped addresses once the crossing is known: cache hits), so that no older piece can arrive
        //      in a ring slot after a newer one.
        const long long gn = g + gridDim.x;
        s8_wait_vm<0>();
        s8_fence();
        if (gn < groups) {
            sb = reinterpret_cast<const char *>(p.in + 4 * gn * p.frame_stride);
            rowoff = row_offset(gn);
#pragma unroll
            for (int k = 0; k < AHEAD; ++k) issue_at(sb, rowoff, limit0, k, 256u * k);
        }
        // ---- the row's verdict: first maximum over its lanes; a row that met an untrusted lag goes to the slow list
        const S8Cand b = s8_row_best(s.best);
        const unsigned long long ambm = __builtin_amdgcn_ballot_w64(s.amb);
        const bool row_amb = ((ambm >> (16 * row)) & 0xFFFFull) != 0ull;
        if (lp == 0 && live) {
            if (row_amb) {
                p.d_hat[f] = -1;
                p.slow_list[atomicAdd(p.slow_count, 1)] = (int32_t)f;
            } else if (b.lag == INT_MAX) {
                p.d_hat[f] = -1;
            } else {
                p.d_hat[f] = b.lag;
                p.exact[f] = ScExact{b.pr, b.pi, b.num, b.den};
            }
        }
        if (gn >= groups) break;
        g = gn;
    }
    s8_wait_vm<0>();
}

// L = 80, W = 240, frames of even length (LDS-DMA in 16-byte units), four frames' row offsets in 32 bits
bool sc80_ok(const ScParams &p) {
    if (p.mode != 0 || p.L != 80 || p.W != 240 || p.tiles_per_frame != 1) return false;
    // (LDS-DMA takes any 4-byte aligned global address -- tools/lab/glds_align.hip --, so rows need only their natural 8-byte alignment and
    //  any stride.  The stream is clamped in 16-byte units counted from the frame's first sample, so the kernel is given an EVEN slot length:
    //  an odd one is rounded up -- the extra sample is read, never used: every lag's window ends at or before the true last sample -- when
    //  that sample is mapped, i.e. when rows have slack; ofdm_abi_sc_run peels the last frame off a batch of tight odd rows)
    if ((reinterpret_cast<uintptr_t>(p.in) & 7) != 0) return false;
    if ((p.frame_len & 1) != 0 && !(p.tail_mapped || (p.n_frames > 1 ? p.frame_stride > p.frame_len : false))) return false;
    if (p.n_lags <= 0 || p.n_lags + 319 > p.frame_len || p.frame_len > (1LL << 27)) return false;
    if (p.n_frames > 1 && (p.frame_stride <= 0 || p.frame_stride > (1LL << 27))) return false;
    return p.threshold > 0.0;
}

// k_sc80 handles a frame in ONE row of one wavefront (~0.2 Gsamples/s per row): right for batches, wrong for a handful of very long
// captures, which the multi-tile search spreads over the whole chip.  Slots up to the one-tile size of the filter pair always take it
// (as before round 5's extension); longer slots when the batch fills at least a sixteenth of the chip's 8 192 rows.
bool sc80_wanted(const ScParams &p) {
    return sc80_ok(p) && (p.n_lags + p.W + p.L <= 2560 || p.n_frames >= 512);
}

hipError_t launch_sc80(const ScParams &p, ScExact *exact, int32_t *slow_list, int32_t *slow_count, int num_cu, hipStream_t st) {
    if (p.n_frames <= 0) return hipSuccess;
    S80Params q;
    q.in = p.in; q.n_frames = p.n_frames; q.frame_stride = p.n_frames > 1 ? p.frame_stride : 0; q.frame_len = p.frame_len + (p.frame_len & 1);
    q.n_lags = (int)p.n_lags; q.threshold = p.threshold;
    q.d_hat = p.d_hat; q.exact = exact; q.slow_list = slow_list; q.slow_count = slow_count;
    const Tuning &tu = tuning_or_default(p.tune);
    long long grid = (long long)num_cu * 8;   // one wavefront per workgroup, two per SIMD; 15 KB of LDS each
    if (tu.grid_cap > 0 && grid > tu.grid_cap) grid = tu.grid_cap;
    const long long groups = (p.n_frames + 3) / 4;
    if (grid > groups) grid = groups;
    trace_add(p.trace, "k_sc80");
    if (tu.sc80_depth == 2) hipLaunchKernelGGL(k_sc80<2>, dim3((unsigned)grid), dim3(64), 0, st, q);
    else hipLaunchKernelGGL(k_sc80<1>, dim3((unsigned)grid), dim3(64), 0, st, q);
    return hipGetLastError();
}

} // namespace ofdm
