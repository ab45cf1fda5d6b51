// kernels_mid.hip -- symbol-stream RX demod and TX for the transform lengths between the two headline shapes:
// N = 64 R, R in {2, 4, 8, 16, 32} (EXT-4: N = 128 .. 2048), as an R x 64 two-stage transform, the layout k_demod4096 /
// k_tx4096 (kernels_fast.hip) use for R = 64:
//     X[c + R d] = sum_b W64^(b d) * [ W_N^(b c) * sum_a x[64 a + b] W_R^(a c) ]          a, c < R;  b, d < 64
//   stage A  the R-point transform over a for each of the 64 columns b, straight from HBM (64 consecutive columns = 512
//            contiguous bytes per row a), eight points per lane:
//              R <  8   a lane owns 8/R whole columns: radix-R butterflies in registers;
//              R =  8   one column per lane: one radix-8 butterfly;
//              R >  8   Q = R/8 adjacent lanes share a column: radix-8 in registers, W_R^(u j), radix-Q across the lanes by DPP
//                       quad permutes (no LDS);
//   twiddle  W_N^(b c): eight loop-invariant registers per lane;
//   transpose through LDS: row slot rho(c) x column b, 72-sample rows;
//   stage B  the FFT64 over b for each row c: 8 lanes x 8 points, wave-local (the k_demod64 layout);
//   epilogue as k_demod4096: the carrier class of bin c + R d is class64(d) (src/receiver.rs:122-133 tiled N/64 times), so a
//            lane's eight bins d = t + 8 q have the same classes for every row.
// A symbol occupies 8 R lanes in both stages; a 256-thread workgroup carries 32 / R symbols per step.  For R <= 8 a symbol
// lives inside one wavefront and every exchange is ordered by the in-order LDS pipe (a wavefront fence, no barrier).
// Roofline: HBM, one read of every sample after the prefix (RX) / one write of every sample (TX); see DESIGN.md section 5.4.
#include "device_common.hpp"
#include "kernels.hpp"
#include <stdlib.h>
#include <type_traits>

extern "C" __device__ float __ocml_atan2pi_f32(float, float); // atan2(y, x) / pi (ROCm device library)

namespace ofdm {

namespace {

template <int CTRL> __device__ __forceinline__ float dppq(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ cf dppq_cf(cf a) { return make_float2(dppq<CTRL>(a.x), dppq<CTRL>(a.y)); }

template <int R> struct Mid {
    static constexpr int N = 64 * R, CP = N / 4, S = N + CP;
    static constexpr int LPS = 8 * R;               // lanes per symbol
    static constexpr int WG = 256;
    static constexpr int G = WG / LPS;              // symbols per workgroup step
    static constexpr int Q = R >= 8 ? R / 8 : 1;    // lanes per stage-A column
    static constexpr int P = R >= 8 ? 1 : 8 / R;    // stage-A columns per lane
    static constexpr int TS = 72, SLAB = 8 * 72;
    // waves per SIMD (= workgroups per CU) the kernels are built for: the RX kernel for R >= 16 keeps 14 more constant
    // registers and spills 10-20 of them under the 128-VGPR limit of 4 waves (measured: N = 1024 RX 0.30 -> 0.45 of the
    // roofline at 3 waves, TX 0.49 -> 0.44 -- TX has no spills at 4)
    static constexpr int OCC_RX = R >= 16 ? 3 : 4, OCC_TX = 4;
    // Row slot of output row c = e + 8 u (stage-A output e of the column's lane u) in the transpose buffer.  A stage-A store has the
    // lanes of a half wavefront write one row slot each per lane group u, 32 / Q consecutive columns wide; rows are 72 samples = 144
    // banks apart, i.e. 16 banks per slot (mod 64), so the slots of the Q groups must differ by 64 / Q banks: Q = 4: adjacent slots
    // (u), Q = 2: slots 2 apart (round 3 used adjacent slots for Q = 2 as well: the second half of group 0's 16 columns shared its
    // banks with the first half of group 1's -- a 2-way conflict on every stage-A store of N = 1024).
    __device__ static int slot_of_row(int c) {
        if (R < 8) return c;
        const int e = c & 7, u = c >> 3;
        return Q == 2 ? 2 * u + (e & 1) + 4 * (e >> 1) : e * Q + u;
    }
    __device__ static constexpr int slot_of_eu(int e, int u) { return Q == 2 ? 2 * u + (e & 1) + 4 * (e >> 1) : e * Q + u; }
    __device__ static int row_of_slot(int rho) {
        if (R < 8) return rho;
        return Q == 2 ? ((rho & 1) + 2 * (rho >> 2)) + 8 * ((rho >> 1) & 1) : rho / Q + 8 * (rho % Q);
    }
    // Where sample n of a symbol sits in the SECOND transpose of the TX kernels (sample order, so that the stores are 16 bytes per
    // lane): row n / 64, column n % 64 -- for R >= 16 with the column XOR-ed by 2 (row mod 4) and by 4 (column / 32).  A lane writes
    // n = cB + R (t + 8 q); unswizzled, the bank of that store depends on t / 2 and on two bits of cB only, and they ADD: 8 lanes of a
    // half wavefront per bank pair at R = 32 (PMC round 3: 53 % of k_tx_mid<32>'s LDS cycles were conflicts), 4 at R = 16.  The XORs
    // put the row into bits 1-2 and the upper half of the row into bit 2: conflict-free at R = 16, 2-way at R = 32 (all of a half
    // wavefront's columns are even there: 16 bank pairs for 32 lanes is the floor).  Both XORs are even, so sample pairs stay
    // adjacent and 16-byte aligned for the float4 reads of the store loop.
    __device__ static int t2_index(int n) {
        const int row = n >> 6, col = n & 63;
        return row * TS + (R >= 16 ? col ^ (2 * (row & 3)) ^ (4 * (col >> 5)) : col);
    }
};

template <int LPS> __device__ __forceinline__ void symbol_sync() {
    if (LPS > 64) __syncthreads();
    else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// Stage A on the eight points of one lane.  In: R >= 8: v[m] = x[64 (u + Q m) + b];  R < 8: v[i R + a] = x[64 a + b_i].
// Out: R >= 8: v[j] = Y_b[j + 8 u];  R < 8: v[i R + c] = Y_{b_i}[c]   (before the W_N^(b c) twiddle).
template <int R, bool INV> __device__ __forceinline__ void stage_a(cf *v, const cf *tA, int u) {
    constexpr int Q = Mid<R>::Q;
    if (R == 1) return; // N = 64 (TX only): the symbol is a single row
    if (R == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const cf a = v[2 * i], b = v[2 * i + 1]; v[2 * i] = cadd(a, b); v[2 * i + 1] = csub(a, b); }
    } else if (R == 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            cf *x = v + 4 * i;
            const cf s0 = cadd(x[0], x[2]), d0 = csub(x[0], x[2]), s1 = cadd(x[1], x[3]), d1 = mul_mj<INV>(csub(x[1], x[3]));
            x[0] = cadd(s0, s1); x[2] = csub(s0, s1); x[1] = cadd(d0, d1); x[3] = csub(d0, d1);
        }
    } else {
        bfly8<INV>(v);
        if (Q > 1) {
#pragma unroll
            for (int j = 1; j < 8; ++j) v[j] = cmul(v[j], tA[j - 1]); // W_R^(u j)
        }
        if (Q == 2) { // radix-2 across the lane pair: Y_e = Z_0 + (-1)^e Z_1
            const float sg = u ? -1.f : 1.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const cf o = dppq_cf<0xB1>(v[j]); // quad_perm [1,0,3,2]
                v[j] = make_float2(fmaf(sg, v[j].x, o.x), fmaf(sg, v[j].y, o.y));
            }
        } else if (Q == 4) { // radix-4 across the quad: Y_e = (Z_0 + (-1)^e Z_2) + W4^e (Z_1 + (-1)^e Z_3)
            const float s2 = (u & 1) ? -1.f : 1.f;
            const float rx = u == 0 ? 1.f : (u == 2 ? -1.f : 0.f);
            float ry = u == 1 ? -1.f : (u == 3 ? 1.f : 0.f);
            if (INV) ry = -ry;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const cf z0 = dppq_cf<0x00>(v[j]), z1 = dppq_cf<0x55>(v[j]), z2 = dppq_cf<0xAA>(v[j]), z3 = dppq_cf<0xFF>(v[j]);
                const cf pe = make_float2(fmaf(s2, z2.x, z0.x), fmaf(s2, z2.y, z0.y));
                const cf po = make_float2(fmaf(s2, z3.x, z1.x), fmaf(s2, z3.y, z1.y));
                v[j] = make_float2(fmaf(-ry, po.y, fmaf(rx, po.x, pe.x)), fmaf(ry, po.x, fmaf(rx, po.y, pe.y)));
            }
        }
    }
}

// Stage B: FFT64 over the eight lanes of a row through the wave's XOR-swizzled slab (in: v[m] = Z[t + 8 m]; out: v[q] = X[t + 8 q])
template <bool INV> __device__ __forceinline__ void stage_b(cf *v, cf *buf, int t, int wr, const cf *w) {
    bfly8<INV>(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) buf[wr ^ r] = v[r];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = buf[8 * m + (t ^ m)];
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], w[r - 1]);
    bfly8<INV>(v);
}

struct MidRxParams {
    const float2 *in;
    long long frame_stride;
    long long total;          // symbols
    int syms_per_frame, first_symbol;
    long long step_f;         // frames / symbols one grid step advances
    int step_k;
    const float2 *tw;         // exp(-2 pi i m / N), m < N
    const float2 *hk;         // optional channel, hk_stride = 0 (shared) or N (per frame)
    long long hk_stride;
    unsigned char *out;
    long long out_stride;
    // FRAME = true (the decode chain after timing, src/receiver.rs:20-83): per-frame start of the trimmed frame, CFO and live
    // symbol count; samples at or past frame_len read as zero (pad_chunk, receiver.rs:203-210)
    const int32_t *offset;
    const double *f_delta;
    const int32_t *nsym_frame;
    long long frame_len;
};

template <int R, int BPS, bool GUARD, bool FRAME>
__global__ __launch_bounds__(256, Mid<R>::OCC_RX - (FRAME ? 1 : 0)) void k_demod_mid(MidRxParams p) { // frame mode carries offsets, CFO phasors and the
    // zero-fill masks of its branch-free fetch: at the stream kernel's occupancy it spills 4-16 registers, and a spill reload waits for the prefetch
    typedef Mid<R> M;
    constexpr int N = M::N, S = M::S, CP = M::CP, LPS = M::LPS, G = M::G, Q = M::Q, TS = M::TS;
    constexpr int ND = GUARD ? 48 * R : N;
    constexpr int IMG_DW = ND * BPS / 32;           // packed bytes of one symbol, in dwords
    constexpr int nbytes = ND * BPS / 8;
    __shared__ cf slab_all[4 * M::SLAB];            // [4 waves][8 x 72] FFT64 transpose slabs (stage B)
    __shared__ cf T[32 * TS];                       // [G symbols x R row slots][72]
    __shared__ unsigned img[G * IMG_DW];            // the symbols' packed output images
    __shared__ float red[4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = tid / LPS, l = tid % LPS;
    // stage A role
    const int u = R >= 8 ? l % Q : 0;
    const int colA = R >= 8 ? l / Q : l;            // R < 8: columns colA + LPS i, i < 8 / R
    // stage B role
    const int t = tid & 7, rs = tid >> 3;           // rs: row slot among the workgroup's 32
    const int cB = M::row_of_slot(rs % R);
    cf *buf = slab_all + wave * M::SLAB + (lane >> 3) * 72;
    const int wr = swz(8 * t);

    // W64^(r t), r = 1 .. 7: a 56-entry LDS table read at use instead of fourteen loop-invariant registers -- the instantiations that ran
    // out of registers reloaded their spills from scratch, a VMEM load queued behind the next symbol's prefetch (round-5 ISA scan)
    // (R >= 16 runs at three waves per SIMD and never spilled: there the registers stay -- the table's reads cost it 5 % when tried)
    constexpr bool W_LDS = R < 16;
    __shared__ cf wtab[56];
    cf wreg[7];
    if (W_LDS) {
        if (tid < 56) wtab[tid] = p.tw[R * (tid % 7 + 1) * (tid / 7)];
        __syncthreads();
    } else {
#pragma unroll
        for (int r = 1; r < 8; ++r) wreg[r - 1] = p.tw[R * r * t];
    }
    const cf *w = W_LDS ? wtab + 7 * t : wreg;
    cf tA[7];                                       // W_R^(u j)
#pragma unroll
    for (int j = 1; j < 8; ++j) tA[j - 1] = Q > 1 ? p.tw[64 * u * j] : make_float2(1.f, 0.f);
    cf z[8];                                        // W_N^(b c) of stage-A output e
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = R >= 8 ? p.tw[colA * (e + 8 * u)] : p.tw[(colA + LPS * (e / R)) * (e % R)];
    int boff[8];                                    // bit offset of bin cB + R d (d = t + 8 q) in the image, -1 = not a data bin
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int d = t + 8 * q;
        boff[q] = carrier_class64(d, GUARD) == 0 ? ((GUARD ? data_classes_below64(d) : d) * R + cB) * BPS : -1;
    }
    unsigned *myimg = img + g * IMG_DW;

    // (frame, symbol) of this lane's symbol, one step ahead (the prefetch) and now; advanced without divisions
    long long fn; int kn;
    {
        const long long sg0 = (long long)blockIdx.x * G + g;
        fn = sg0 / p.syms_per_frame; kn = (int)(sg0 - fn * p.syms_per_frame);
    }
    // Loads are issued without a branch (a load under `if (in range)` is followed by s_waitcnt vmcnt(0) at the join, i.e. is
    // synchronous): out-of-range elements read the twiddle table instead and are zeroed when they leave the prefetch registers.
    // room = samples from this lane's first one to the end of the capture (FRAME), 0 past the batch.
    auto elem = [&](int e) -> int { return R >= 8 ? 64 * (u + Q * e) : 64 * (e % R) + LPS * (e / R); }; // stage-A input e, relative to the lane's first sample
    // FRAME: the per-frame scalars (trimmed start, live-symbol count, CFO) of a step are REQUESTED two steps ahead and taken one step
    // ahead (FS t2 -> s1): read where they were used -- offset, then the samples; the symbol count; the CFO; the channel, one after the
    // other, each under its own condition -- every one of them was followed by s_waitcnt vmcnt(0), i.e. four serialized round trips to
    // memory per step with the sample prefetch drained by the first (round-5 ISA scan; the stream mode of this kernel has none of them
    // and ran twice as fast per symbol).  All loads are unconditional, from addresses that are always mapped.
    struct FS { int off, ns; double fd; };
    auto load_scalars = [&](bool in, long long fr_) -> FS {
        const long long fr = in ? fr_ : 0;
        FS r;
        r.off = p.offset ? p.offset[fr] : 0;
        r.ns = p.nsym_frame ? p.nsym_frame[fr] : p.syms_per_frame;
        r.fd = p.f_delta ? p.f_delta[fr] : 0.0;
        return r;
    };
    auto fetch = [&](long long sg, long long fr_, int kk, long long off_, cf *dst, int &room) {
        const bool in = sg < p.total;
        const long long fr = in ? fr_ : 0;
        const long long off = FRAME ? off_ : 0;
        const long long n0 = off + (long long)(p.first_symbol + kk) * S + CP + colA; // first sample of this lane, inside the frame
        const cf *src = p.in + fr * p.frame_stride + n0;
        long long rm = FRAME ? p.frame_len - n0 : (long long)N;
        rm = in ? rm : 0;
        room = (int)(rm < 0 ? 0 : (rm > N ? N : rm));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i = elem(e);
            const cf *a = FRAME ? (i < room ? src + i : p.tw + i) : (in ? src : p.tw) + i; // p.tw: N mapped entries
            dst[e] = *a;
        }
    };
    // The image of step j leaves for HBM in step j + 1, behind that step's first synchronisation (see k_demod4096).
    auto flush = [&](unsigned *dst) {
#pragma unroll
        for (int i = l; i < IMG_DW; i += LPS) { dst[i] = myimg[i]; myimg[i] = 0u; }
    };
    for (int i = l; i < IMG_DW; i += LPS) myimg[i] = 0u;
    cf pre[8];
    int room_pre = 0;
    unsigned *pending = nullptr;
    const long long stride = (long long)gridDim.x * G;
    auto advance = [&](long long &ff, int &kk) { ff += p.step_f; kk += p.step_k; if (kk >= p.syms_per_frame) { kk -= p.syms_per_frame; ++ff; } };
    // positions of this lane's symbol: now (f0, k0), one step ahead (fn, kn: the sample prefetch), two ahead (f2, k2: the scalar prefetch)
    long long f0 = fn, f2; int k0 = kn, k2;
    advance(fn, kn);
    f2 = fn; k2 = kn; advance(f2, k2);
    const long long sg_first = (long long)blockIdx.x * G + g;
    FS tq = FS{0, 0, 0.0};                          // scalars of the step one ahead, in flight since the step before
    int ns_cur = 0; double fd_cur = 0.0;            // of the step being consumed
    {
        FS s0 = FS{0, 0, 0.0};
        if (FRAME) s0 = load_scalars(sg_first < p.total, f0);
        fetch(sg_first, f0, k0, s0.off, pre, room_pre);
        ns_cur = s0.ns; fd_cur = s0.fd;
        if (FRAME) tq = load_scalars(sg_first + stride < p.total, fn);
    }

    for (long long base = (long long)blockIdx.x * G; base < p.total; base += stride) {
        const long long sg = base + g;
        const long long f = f0; const int k = k0;
        cf v[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = (!FRAME || elem(m) < room_pre) ? pre[m] : make_float2(0.f, 0.f);
        const FS sn = tq;                            // requested a step ago, behind that step's samples: here by now
        // FRAME: the channel of THIS step's frame goes out first (its use, behind stage B, then waits for nothing younger), then the next
        // step's samples from the start offset that has just arrived, then the scalars of the step after that
        cf hkv[8];
        if (FRAME) {
            const cf *h = p.hk ? p.hk + (sg < p.total ? f : 0) * p.hk_stride + cB : p.tw;
#pragma unroll
            for (int q = 0; q < 8; ++q) hkv[q] = h[R * (t + 8 * q)];
        }
        fetch(sg + stride, fn, kn, sn.off, pre, room_pre);
        if (FRAME) tq = load_scalars(sg + 2 * stride < p.total, f2);
        bool live = sg < p.total;
        if (FRAME) {
            if (live && k >= ns_cur) live = false; // fewer symbols in this frame (short capture / failed sync): nothing is written
            if (!live) {
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = make_float2(0.f, 0.f);
            } else if (p.f_delta) { // CFO derotation, sample ids count from the trimmed start (receiver.rs:44-50); phase reduced in f64
                const double turns = fd_cur * 0.15915494309189533577; // 1 / (2 pi)
                const long long n0 = (long long)(p.first_symbol + k) * S + CP + colA;
                if (R >= 8) {
                    cf ph = cfo_phasor(turns, n0 + 64 * u);
                    const cf st = cfo_phasor(turns, 64 * Q);
#pragma unroll
                    for (int m = 0; m < 8; ++m) { v[m] = cmul(v[m], ph); ph = cmul(ph, st); }
                } else {
                    const cf st = cfo_phasor(turns, 64);
#pragma unroll
                    for (int i = 0; i < 8 / R; ++i) {
                        cf ph = cfo_phasor(turns, n0 + LPS * i);
#pragma unroll
                        for (int a = 0; a < R; ++a) { v[i * R + a] = cmul(v[i * R + a], ph); ph = cmul(ph, st); }
                    }
                }
            }
        }
        // rotate the pipeline: positions move up one step, the next step's symbol count and CFO are kept for it
        f0 = fn; k0 = kn; fn = f2; kn = k2; advance(f2, k2);
        ns_cur = sn.ns; fd_cur = sn.fd;
        unsigned *const mine = live ? reinterpret_cast<unsigned *>(p.out + f * p.out_stride + (long long)k * nbytes) : nullptr;
        // ---- stage A, twiddle, transpose
        stage_a<R, false>(v, tA, u);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int slot = R >= 8 ? M::slot_of_eu(e, u) : e % R; // = slot_of_row(e + 8 u), with e a compile-time constant
            const int col = R >= 8 ? colA : colA + LPS * (e / R);
            T[(g * R + slot) * TS + col] = (R < 8 && e % R == 0) ? v[e] : cmul(v[e], z[e]);
        }
        symbol_sync<LPS>(); // T complete; the PREVIOUS step's image complete (two synchronisations per step, not three: as k_demod4096)
        if (pending) flush(pending); // ... so it leaves for HBM here and is cleared for this step's fields, which are written behind the next one
        pending = mine;
        // ---- stage B: FFT64 over b for row cB
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = T[rs * TS + t + 8 * m];
        stage_b<false>(v, buf, t, wr, w);
        // v[q] = X[cB + R (t + 8 q)]
        if (p.hk) { // equalise: Y /= H (src/receiver.rs:68-70)
            const cf *h = p.hk + f * p.hk_stride;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const cf hh = FRAME ? hkv[q] : h[cB + R * (t + 8 * q)];
                const float rn = __builtin_amdgcn_rcpf(hh.x * hh.x + hh.y * hh.y);
                const cf e = cmulc(v[q], hh);
                v[q] = make_float2(e.x * rn, e.y * rn);
            }
        }
        cf rot = make_float2(1.f, 0.f);
        if (GUARD) { // decode_block (src/receiver.rs:106-145): mean angle of the 4 R pilots, rotate by -phase
            // pilot classes 6, 25, 39, 58 = (t, q) = (6, 0), (1, 3), (7, 4), (2, 7); other lanes feed (1, 0) -> angle 0
            cf pv = make_float2(1.f, 0.f);
            pv = (t == 6) ? v[0] : pv;
            pv = (t == 1) ? v[3] : pv;
            pv = (t == 7) ? v[4] : pv;
            pv = (t == 2) ? v[7] : pv;
            float a = __ocml_atan2pi_f32(pv.y, pv.x);
            constexpr int WL = LPS < 64 ? LPS : 64;
#pragma unroll
            for (int sh = WL / 2; sh >= 1; sh >>= 1) a += __shfl_xor(a, sh, 64);
            if (LPS > 64) {
                if (lane == 0) red[wave] = a;
                __syncthreads();
                a = 0.f;
#pragma unroll
                for (int i = 0; i < LPS / 64; ++i) a += red[g * (LPS / 64) + i];
            }
            const float trn = a * (0.5f / (4.0f * R)); // mean of the 4 R pilot angles, in turns -> hardware sin / cos
            rot = make_float2(__builtin_amdgcn_cosf(trn), -__builtin_amdgcn_sinf(trn)); // applied inside the demapper
        }
        // Fields go into the image behind the pilot barrier, which also orders them after the flush above; where there is none
        // (no guard bands, or a symbol inside one wavefront) one stands here.  Nothing closes the step: the image is read only behind
        // the next step's first synchronisation, T is rewritten only by wavefronts that are past this one (every stage-B read of T
        // lies before it), the pilot sums only behind the next step's first.
        if (!GUARD || LPS <= 64) symbol_sync<LPS>();
        // demodulate (src/receiver.rs:147-190) and pack LSB-first (src/utils.rs:30-36): OR every field into the image
        // A dead symbol (past the batch, or k >= nsym_frame[f]) must leave the image untouched: nothing flushes (and clears) it
        // after such a step, and demap_point(0) is not 0 for BPS >= 2 -- stale bits would be OR-ed into the next live symbol.
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (live && boff[q] >= 0) {
                const unsigned idx = GUARD ? demap_point_rot(v[q], rot, BPS) : demap_point(v[q], BPS);
                const int wd = boff[q] >> 5, sh = boff[q] & 31;
                atomicOr(&myimg[wd], idx << sh);
                if (BPS > 1 && (32 % BPS) != 0) { // a field may straddle two dwords (only for 6-bit fields)
                    if (sh + BPS > 32) atomicOr(&myimg[wd + 1], idx >> (32 - sh));
                }
            }
        }
    }
    symbol_sync<LPS>(); // the last image is complete
    if (pending) flush(pending);
}

// Persistent grid: OCC resident workgroups per CU (39 KB of LDS each).  Tuning::grid_cap caps it (test hook: a small
// grid makes every workgroup run many steps of the prefetch / deferred-store pipeline on a small batch).
static long long mid_grid(long long steps, int num_cu, int occ, long long cap) {
    long long grid = (long long)num_cu * occ;
    if (cap > 0 && cap < grid) grid = cap;
    return grid > steps ? steps : grid;
}

template <int R, int BPS, bool GUARD> hipError_t launch_demod_mid(const MidRxParams &p0, bool frame, hipStream_t st, int num_cu, long long cap) {
    MidRxParams p = p0;
    constexpr int G = Mid<R>::G;
    const long long steps = (p.total + G - 1) / G;
    const long long grid = mid_grid(steps, num_cu, Mid<R>::OCC_RX - (frame ? 1 : 0), cap);
    const long long adv = grid * G;
    p.step_f = adv / p.syms_per_frame;
    p.step_k = (int)(adv - p.step_f * p.syms_per_frame);
    if (frame) hipLaunchKernelGGL((k_demod_mid<R, BPS, GUARD, true>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_demod_mid<R, BPS, GUARD, false>), dim3((unsigned)grid), dim3(256), 0, st, p);
    return hipGetLastError();
}
template <int R> hipError_t dispatch_demod_mid(const MidRxParams &p, int bps, bool guard, bool frame, hipStream_t st, int num_cu, long long cap) {
    switch (bps * 2 + (guard ? 1 : 0)) {
    case 2: return launch_demod_mid<R, 1, false>(p, frame, st, num_cu, cap);
    case 3: return launch_demod_mid<R, 1, true>(p, frame, st, num_cu, cap);
    case 4: return launch_demod_mid<R, 2, false>(p, frame, st, num_cu, cap);
    case 5: return launch_demod_mid<R, 2, true>(p, frame, st, num_cu, cap);
    case 8: return launch_demod_mid<R, 4, false>(p, frame, st, num_cu, cap);
    case 9: return launch_demod_mid<R, 4, true>(p, frame, st, num_cu, cap);
    case 12: return launch_demod_mid<R, 6, false>(p, frame, st, num_cu, cap);
    case 13: return launch_demod_mid<R, 6, true>(p, frame, st, num_cu, cap);
    case 16: return launch_demod_mid<R, 8, false>(p, frame, st, num_cu, cap);
    case 17: return launch_demod_mid<R, 8, true>(p, frame, st, num_cu, cap);
    }
    return hipErrorNotSupported;
}

// ---------------------------------------------------------------------------------------------------------------
// k_tx_mid: modulate + encode_block + prefix_block (src/transmitter.rs:108-181) for a continuous stream of N = 64 R
// symbols, the mirror image of k_demod_mid:
//     x[c + R d] = 1/N sum_b W64^(-b d) * [ W_N^(-b c) * sum_a X[64 a + b] W_R^(-a c) ]
// Bin 64 a + b has carrier class class64((64 a + b) / R) and, among the data bins, ordinal dcb(class) R + b mod R.
struct MidTxParams {
    const uint8_t *bytes;
    long long n_bytes, n_sym;
    const float2 *tw;   // exp(-2 pi i m / N)
    float2 *out;        // n_sym x (N + N/4) samples
    int bps;
};

template <int R, bool GUARD>
__global__ __launch_bounds__(256, Mid<R>::OCC_TX) void k_tx_mid(MidTxParams p) {
    typedef Mid<R> M;
    constexpr int N = M::N, S = M::S, CP = M::CP, LPS = M::LPS, G = M::G, Q = M::Q, TS = M::TS;
    constexpr int ND = GUARD ? 48 * R : N;
    constexpr int SB_DW = 16 * R + 2;               // a symbol's bytes as dwords (<= 64 R bytes) + slack for the two-byte window
    __shared__ cf slab_all[4 * M::SLAB];
    __shared__ __align__(16) cf T[32 * TS];
    __shared__ unsigned sbw_all[G * SB_DW];
    __shared__ float lvl[16];                       // axis levels by raw bit field (transmitter.rs:108-140)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = tid / LPS, l = tid % LPS;
    const int u = R >= 8 ? l % Q : 0;
    const int colA = R >= 8 ? l / Q : l;
    const int t = tid & 7, rs = tid >> 3;
    const int cB = M::row_of_slot(rs % R);
    cf *buf = slab_all + wave * M::SLAB + (lane >> 3) * 72;
    const int wr = swz(8 * t);
    unsigned *sbw = sbw_all + g * SB_DW;
    cf *Tsym = T + g * R * TS;

    if (tid < 16) lvl[tid] = p.bps > 1 && tid < (1 << (p.bps >> 1)) ? axis_level((unsigned)tid, p.bps >> 1) : 0.f;
    const unsigned char *sb = reinterpret_cast<const unsigned char *>(sbw);
    __shared__ cf wtab[56];                         // conj W64^(r t), r = 1 .. 7: read at use (fourteen registers less: see k_demod_mid)
    if (tid < 56) { const cf x = p.tw[R * (tid % 7 + 1) * (tid / 7)]; wtab[tid] = make_float2(x.x, -x.y); }
    __syncthreads();
    const cf *w = wtab + 7 * t;
    cf tA[7];
#pragma unroll
    for (int j = 1; j < 8; ++j) { const cf x = Q > 1 ? p.tw[64 * u * j] : make_float2(1.f, 0.f); tA[j - 1] = make_float2(x.x, -x.y); }
    cf z[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const cf x = R >= 8 ? p.tw[colA * (e + 8 * u)] : p.tw[(colA + LPS * (e / R)) * (e % R)];
        z[e] = make_float2(x.x, -x.y);
    }
    int boff[8]; // stage-A input e: bit offset of its bin inside the symbol's stream, -1 = null, -2 = pilot
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int a = R >= 8 ? u + Q * e : e % R;
        const int b = R >= 8 ? colA : colA + LPS * (e / R);
        const int bin = 64 * a + b, kc = bin / R, cls = carrier_class64(kc, GUARD);
        boff[e] = cls == 0 ? (GUARD ? data_classes_below64(kc) * R + (b % R) : bin) * p.bps : (cls == 2 ? -2 : -1);
    }
    const int sym_bytes = ND * p.bps / 8;           // <= 64 R, a multiple of 4 (checked by the launcher)
    const bool aligned = (reinterpret_cast<uintptr_t>(p.bytes) & 3) == 0;
    // (The branch-free paydw_issue / paydw_settle form of k_tx4096 measured 13-20 % SLOWER in this kernel -- 0.59 -> 0.51 of the
    // roofline at N = 64 .. 512, 0.49 -> 0.39 at N = 1024 -- so it keeps the conditional loads: the symbol after next is requested a
    // whole step ahead and the wait lands where the step would wait for its LDS traffic anyway.)
    auto dword = [&](long long by) -> unsigned {    // stream bytes by .. by + 3, zero past the end
        if (aligned && by + 4 <= p.n_bytes) return *reinterpret_cast<const unsigned *>(p.bytes + by);
        unsigned v = 0;
        for (int j = 0; j < 4; ++j) if (by + j < p.n_bytes) v |= (unsigned)p.bytes[by + j] << (8 * j);
        return v;
    };
    auto fetch = [&](long long sg, unsigned &d0, unsigned &d1) {
        d0 = d1 = 0u;
        if (sg >= p.n_sym) return;
        const long long base = sg * sym_bytes;
        if (4 * l < sym_bytes) d0 = dword(base + 4 * l);
        if (4 * (l + LPS) < sym_bytes) d1 = dword(base + 4 * (l + LPS));
    };
    const long long stride = (long long)gridDim.x * G;
    unsigned d0, d1;
    fetch((long long)blockIdx.x * G + g, d0, d1);
    sbw[l] = d0;
    sbw[l + LPS] = d1;
    if (l < 2) sbw[2 * LPS + l] = 0u;
    fetch((long long)blockIdx.x * G + g + stride, d0, d1);
    __syncthreads();

    for (long long base = (long long)blockIdx.x * G; base < p.n_sym; base += stride) {
        const long long sg = base + g;
        long long left = p.n_bytes - sg * sym_bytes;              // stream bytes that belong to this symbol
        left = left < 0 ? 0 : (left < sym_bytes ? left : sym_bytes);
        const int live_bits = sg < p.n_sym ? (int)(((left * 8 + p.bps - 1) / p.bps) * p.bps) : 0;
        cf v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // two byte reads + two axis-level reads per point.  The dword + point-table mapping of k_txframe_mid / k_tx4096 (two LDS
            // instructions instead of four) measured 10 % SLOWER in this kernel (0.238 -> 0.265 ms per 2^27 samples at N = 64 / 512), and
            // the branch-free form 30 % slower: left as it was
            cf pt = make_float2(0.f, 0.f);
            if (boff[e] == -2) pt = make_float2(1.f, 0.f);
            else if (boff[e] >= 0 && boff[e] < live_bits) {
                const int bit = boff[e];
                const unsigned two = (unsigned)sb[bit >> 3] | ((unsigned)sb[(bit >> 3) + 1] << 8);
                const unsigned idx = (two >> (bit & 7)) & ((1u << p.bps) - 1u);
                pt = p.bps == 1 ? map_point(idx, 1) : make_float2(lvl[idx & ((1u << (p.bps >> 1)) - 1u)], lvl[idx >> (p.bps >> 1)]);
            }
            v[e] = pt;
        }
        stage_a<R, true>(v, tA, u);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int slot = R >= 8 ? M::slot_of_eu(e, u) : e % R; // = slot_of_row(e + 8 u), with e a compile-time constant
            const int col = R >= 8 ? colA : colA + LPS * (e / R);
            Tsym[slot * TS + col] = (R < 8 && e % R == 0) ? v[e] : cmul(v[e], z[e]);
        }
        symbol_sync<LPS>(); // T complete; every lane of the symbol is past the mapping stage
        sbw[l] = d0;        // next symbol's bytes
        sbw[l + LPS] = d1;
        fetch(sg + 2 * stride, d0, d1);
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = T[rs * TS + t + 8 * m];
        stage_b<true>(v, buf, t, wr, w);
        // v[q] = N x[cB + R (t + 8 q)]: through T once more in sample order ([n >> 6][n & 63]) so that every store is a
        // full 16 bytes per lane
        symbol_sync<LPS>(); // every lane has read its stage-B inputs out of T
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = cB + R * (t + 8 * q);
            Tsym[M::t2_index(n)] = make_float2(v[q].x * (1.0f / N), v[q].y * (1.0f / N));
        }
        symbol_sync<LPS>();
        if (sg < p.n_sym) { // prefix_block: out = [x[N - CP .. N), x[0 .. N)]
            float4 *dst4 = reinterpret_cast<float4 *>(p.out + sg * S);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = l + LPS * j, n = 2 * i;                        // sample pair (n, n + 1)
                const float4 y = *reinterpret_cast<const float4 *>(Tsym + M::t2_index(n));
                dst4[(CP >> 1) + i] = y;
                if (j == 3) dst4[i - ((N - CP) >> 1)] = y;                   // n >= N - CP: the cyclic prefix
            }
        }
        symbol_sync<LPS>(); // sbw / T are reused by the next step
    }
}

template <int R> hipError_t launch_tx_mid(const MidTxParams &p, bool guard, hipStream_t st, int num_cu, long long cap) {
    constexpr int G = Mid<R>::G;
    const long long steps = (p.n_sym + G - 1) / G;
    const long long grid = mid_grid(steps, num_cu, Mid<R>::OCC_TX, cap);
    if (guard) hipLaunchKernelGGL((k_tx_mid<R, true>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_tx_mid<R, false>), dim3((unsigned)grid), dim3(256), 0, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// k_txframe_mid: encode (src/transmitter.rs:11-58) for N = 64 R, R in {1 .. 32}, in ONE pass over HBM: the data symbols of a
// frame are built TWICE -- first only for the frame's signed maximum (normalize, transmitter.rs:184-188, needs it before
// the first sample can leave), then again to be stored divided by it -- instead of written, read back and rewritten
// (k_sym<N, M_TX> + k_tx_finish: three passes of 8 B per sample).  The inverse transform is the one of k_tx_mid.
// A workgroup takes `fpw` whole frames per round (their D data symbols fill its 32 / R symbol slots step by step); the ten
// constant header blocks are copied from the context's table, scaled like the rest.
struct MidTxFrameParams {
    const uint8_t *payload;
    long long payload_stride;
    const int32_t *payload_len;
    int payload_bytes;
    long long n_frames;
    int D, fpw;          // data symbols per frame; frames per workgroup round (<= 32)
    const float2 *tw;
    const float2 *header; // 10 S samples
    float header_max;
    float2 *out;
    long long out_stride; // samples
    int bps;
    int optimistic;      // 1: samples leave scaled by 1 / header_max while the frame's maximum forms; a round is rebuilt only when a frame exceeds it
};

// KEEP > 0: a round's symbols fit KEEP steps, so every lane KEEPS its KEEP x 8 points in registers while the round's maxima form and
// the symbols are built ONCE (N = 128 / 256 / 512 frames of up to 16 / 16 / 16 data symbols); KEEP == 0: the two-pass scheme above.
// (R <= 8, two-pass / optimistic form: 117-128 VGPRs and, with the transpose slabs folded into T, 23 KB of LDS -> FOUR waves per SIMD)
template <int R, bool GUARD, int KEEP>
__global__ __launch_bounds__(256, (R <= 8 && KEEP == 0) ? 4 : 3) void k_txframe_mid(MidTxFrameParams p) {
    typedef Mid<R> M;
    constexpr int N = M::N, S = M::S, CP = M::CP, LPS = M::LPS, G = M::G, Q = M::Q, TS = M::TS;
    constexpr int ND = GUARD ? 48 * R : N;
    constexpr int SB_DW = 16 * R + 2;
    __shared__ __align__(16) cf T[32 * TS];
    __shared__ unsigned sbw_all[G * SB_DW];
    __shared__ cf ptab[256];
    __shared__ unsigned fmax[32];                   // per frame of the round: max(0, re, im) of its data symbols, as float bits

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = tid / LPS, l = tid % LPS;
    const int u = R >= 8 ? l % Q : 0;
    const int colA = R >= 8 ? l / Q : l;
    const int t = tid & 7, rs = tid >> 3;
    const int cB = M::row_of_slot(rs % R);
    // the FFT64 transposes of stage B run IN PLACE in the eight-lane group's own row of T (the row it has just read its inputs from: nobody
    // else reads it, and the in-order LDS pipe keeps the reads ahead of the writes) -- as k_rxframe1024 does: no separate slabs, 18 KB less LDS
    cf *buf = T + rs * TS;
    const int wr = swz(8 * t);
    unsigned *sbw = sbw_all + g * SB_DW;
    cf *Tsym = T + g * R * TS;

    if (tid < (1 << p.bps)) ptab[tid] = map_point((unsigned)tid, p.bps);
    const unsigned fmask = (1u << p.bps) - 1u;
    if (l < 2) sbw[2 * LPS + l] = 0u;               // slack for the two-byte window
    __shared__ cf wtab[56];                         // conj W64^(r t), r = 1 .. 7: read at use (fourteen registers less: see k_demod_mid)
    if (tid < 56) { const cf x = p.tw[R * (tid % 7 + 1) * (tid / 7)]; wtab[tid] = make_float2(x.x, -x.y); }
    __syncthreads();
    const cf *w = wtab + 7 * t;
    cf tA[7];
#pragma unroll
    for (int j = 1; j < 8; ++j) { const cf x = Q > 1 ? p.tw[64 * u * j] : make_float2(1.f, 0.f); tA[j - 1] = make_float2(x.x, -x.y); }
    cf z[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const cf x = R >= 8 ? p.tw[colA * (e + 8 * u)] : p.tw[(colA + LPS * (e / R)) * (e % R)];
        z[e] = make_float2(x.x, -x.y);
    }
    int boff[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int a = R >= 8 ? u + Q * e : e % R;
        const int b = R >= 8 ? colA : colA + LPS * (e / R);
        const int bin = 64 * a + b, kc = bin / R, cls = carrier_class64(kc, GUARD);
        boff[e] = cls == 0 ? (GUARD ? data_classes_below64(kc) * R + (b % R) : bin) * p.bps : (cls == 2 ? -2 : -1);
    }
    const int sym_bytes = ND * p.bps / 8;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p.payload) | (uintptr_t)p.payload_stride) & 3) == 0;
    const int slots = p.fpw * p.D, steps = (slots + G - 1) / G;
    const long long rounds = (p.n_frames + p.fpw - 1) / p.fpw;

    // The stream bytes of symbol (f, k) -- two dwords per lane -- and the frame's length are REQUESTED one item ahead (issue_sym) and
    // taken out of the registers when the symbol is built (settle in build): read where they were used, the length and each dword cost a
    // dependent round trip per symbol, and every `s_waitcnt vmcnt(0)` behind such a load also waited for the PREVIOUS symbol's stores to be
    // acknowledged (round-5 ISA scan: eleven serialized memory round trips per N = 256 frame).  The loads are unconditional: a row is
    // readable for payload_bytes whatever its own length (include/ofdm_hip.h), the verdict is taken at the settle (paydw_settle).
    struct Pre { unsigned d0, d1; int len_raw; };
    auto issue_sym = [&](bool valid, long long f, int k, Pre &pr) {
        const long long fc = valid ? f : 0;
        const uint8_t *pay = p.payload + fc * p.payload_stride;
        const long long by0 = (long long)k * sym_bytes + 4 * l - 16, by1 = by0 + 4 * LPS;   // payload byte of this lane's two dwords
        pr.d0 = paydw_issue(pay, by0, p.payload_bytes, valid && 4 * l < sym_bytes, aligned, p.tw);
        pr.d1 = paydw_issue(pay, by1, p.payload_bytes, valid && 4 * (l + LPS) < sym_bytes, aligned, p.tw);
        pr.len_raw = p.payload_len ? p.payload_len[fc] : p.payload_bytes;
    };
    // builds symbol (f, k): v[q] = N x[cB + R (t + 8 q)] (everything before the 1/N scale of k_tx_mid)
    auto build = [&](bool valid, long long f, int k, cf *v, const Pre &pr) {
        long long len = 0;
        if (valid) {
            len = row_len(pr.len_raw, p.payload_bytes);
            const uint8_t *pay = p.payload + f * p.payload_stride;
            const long long sb0 = (long long)k * sym_bytes;
            // stream bytes of a frame: [16-byte little-endian length | payload | zeros] (src/packets/mod.rs:20-32)
            auto word = [&](unsigned raw, long long sb, bool want) -> unsigned {
                if (!want) return 0u;
                if (sb < 16) return sb < 8 ? (unsigned)((unsigned long long)len >> (8 * sb)) : 0u;
                return paydw_settle(raw, pay, sb - 16, len, true, aligned);
            };
            sbw[l] = word(pr.d0, sb0 + 4 * l, 4 * l < sym_bytes);
            sbw[l + LPS] = word(pr.d1, sb0 + 4 * (l + LPS), 4 * (l + LPS) < sym_bytes);
        }
        symbol_sync<LPS>();
        long long left = 16 + len - (long long)k * sym_bytes;   // stream bytes that belong to this symbol
        left = left < 0 ? 0 : (left < sym_bytes ? left : sym_bytes);
        const int live_bits = valid ? (int)(((unsigned)left * 8u + (unsigned)p.bps - 1u) / (unsigned)p.bps) * p.bps : 0; // left <= 2048
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = tx_point<false>(sbw, ptab, boff[e], live_bits, fmask);
        }
        stage_a<R, true>(v, tA, u);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int slot = R >= 8 ? M::slot_of_eu(e, u) : e % R; // = slot_of_row(e + 8 u), with e a compile-time constant
            const int col = R >= 8 ? colA : colA + LPS * (e / R);
            Tsym[slot * TS + col] = (R < 8 && e % R == 0) ? v[e] : cmul(v[e], z[e]);
        }
        symbol_sync<LPS>();
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = T[rs * TS + t + 8 * m];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the row reads above precede stage B's in-place writes
        stage_b<true>(v, buf, t, wr, w);
        symbol_sync<LPS>(); // T and the byte window are free again
    };

    // the samples of symbol (f0 + fl, k), divided by the frame's maximum, through T into sample order and out (prefix_block: out = [x[N - CP .. N), x[0 .. N)]).
    // RAW (the build-once scheme): the symbol's N samples go out UNNORMALISED (x 1 / N only) and without the prefix; rescale() finishes them.
    auto emit = [&](bool valid, long long f0, int fl, int k, const cf *v, auto raw_tag, bool header_only = false) {
        constexpr bool RAW = decltype(raw_tag)::value;
        const float mx = RAW ? 1.f : header_only ? p.header_max : fmaxf(p.header_max, __uint_as_float(fmax[valid ? fl : 0]));
        // ONE division per lane and symbol, then multiplies (<= 1 ulp from the two roundings x / N, / max of the staged path -- inside the
        // 1e-5 of the parity rule): sixteen IEEE divisions per lane and symbol were half of this kernel's arithmetic (round 5)
        const float sc = RAW ? 1.0f / N : (1.0f / N) / mx;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = cB + R * (t + 8 * q);
            Tsym[M::t2_index(n)] = make_float2(v[q].x * sc, v[q].y * sc);
        }
        symbol_sync<LPS>();
        if (valid) {
            float4 *dst4 = reinterpret_cast<float4 *>(p.out + (f0 + fl) * p.out_stride + (long long)(10 + k) * S);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = l + LPS * j, n = 2 * i;
                const float4 y = *reinterpret_cast<const float4 *>(Tsym + M::t2_index(n));
                dst4[(CP >> 1) + i] = y;
                if (!RAW && j == 3) dst4[i - ((N - CP) >> 1)] = y;
            }
        }
        symbol_sync<LPS>();
    };
    // Build-once scheme, second half: every lane takes back exactly the four sample pairs IT stored for symbol (f0 + fl, k) -- they
    // are minutes old in L2 / the memory-side cache --, divides them by the frame's maximum (the same two roundings as the
    // two-pass kernel: x / N, then / max) and stores them again, now with the cyclic prefix.
    auto rescale = [&](bool valid, long long f0, int fl, int k) {
        if (!valid) return;
        const float inv = 1.0f / fmaxf(p.header_max, __uint_as_float(fmax[fl]));
        float4 *dst4 = reinterpret_cast<float4 *>(p.out + (f0 + fl) * p.out_stride + (long long)(10 + k) * S);
        float4 y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = dst4[(CP >> 1) + l + LPS * j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = l + LPS * j;
            const float4 o = make_float4(y[j].x * inv, y[j].y * inv, y[j].z * inv, y[j].w * inv);
            dst4[(CP >> 1) + i] = o;
            if (j == 3) dst4[i - ((N - CP) >> 1)] = o;
        }
    };
    auto note_max = [&](bool valid, int fl, const cf *v) {
        float mine = 0.f;   // max first, ONE scaling behind it (x -> x / N is monotone: the same bits as scaling every sample), three-operand maxima
#pragma unroll
        for (int q = 0; q < 8; ++q) mine = __builtin_fmaxf(mine, __builtin_fmaxf(v[q].x, v[q].y));
        mine *= 1.0f / N;
        constexpr int WL = LPS < 64 ? LPS : 64;
#pragma unroll
        for (int sh = WL / 2; sh >= 1; sh >>= 1) mine = fmaxf(mine, __shfl_xor(mine, sh, 64));
        if (valid && (l & (WL - 1)) == 0) atomicMax(&fmax[fl], __float_as_uint(mine));
    };
    // ---- header blocks (src/transmitter.rs:22-34), divided by the frame maximum like the data
    auto emit_headers = [&](long long f0) {
        for (int fl = 0; fl < p.fpw && f0 + fl < p.n_frames; ++fl) {
            const float inv = 1.0f / fmaxf(p.header_max, __uint_as_float(fmax[fl]));   // one division, then multiplies (<= 1 ulp)
            float4 *dst4 = reinterpret_cast<float4 *>(p.out + (f0 + fl) * p.out_stride);
            const float4 *h4 = reinterpret_cast<const float4 *>(p.header);
            // four table reads in flight per lane, then their four stores (one load, its wait and its store at a time cost 5 S / 256
            // dependent round trips per frame); the reads are unconditional from a clamped index, only the stores are predicated
            for (int i0 = tid; i0 < 5 * S; i0 += 4 * 256) {
                float4 h[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int i = i0 + 256 * j; h[j] = h4[i < 5 * S ? i : 5 * S - 1]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i0 + 256 * j;
                    if (i < 5 * S) dst4[i] = make_float4(h[j].x * inv, h[j].y * inv, h[j].z * inv, h[j].w * inv);
                }
            }
        }
    };

    // item = (round, step): its symbol slot sigma = step * G + g of the round's fpw x D slots
    auto item_sym = [&](long long round, int step, long long &f, int &fl, int &k, bool &valid) {
        const int sigma = step * G + g;
        fl = sigma / p.D; k = sigma - fl * p.D;
        f = round * p.fpw + fl;
        valid = round < rounds && sigma < slots && f < p.n_frames;
    };
    Pre pre;
    {   // the first item's bytes
        long long f; int fl, k; bool valid;
        item_sym(blockIdx.x, 0, f, fl, k, valid);
        issue_sym(valid, f, k, pre);
    }
    for (long long round = blockIdx.x; round < rounds; round += gridDim.x) {
        if (tid < 32) fmax[tid] = 0u;
        lds_barrier();   // (LDS only: a __syncthreads() here also waited for the previous round's stores to be acknowledged)
        const long long f0 = round * p.fpw;
        if (KEEP > 0) { // steps <= KEEP (launcher): build once, keep, scale, store
            cf vv[KEEP > 0 ? KEEP : 1][8];
#pragma unroll
            for (int step = 0; step < KEEP; ++step) {
                if (step < steps) {
                    long long f; int fl, k; bool valid;
                    item_sym(round, step, f, fl, k, valid);
                    const Pre cur = pre;
                    {   // the next item: the next step of this round, or the first of the workgroup's next round
                        long long fn; int fln, kn; bool vn;
                        if (step + 1 < steps) item_sym(round, step + 1, fn, fln, kn, vn); else item_sym(round + gridDim.x, 0, fn, fln, kn, vn);
                        issue_sym(vn, fn, kn, pre);
                    }
                    build(valid, f, k, vv[step], cur);
                    note_max(valid, fl, vv[step]);
                }
            }
            lds_barrier();
            emit_headers(f0);
#pragma unroll
            for (int step = 0; step < KEEP; ++step) {
                if (step < steps) {
                    long long f; int fl, k; bool valid;
                    item_sym(round, step, f, fl, k, valid);
                    emit(valid, f0, fl, k, vv[step], std::false_type{});
                }
            }
            lds_barrier(); // fmax is reset by the next round
            continue;
        }
        if (KEEP < 0) { // build once: unnormalised samples out, the frame's maximum, then the rescale sweep over what was just written
            for (int step = 0; step < steps; ++step) {
                long long f; int fl, k; bool valid;
                item_sym(round, step, f, fl, k, valid);
                const Pre cur = pre;
                {
                    long long fn; int fln, kn; bool vn;
                    if (step + 1 < steps) item_sym(round, step + 1, fn, fln, kn, vn); else item_sym(round + gridDim.x, 0, fn, fln, kn, vn);
                    issue_sym(vn, fn, kn, pre);
                }
                cf v[8];
                build(valid, f, k, v, cur);
                note_max(valid, fl, v);
                emit(valid, f0, fl, k, v, std::true_type{});
            }
            __syncthreads(); // every maximum of the round is final (and this thread's stores are acknowledged: s_waitcnt vmcnt(0))
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            emit_headers(f0);
            for (int step = 0; step < steps; ++step) {
                long long f; int fl, k; bool valid;
                item_sym(round, step, f, fl, k, valid);
                rescale(valid, f0, fl, k);
            }
            __syncthreads(); // fmax is reset by the next round
            continue;
        }
        // The frame maximum (normalize, transmitter.rs:184-188) is max(header maximum, data maximum), and the constant header blocks
        // carry the full-scale locking signal: with the reference's constellations a data symbol exceeds it only for payloads crafted
        // to line the carriers up (re x[n] <= sqrt 2 against 1.0).  OPTIMISTIC scheme (round 5): pass 0 builds every symbol of the round
        // once, notes its maximum and stores it divided by the HEADER maximum; only if a frame of the round turned out larger is the
        // round built again (pass 1) and stored divided by the true maxima -- same thread, same addresses, program order.
        // Without p.optimistic: pass 0 only forms the maxima, pass 1 stores (every symbol built twice; the A/B).
        // ONE instance of the symbol builder, the pass is a uniform branch around it (two inlined instances spill at 4 waves per SIMD).
        const bool opt = p.optimistic != 0;
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {
                lds_barrier();       // every maximum of the round is final
                emit_headers(f0);
                bool redo = false;
                for (int fl = 0; fl < p.fpw; ++fl) redo |= __uint_as_float(fmax[fl]) > p.header_max;
                if (opt && !redo) break;
            }
            for (int step = 0; step < steps; ++step) {
                long long f; int fl, k; bool valid;
                item_sym(round, step, f, fl, k, valid);
                Pre cur;
                if (opt && pass == 1) issue_sym(valid, f, k, cur);   // the rare rebuild asks for its bytes where it uses them; `pre` keeps the next round's
                else {
                    cur = pre;
                    // the next item: next step, the first step of pass 1 (two-pass scheme), or the first step of the workgroup's next round
                    long long fn; int fln, kn; bool vn;
                    if (step + 1 < steps) item_sym(round, step + 1, fn, fln, kn, vn);
                    else if (pass == 0 && !opt) item_sym(round, 0, fn, fln, kn, vn);
                    else item_sym(round + gridDim.x, 0, fn, fln, kn, vn);
                    issue_sym(vn, fn, kn, pre);
                }
                cf v[8];
                build(valid, f, k, v, cur);
                if (pass == 0) {
                    note_max(valid, fl, v);
                    if (!opt) continue;
                }
                emit(valid, f0, fl, k, v, std::false_type{}, pass == 0);
            }
        }
        lds_barrier(); // fmax is reset by the next round
    }
}

template <int R, int KEEP> static void launch_txframe_mid_k(const MidTxFrameParams &p, bool guard, dim3 grid, hipStream_t st) {
    if (guard) hipLaunchKernelGGL((k_txframe_mid<R, true, KEEP>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_txframe_mid<R, false, KEEP>), grid, dim3(256), 0, st, p);
}
template <int R> hipError_t launch_txframe_mid(MidTxFrameParams p, bool guard, hipStream_t st, int num_cu, long long cap, int keep_max, bool rewrite) {
    constexpr int G = Mid<R>::G;
    // a frame whose symbols fit keep_max (<= 4) steps of the workgroup is built ONCE, its points kept in registers until the frame's
    // maximum is known; otherwise frames per round: the count (<= 8, <= 32 slots' worth) that wastes the fewest symbol slots of
    // the last step, and every symbol is built twice
    const int steps1 = (p.D + G - 1) / G;
    // (R <= 4: the optimistic two-pass form runs at four waves per SIMD and is as fast or faster -- tools/lab/enc_keep_ab.py: N = 128, D = 8:
    //  0.65 against 0.575 of the one-write roofline; N = 256, D = 8: 0.64 against 0.60; from R = 8 on the kept form is 0-3 % ahead)
    const int keep = (steps1 <= keep_max && steps1 <= 1 && R >= 8) ? 1 : 0;   // KEEP = 2 / 4 instantiate the symbol builder 2 / 4 times: 18-170 spilled registers, not built
    if (keep) p.fpw = 1;
    else {
        int best = 1; double waste = 2.0;
        for (int f = 1; f <= 8; ++f) {
            const long long slots = (long long)f * p.D, padded = (slots + G - 1) / G * G;
            const double wst = (double)(padded - slots) / (double)padded;
            if (wst < waste - 1e-9) { waste = wst; best = f; }
        }
        p.fpw = best;
    }
    const long long rounds = (p.n_frames + p.fpw - 1) / p.fpw;
    const dim3 grid((unsigned)mid_grid(rounds, num_cu, (R <= 8 && !keep && !rewrite) ? 4 : 3, cap)); // waves per SIMD the instantiation is built for
    if (keep) launch_txframe_mid_k<R, 1>(p, guard, grid, st);
    else if (rewrite) launch_txframe_mid_k<R, -1>(p, guard, grid, st);
    else launch_txframe_mid_k<R, 0>(p, guard, grid, st);
    return hipGetLastError();
}

} // namespace

// RX demod for N in {128 .. 2048}: regular symbol streams, and the data symbols of frames after timing (per-frame offset, CFO,
// live-symbol count, zero-fill past the capture).  hipErrorNotSupported => caller uses k_sym<N, M_DEMOD>.
hipError_t run_demod_mid(int n_fft, const SymParams &sp, hipStream_t st, int num_cu) {
    if (n_fft < 128 || n_fft > 2048) return hipErrorNotSupported;
    if (sp.soft || sp.syms_per_frame <= 0) return hipErrorNotSupported;
    const int S = n_fft + n_fft / 4, R = n_fft / 64;
    const bool frame = sp.offset || sp.f_delta || sp.nsym_frame ||
                       (long long)(sp.first_symbol + sp.syms_per_frame) * S > sp.frame_len; // tail padding needs the bounds checks
    if (sp.in_sym_stride != S || sp.in_skip != n_fft / 4) return hipErrorNotSupported;
    if ((reinterpret_cast<uintptr_t>(sp.out_bytes) & 3) || (sp.out_stride & 3)) return hipErrorNotSupported;
    if (sp.hk && sp.hk_stride != 0 && sp.hk_stride != n_fft) return hipErrorNotSupported;
    MidRxParams p;
    p.in = sp.in; p.frame_stride = sp.frame_stride; p.total = sp.n_frames * (long long)sp.syms_per_frame;
    p.syms_per_frame = sp.syms_per_frame; p.first_symbol = sp.first_symbol; p.step_f = 0; p.step_k = 0;
    p.tw = sp.tw; p.hk = sp.hk; p.hk_stride = sp.hk_stride; p.out = sp.out_bytes; p.out_stride = sp.out_stride;
    p.offset = sp.offset; p.f_delta = sp.f_delta; p.nsym_frame = sp.nsym_frame; p.frame_len = sp.frame_len;
    if (p.total <= 0) return hipSuccess;
    const long long cap = tuning_or_default(sp.tune).grid_cap;
    trace_add(sp.trace, frame ? "k_demod_mid<frame>" : "k_demod_mid");
    switch (R) {
    case 2: return dispatch_demod_mid<2>(p, sp.bps, sp.guard != 0, frame, st, num_cu, cap);
    case 4: return dispatch_demod_mid<4>(p, sp.bps, sp.guard != 0, frame, st, num_cu, cap);
    case 8: return dispatch_demod_mid<8>(p, sp.bps, sp.guard != 0, frame, st, num_cu, cap);
    case 16: return dispatch_demod_mid<16>(p, sp.bps, sp.guard != 0, frame, st, num_cu, cap);
    case 32: return dispatch_demod_mid<32>(p, sp.bps, sp.guard != 0, frame, st, num_cu, cap);
    }
    return hipErrorNotSupported;
}

// Continuous-stream TX for N in {64 .. 2048} (N = 64: R = 1, stage A is the identity and 32 symbols share a workgroup step).
// hipErrorNotSupported => caller uses k_sym<N, M_TX>.
hipError_t run_tx_mid(int n_fft, const SymParams &sp, hipStream_t st, int num_cu) {
    if (n_fft < 64 || n_fft > 2048) return hipErrorNotSupported;
    if (sp.tx_raw_total < 0 || sp.syms_per_frame != 1 || sp.payload_len) return hipErrorNotSupported;
    const int R = n_fft / 64, S = n_fft + n_fft / 4;
    const int nd = sp.guard ? 48 * R : n_fft;
    const int sym_bytes = nd * sp.bps / 8;
    if ((sym_bytes & 3) || sp.payload_stride != sym_bytes || sp.out_stride_s != S) return hipErrorNotSupported;
    if (reinterpret_cast<uintptr_t>(sp.out) & 15) return hipErrorNotSupported;
    if (sp.n_frames <= 0) return hipSuccess;
    MidTxParams p;
    p.bytes = sp.payload; p.n_bytes = sp.tx_raw_total; p.n_sym = sp.n_frames; p.tw = sp.tw; p.out = sp.out; p.bps = sp.bps;
    const long long cap = tuning_or_default(sp.tune).grid_cap;
    trace_add(sp.trace, "k_tx_mid");
    switch (R) {
    case 1: return launch_tx_mid<1>(p, sp.guard != 0, st, num_cu, cap);
    case 2: return launch_tx_mid<2>(p, sp.guard != 0, st, num_cu, cap);
    case 4: return launch_tx_mid<4>(p, sp.guard != 0, st, num_cu, cap);
    case 8: return launch_tx_mid<8>(p, sp.guard != 0, st, num_cu, cap);
    case 16: return launch_tx_mid<16>(p, sp.guard != 0, st, num_cu, cap);
    case 32: return launch_tx_mid<32>(p, sp.guard != 0, st, num_cu, cap);
    }
    return hipErrorNotSupported;
}

// encode for N in {64 .. 2048}: one pass over HBM (N = 64: for the frames k_txframe64 does not take, more than 56 data
// symbols).  hipErrorNotSupported => caller runs k_sym<N, M_TX> + k_tx_finish.
hipError_t run_txframe_mid(int n_fft, const SymParams &sp, const float2 *header, float header_max, hipStream_t st, int num_cu) {
    if (n_fft < 64 || n_fft > 2048) return hipErrorNotSupported;
    if (sp.tx_raw_total >= 0 || sp.syms_per_frame <= 0) return hipErrorNotSupported;
    const int R = n_fft / 64;
    if ((reinterpret_cast<uintptr_t>(sp.out) & 15) || (sp.out_stride_s & 1)) return hipErrorNotSupported;
    if (sp.n_frames <= 0) return hipSuccess;
    MidTxFrameParams p;
    p.payload = sp.payload; p.payload_stride = sp.payload_stride; p.payload_len = sp.payload_len; p.payload_bytes = sp.payload_bytes;
    p.n_frames = sp.n_frames; p.D = sp.syms_per_frame; p.fpw = 1; p.tw = sp.tw; p.header = header; p.header_max = header_max;
    p.out = sp.out; p.out_stride = sp.out_stride_s; p.bps = sp.bps;
    const Tuning &tu = tuning_or_default(sp.tune);
    p.optimistic = tu.no_txframe_optimistic ? 0 : 1;
    const long long cap = tu.grid_cap;
    const int keep_max = tu.txframe_keep_steps;   // 0 = always build twice (A/B)
    const int G = 32 / R, steps1 = (p.D + G - 1) / G;
    const bool once = steps1 <= keep_max && steps1 <= 1 && R >= 8;   // (launch_txframe_mid's condition)
    const bool rewrite = tu.txframe_rewrite != 0 && !once;
    trace_add(sp.trace, once ? "k_txframe_mid<once>" : rewrite ? "k_txframe_mid<rewrite>" : "k_txframe_mid");
    switch (R) {
    case 1: return launch_txframe_mid<1>(p, sp.guard != 0, st, num_cu, cap, keep_max, rewrite);
    case 2: return launch_txframe_mid<2>(p, sp.guard != 0, st, num_cu, cap, keep_max, rewrite);
    case 4: return launch_txframe_mid<4>(p, sp.guard != 0, st, num_cu, cap, keep_max, rewrite);
    case 8: return launch_txframe_mid<8>(p, sp.guard != 0, st, num_cu, cap, keep_max, rewrite);
    case 16: return launch_txframe_mid<16>(p, sp.guard != 0, st, num_cu, cap, keep_max, rewrite);
    case 32: return launch_txframe_mid<32>(p, sp.guard != 0, st, num_cu, cap, keep_max, rewrite);
    }
    return hipErrorNotSupported;
}

} // namespace ofdm
