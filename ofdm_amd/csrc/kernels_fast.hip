// kernels_fast.hip -- the shape-specialised kernels of the hot path (everything else is the generic k_sym):
//   k_demod64      N = 64 RX demod for regular streams (BASELINE config 2, the headline)      -- described below
//   k_rxframe64    N = 64 per-frame receive body after timing (config 3): channel estimate + demod [+ finish]
//   k_txframe64    N = 64 encode: frame built in LDS, one HBM pass
//   (k_rxframe1024, the N = 1024 per-frame receive body of config 4, lives in kernels_rx1024.hip)
//   k_demod4096 / k_tx4096   N = 4096 RX demod / continuous TX (config 5), FFT as 64 x 64
//
// k_demod64: CP strip + FFT64 + [equalise] + pilot phase + hard demap + LSB-first bit packing, for regularly spaced,
// HBM-resident symbols.
//
// Wave-centric: one 64-lane wavefront owns 8 consecutive OFDM symbols (8 lanes x 8 points each) per iteration and
// never meets a workgroup barrier.  Per iteration and lane:
//   8 x global_load_dwordx2 with immediate offsets (the 128-byte cyclic prefix of each 640-byte symbol is a whole,
//     aligned cache line and is never fetched), scalar base address, no integer division anywhere;
//   radix-8 butterfly -> XOR-swizzled LDS slab -> radix-8 butterfly (7 loop-invariant twiddles in registers);
//   pilot phase: atan2 on the 4 pilot lanes, 3 DPP adds inside the 8-lane group, one sincos;
//   hard decisions; bit fields OR-ed into the wave's packed output image in LDS (ds_or_b32), then the image is
//   stored with unit-stride dword stores (36 B per symbol for 64-QAM with guard bands).
// Roofline: HBM -- 640 algorithmic bytes read per symbol (512 actually fetched) + packed bytes written.
#include "device_common.hpp"
#include "kernels.hpp"
#include <type_traits>
#include <mutex>
#include <stdlib.h>

extern "C" __device__ float __ocml_atan2pi_f32(float, float); // atan2(y, x) / pi (ROCm device library)

namespace ofdm {

// DPP helpers: sum over the 8 lanes of a symbol group (lanes 8s .. 8s+7)
template <int CTRL> __device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum8(float x) {
    x += dpp_f<0xB1>(x);  // quad_perm [1,0,3,2]  : lane ^ 1
    x += dpp_f<0x4E>(x);  // quad_perm [2,3,0,1]  : lane ^ 2
    x += dpp_f<0x141>(x); // row_half_mirror      : lane -> 7 - lane inside each 8-lane half row
    return x;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x2 *lds_cf_ptr; // LDS pointer built from a 32-bit offset held in a VGPR
typedef __attribute__((address_space(3))) unsigned *lds_u32_ptr;
__device__ __forceinline__ unsigned lds_offset(const void *p) {
    return (unsigned)(unsigned long)((const __attribute__((address_space(3))) char *)p);
}
// A per-lane constant the compiler must KEEP in a register: without the empty asm it rematerialises the XOR / shift / mask
// that produced it inside the loop (to stay under 96 VGPRs), once per use and iteration.
__device__ __forceinline__ unsigned pinned(unsigned v) { asm volatile("" : "+v"(v)); return v; }

// BURST (4, 8 or 16): a wavefront takes BURST consecutive 8-symbol groups per step and stores their packed images together -- for the
// 288-byte images of 64-QAM with guard bands, 4 groups = 1152 bytes = nine WHOLE 128-byte lines (one group's image starts at
// a multiple of 288 bytes: 2.25 lines, shared with the neighbours).  Needs a contiguous output (rows back to back).
template <int BPS, bool GUARD, bool HK, int BURST = 1>
__global__ __launch_bounds__(256, 4) void k_demod64(Fast64Params p) {
    constexpr int S = 80, CP = 16;
    constexpr int ND = GUARD ? 48 : 64;          // data carriers per symbol
    constexpr int REGION_DW = ND * BPS / 4;      // packed output of 8 symbols, in dwords
    constexpr int SLAB = 8 * 72;                 // 8 symbols x (64 + 8 pad) points

    __shared__ cf slab_all[4 * SLAB];
    __shared__ __align__(16) unsigned img_all[4 * BURST * REGION_DW];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = lane >> 3, t = lane & 7;
    cf *buf = slab_all + wave * SLAB + s * 72;
    unsigned *img0 = img_all + wave * BURST * REGION_DW;

    // loop-invariant per-lane constants
    cf w[7];
#pragma unroll
    for (int r = 1; r < 8; ++r) w[r - 1] = p.tw[r * t];
    cf g[8];
    if (HK) {
#pragma unroll
        for (int m = 0; m < 8; ++m) { // 1 / H  (equalise: Y /= H, src/receiver.rs:68-70)
            cf h = p.hk[t + 8 * m];
            float ns = h.x * h.x + h.y * h.y;
            g[m] = make_float2(h.x / ns, -h.y / ns);
        }
    }
    // LDS byte addresses of the transpose (write: swz(8t + r) = 8t + (r ^ t); read: swz(t + 8m) = 8m + (t ^ m)) and of every
    // field's dword in the packed image, with the field's shift: 8 + 8 + 8 + 8 registers instead of ~5 integer instructions
    // per access and iteration
    unsigned wa[8], ra[8], fa[8], fs[8]; // fs: bit shift inside the dword, 0xFFFFFFFF = not a data bin
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        wa[r] = pinned(lds_offset(buf + (swz(8 * t) ^ r)));
        ra[r] = pinned(lds_offset(buf + 8 * r + (t ^ r)));
        const int c = t + 8 * r;
        const int q = GUARD ? data_classes_below64(c) : c;
        const int bo = (s * ND + q) * BPS;
        const bool data = carrier_class64(c, GUARD) == 0;
        fa[r] = pinned(lds_offset(img0 + (bo >> 5)));
        fs[r] = pinned(data ? (unsigned)(bo & 31) : 0xFFFFFFFFu);
    }
    const int lane_off = s * S + t; // sample offset of this lane inside the 8-symbol group

    // one 8-symbol group: CP strip + FFT64 + [equalise] + pilot phase + demap + packing into the LDS image at byte offset img_off
    auto group = [&](long long f, int kk, unsigned img_off) {
        const cf *src = p.in + f * p.frame_stride + (long long)(p.first_symbol + kk * 8) * S + CP + lane_off;
        cf v[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = src[8 * m];
        bfly8<false>(v);
        if (kProfile && p.debug == 3) { // profiling aid: loads + one butterfly
            float acc = 0.f;
#pragma unroll
            for (int m = 0; m < 8; ++m) acc += v[m].x + v[m].y;
            if (acc == 12345.678f) p.out[0] = 1;
            return;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) *(lds_cf_ptr)(unsigned long)wa[r] = f32x2{v[r].x, v[r].y};
#pragma unroll
        for (int m = 0; m < 8; ++m) { const f32x2 q = *(lds_cf_ptr)(unsigned long)ra[m]; v[m] = make_float2(q.x, q.y); }
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], w[r - 1]);
        bfly8<false>(v);
        // v[m] = X[t + 8m]
        if (HK) {
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = cmul(v[m], g[m]);
        }
        cf rot = make_float2(1.f, 0.f);
        if (GUARD) {
            // decode_block (src/receiver.rs:106-145): mean of the 4 pilot angles, rotate data points by -phase
            // pilots sit at bins 6, 25, 39, 58 = lanes t = 6 (m 0), 1 (m 3), 7 (m 4), 2 (m 7); the other lanes
            // feed (1, 0) -> angle 0, so ONE atan2 evaluation serves the whole wave
            cf pv = make_float2(1.f, 0.f);
            pv = (t == 6) ? v[0] : pv;
            pv = (t == 1) ? v[3] : pv;
            pv = (t == 7) ? v[4] : pv;
            pv = (t == 2) ? v[7] : pv;
            // mean angle in TURNS (sum of the four atan2pi values / 8), then the hardware sine / cosine, which take turns:
            // max abs error 1.3e-7 over [-pi, pi] on gfx950 (tools/lab/trig_probe.cpp; sincospif: 5e-8) for 2 instructions instead of ~35
            const float turns = sum8(__ocml_atan2pi_f32(pv.y, pv.x)) * 0.125f;
            rot = make_float2(__builtin_amdgcn_cosf(turns), -__builtin_amdgcn_sinf(turns)); // applied inside the demapper (demap_point_rot)
        }
        if (kProfile && p.debug == 2) { // profiling aid: everything but the packing and the stores
            float acc = 0.f;
#pragma unroll
            for (int m = 0; m < 8; ++m) acc += v[m].x + v[m].y;
            if (acc == 12345.678f) p.out[0] = 1;
            return;
        }
        // clear the packed image, OR every field in
        unsigned *img = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(img0) + img_off);
        for (int i = lane; i < REGION_DW; i += 64) img[i] = 0u;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            // bins t + 8 m with m in {1, 2, 5, 6} are data carriers for every lane (the nulls and pilots sit in rows 0, 3, 4, 7):
            // no per-lane test, no exec-mask juggling for half of the fields
            const bool all_data = !GUARD || m == 1 || m == 2 || m == 5 || m == 6;
            if (all_data || fs[m] != 0xFFFFFFFFu) {
                const unsigned idx = GUARD ? demap_point_rot(v[m], rot, BPS) : demap_point(v[m], BPS);
                const lds_u32_ptr wd = (lds_u32_ptr)(unsigned long)(fa[m] + img_off);
                __hip_atomic_fetch_or(wd, idx << fs[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (BPS > 1 && (32 % BPS) != 0) { // a field may straddle two dwords (only for 6-bit fields)
                    if (fs[m] + BPS > 32) __hip_atomic_fetch_or(wd + 1, idx >> (32 - fs[m]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    };
    auto store_image = [&](unsigned *dst, int ndw) { // ndw dwords of this wave's LDS image -> global
        if (kProfile && (p.debug == 1 || p.debug == 2 || p.debug == 3)) { if (img0[lane] == 0x12345678u) dst[0] = 1u; return; } // profiling aid: no stores
        if (kProfile && p.debug == 4) dst = reinterpret_cast<unsigned *>(p.out) + (blockIdx.x & 255) * 4 * BURST * REGION_DW + wave * BURST * REGION_DW; // L2-resident window
        if (kProfile && p.debug == 5) { for (int i = lane; i < ndw; i += 64) __builtin_nontemporal_store(img0[i], dst + i); return; }
        if ((REGION_DW % 4) == 0 && p.wide_stores) { // 16 bytes per lane
            if (p.store_policy) {   // lab key demod64_store_policy: 1 = nt, 2 = sc1, 3 = sc0 sc1 on the image stores (the output-buffer populations)
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                for (int i = lane; i < ndw / 4; i += 64) {
                    const v4u val = reinterpret_cast<const v4u *>(img0)[i];
                    uint4 *a = reinterpret_cast<uint4 *>(dst) + i;
                    if (p.store_policy == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(a), "v"(val) : "memory");
                    else if (p.store_policy == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(a), "v"(val) : "memory");
                    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(a), "v"(val) : "memory");
                }
                return;
            }
            for (int i = lane; i < ndw / 4; i += 64) reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(img0)[i];
        } else for (int i = lane; i < ndw; i += 64) dst[i] = img0[i];
    };

    if (BURST > 1) {
        // wave-uniform: burst j of this wave = groups [BURST j, BURST (j + 1)); one 64-bit division per BURST groups
        const long long n_bursts = p.n_groups / BURST; // the launcher guarantees divisibility and a contiguous output
        for (long long j = (long long)blockIdx.x * 4 + wave; j < n_bursts; j += p.stride_groups) {
            const long long g0 = j * BURST;
            long long f = g0 / p.groups_per_frame;
            int kk = (int)(g0 - f * p.groups_per_frame);
#pragma nounroll
            for (int b = 0; b < BURST; ++b) {
                group(f, kk, (unsigned)(b * REGION_DW * 4));
                if (++kk == p.groups_per_frame) { kk = 0; ++f; }
            }
            store_image(reinterpret_cast<unsigned *>(p.out + g0 * (long long)(REGION_DW * 4)), BURST * REGION_DW);
        }
        return;
    }

    // wave-uniform iteration state (no division in the loop: the host supplies the per-step increments)
    long long f = p.f0 + (long long)blockIdx.x * p.blk_df + wave * p.wave_df;
    int kk = p.k0 + (int)((blockIdx.x * (long long)p.blk_dk + wave * p.wave_dk));
    // normalise kk into [0, groups_per_frame)
    f += kk / p.groups_per_frame;
    kk %= p.groups_per_frame;
    for (long long g_idx = (long long)blockIdx.x * 4 + wave; g_idx < p.n_groups; g_idx += p.stride_groups) {
        group(f, kk, 0u);
        store_image(reinterpret_cast<unsigned *>(p.out + f * p.out_stride + (long long)kk * 8 * (ND * BPS / 8)), REGION_DW);
        // advance (wave-uniform)
        f += p.step_df;
        kk += p.step_dk;
        if (kk >= p.groups_per_frame) { kk -= p.groups_per_frame; f += 1; }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_rxframe64: the per-frame RX body for N = 64 after timing (BASELINE config 3): for every frame, one wavefront does
//   estimate_channel (src/receiver.rs:212-229) on the 5 training blocks  -> 1/H kept in registers
//   then per group of 8 data symbols: CFO derotation (receiver.rs:44-50, phase reduced in f64), CP strip + FFT64
//   (receiver.rs:99-104), equalise (receiver.rs:68-70), pilot phase (receiver.rs:106-145), hard demap + LSB-first packing
//   (receiver.rs:147-190).  Frames may start at any sample offset (8-byte aligned loads), samples at or beyond
//   frame_len read as zero (pad_chunk).  No workgroup barrier: the 4 waves of a workgroup own 4 different frames.
//   (Round 2 measured a variant with the next group's / next frame's samples prefetched into registers: 242 VGPRs, two
//   waves per SIMD, 1.12 ms per 262 144 config-3 frames against 0.94-1.14 ms for this one: no gain, not kept.)
struct RxFrame64Params {
    const float2 *in;
    long long n_frames, frame_stride, frame_len;
    const int32_t *offset;
    const double *f_delta;
    const int32_t *nsym;       // live data symbols per frame (0 = skip)
    const float2 *tw, *inv_training;
    unsigned char *out;        // raw decoded bytes, nsym * bytes_per_symbol per frame
    long long out_stride;
    float2 *hk;                // optional: channel estimate per frame (64 bins)
    // optional fused "finish" (length header parse + truncate, src/receiver.rs:85-95; no outer code): when final_out is set the
    // decoded bytes go straight to their final place (raw byte 16 + i -> final byte i, i < length) and `out` is unused
    unsigned char *final_out;
    long long final_stride;
    int32_t *final_len;
    const int32_t *frame_list;   // optional: only these frames (count on the device)
    const int32_t *frame_count;
    int32_t *cut_list;           // MODE 0: frames whose capture ends inside the frame are appended here (count at cut_count) for the MODE 1 launch
    int32_t *cut_count;
};

__device__ __forceinline__ cf lane_xor_sum(cf v) { // sum over the 8 symbol slots: lanes with equal (lane & 7)
    v.x += dpp_f<0x128>(v.x); v.y += dpp_f<0x128>(v.y);          // row_ror:8  : lane ^ 8
    v.x += __shfl_xor(v.x, 16, 64); v.y += __shfl_xor(v.y, 16, 64);
    v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64);
    return v;
}

// MODE (round 4): the body exists twice -- for frames that lie wholly inside their capture, and for captures that END inside the
// frame (zero-filled tail, pad_chunk) -- and with both in one kernel the register budget is the larger one's: 168 VGPRs, three
// waves per SIMD.  MODE 0 holds only the common body (126 VGPRs with guard bands: FOUR waves per SIMD -- the kernel is VALU-issue
// bound and three waves cannot fill the pipe) and appends the frames it has to leave to a device-side list; MODE 1
// holds only the cut body and runs over that list (normally empty: microseconds); MODE 2 is the round-3 kernel with both bodies,
// for callers that bring their own frame list (the one-pass kernel's slow list).
template <int BPS, bool GUARD, int MODE>
__global__ __launch_bounds__(256, (MODE == 0 && GUARD) ? 4 : 3) void k_rxframe64(RxFrame64Params p) {
    constexpr int S = 80, CP = 16;
    constexpr int ND = GUARD ? 48 : 64;
    constexpr int SYM_BYTES = ND * BPS / 8;   // a multiple of 4, or 6 (BPSK with guard bands: the reference's default frame): eight symbols are whole dwords either way
    constexpr int REGION_DW = ND * BPS / 4;   // 8 symbols
    constexpr int SLAB = 8 * 72;
    __shared__ cf slab_all[4 * SLAB];
    __shared__ cf ginv_all[4 * 64];           // 1 / H per wavefront, read at use (16 fewer live registers than a register copy)
    __shared__ unsigned img_all[4 * REGION_DW];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = lane >> 3, t = lane & 7;
    cf *buf = slab_all + wave * SLAB + s * 72;
    unsigned *img = img_all + wave * REGION_DW;

    // Lane constants that k_demod64 keeps in registers live in LDS here (22 VGPRs): with a second set of sample registers for the
    // next group's prefetch the kernel would otherwise spill at three waves per SIMD, and a spill reload waits for vmcnt(0),
    // i.e. for the prefetch.  twl[m] = W64^m (the stage twiddles W64^(r t) and the one-point-per-lane transform's twiddles: as
    // global loads their six loop-invariant 64-bit addresses are hoisted by the compiler and one of them spills);
    // bofftab[m * 64 + lane] = bit offset of bin t + 8 m, -1 = not data.
    __shared__ cf twl[64];
    __shared__ int bofftab[8 * 64];
    if (threadIdx.x < 64) twl[threadIdx.x] = p.tw[threadIdx.x];
    if (wave == 0) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int c = t + 8 * m;
            const int q = GUARD ? data_classes_below64(c) : c;
            bofftab[m * 64 + lane] = (carrier_class64(c, GUARD) == 0) ? (s * ND + q) * BPS : -1;
        }
    }
    __shared__ int cutbuf[4][1 + 16];         // MODE 0: per wavefront, the cut frames not yet on the global list (batched: one atomic per 16)
    if (MODE == 0 && lane == 0) cutbuf[wave][0] = 0;
    __syncthreads();
    const int wr = swz(8 * t);
    auto cut_flush = [&]() { // lane 0 only; this wavefront's own slots: no other wavefront reads or writes them
        const int m = cutbuf[wave][0];
        if (m > 0) {
            const int base = atomicAdd(p.cut_count, m);
            for (int i = 0; i < m; ++i) p.cut_list[base + i] = cutbuf[wave][1 + i];
            cutbuf[wave][0] = 0;
        }
    };

    const long long n_items = p.frame_list ? (long long)*p.frame_count : p.n_frames;
    // The per-frame scalars (live symbols, offset, CFO) of the NEXT frame of this wavefront are fetched while the current frame is
    // received: read where they are used -- symbol count, branch, offset, CFO, one after the other -- they cost three to four
    // dependent round trips to HBM per frame before the first sample is even requested.
    const long long istep = (long long)gridDim.x * 4;
    // (A VGPR zero the compiler cannot fold keeps these wave-uniform loads per-lane loads: as scalar values they would be moved to
    // SGPRs, and waited for, right where they are issued.)
    int vzero = 0;
    asm volatile("" : "+v"(vzero));
    long long f_n = 0; int ns_v = 0, off_v = 0; double fd_v = 0.0;
    auto fetch_scalars = [&](long long item) {
        if (item < n_items) {
            f_n = p.frame_list ? (long long)p.frame_list[item] : item;
            const long long fi = f_n + vzero;
            ns_v = p.nsym[fi];
            off_v = p.offset ? p.offset[fi] : 0;
            fd_v = p.f_delta ? p.f_delta[fi] : 0.0;
        }
    };
    fetch_scalars((long long)blockIdx.x * 4 + wave);
    for (long long item = (long long)blockIdx.x * 4 + wave; item < n_items; item += istep) {
        const long long f = f_n;
        const int ns = __builtin_amdgcn_readfirstlane(ns_v);
        const long long off = __builtin_amdgcn_readfirstlane(off_v);
        const double turns = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(fd_v)), __builtin_amdgcn_readfirstlane(__double2loint(fd_v))) * 0.15915494309189533577;
        fetch_scalars(item + istep);   // in flight until the next iteration reads them
        if (ns <= 0) { if (p.final_out && lane == 0) p.final_len[f] = 0; continue; } // wave-uniform
        const cf st = cfo_phasor(turns, 8);
        int keep = 0; // fused finish: bytes of this frame's output (known once the first group is demodulated)
        const cf *src = p.in + f * p.frame_stride + off;
        const long long avail = p.frame_len - off; // samples of the trimmed frame

        auto frame_body = [&](auto cut_tag) {
        constexpr bool CUT = decltype(cut_tag)::value;
        // estimate_channel (receiver.rs:212-229) on the 5 training blocks (chunks 5..9): H = mean_b FFT(block_b) / training
        // = FFT(mean_b block_b) / training.  Lane n sums sample n of the 5 derotated blocks, the wavefront transforms the 64
        // sums with one point per lane (lane_fft64), lane l then holds bin bitrev6(l): ONE transform per frame instead of a
        // whole 8-symbol group iteration.  1/H goes through the wave's LDS slab into the (t + 8 m) register layout.
        // The frame's round trips to HBM overlap instead of following one another: the first data group's samples are requested
        // right behind the training blocks (before the channel estimate is computed), every further group while the one before it
        // is transformed.
        // Every load is issued unconditionally, in straight-line code, from an address that is always mapped: loads under
        // exec-mask branches (or on one side of a branch that joins before their first use) make the compiler wait for
        // vmcnt(0) where the FIRST of them is used, i.e. for the prefetch as well.  Lanes of symbols past the frame's count
        // transform the frame's first samples: their image words are never stored.  CUT (a capture that ends inside the frame,
        // wave-uniform per frame) is a second instance of the whole body: there elements at or beyond frame_len read the frame's
        // first samples too and are zeroed where the registers are taken (zmask, bit m).
        const cf *safe = p.in + f * p.frame_stride + t;
        auto issue_group = [&](int k0, cf *v, unsigned &zmask) {
            const int count = ns - k0 < 8 ? ns - k0 : 8;
            const int n0 = (10 + k0 + s) * S + CP + t; // sample id of this lane's first point
            const cf *base = s < count ? src + n0 : safe;
            zmask = 0u;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const bool pad = CUT && s < count && (n0 + 8 * m) >= avail;
                v[m] = ld_cf(pad ? safe : base + 8 * m);
                if (CUT) zmask |= pad ? 1u << m : 0u;
            }
        };
        cf *ginv = ginv_all + wave * 64;
        cf v[8];
        unsigned zm = 0u;
        {
            cf tws[6]; // twiddles of the one-point-per-lane transform
            int lq = lane;
            asm volatile("" : "+v"(lq)); // per frame: keeps the six LDS addresses out of the loop-invariant registers (they spilled)
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int h = 32 >> q;
                const cf x = twl[(lq & (h - 1)) << q];
                tws[q] = (lane & h) ? x : make_float2(1.f, 0.f);
            }
            const int bin = bitrev6(lane);
            const cf invt = p.inv_training[bin]; // requested with the training blocks (read after the transform it is one more dependent round trip)
            const int nb = 5 * S + CP + lane;
            cf xb[5];
#pragma unroll
            for (int b = 0; b < 5; ++b) xb[b] = ld_cf((!CUT || (nb + S * b) < avail) ? src + nb + S * b : safe);
            issue_group(0, v, zm);
#pragma unroll
            for (int b = 0; b < 5; ++b) if (CUT && (nb + S * b) >= avail) xb[b] = make_float2(0.f, 0.f);
            cf acc;
            if (p.f_delta) { // sum_b x_b e^{-j phi (nb + 80 b)}: Horner in the 80-sample step, then this lane's phasor
                const cf s80 = cfo_phasor(turns, S), q0 = cfo_phasor(turns, nb);
                acc = cadd(cmul(xb[4], s80), xb[3]);
                acc = cadd(cmul(acc, s80), xb[2]);
                acc = cadd(cmul(acc, s80), xb[1]);
                acc = cadd(cmul(acc, s80), xb[0]);
                acc = cmul(acc, q0);
            } else acc = cadd(cadd(cadd(xb[4], xb[3]), cadd(xb[2], xb[1])), xb[0]);
            acc = lane_fft64(acc, lane, tws);
            cf h = cmul(acc, invt);
            h = make_float2(h.x * 0.2f, h.y * 0.2f);
            if (p.hk) p.hk[f * 64 + bin] = h;
            const float rn = __builtin_amdgcn_rcpf(h.x * h.x + h.y * h.y);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    // the previous frame's last reads of this table are done
            ginv[bin] = make_float2(h.x * rn, -h.y * rn);            // 1 / H
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        for (int k0 = 0; k0 < ns; k0 += 8) { // data symbols, 8 at a time (chunks 10..)
            const int count = ns - k0 < 8 ? ns - k0 : 8;
            const int n0 = (10 + k0 + s) * S + CP + t; // sample id of this lane's first point
            cf vn[8];
            unsigned zn = 0u;
            issue_group(k0 + 8 < ns ? k0 + 8 : k0, vn, zn); // the next group (past the last one: this group again, never used)
            if (CUT && zm) {
#pragma unroll
                for (int m = 0; m < 8; ++m) if ((zm >> m) & 1u) v[m] = make_float2(0.f, 0.f);
            }
            if (p.f_delta) { // CFO derotation, sample ids count from the trimmed start (receiver.rs:44-50)
                cf ph = cfo_phasor(turns, n0);
#pragma unroll
                for (int m = 0; m < 8; ++m) { v[m] = cmul(v[m], ph); ph = cmul(ph, st); }
            }
            bfly8<false>(v);
#pragma unroll
            for (int r = 0; r < 8; ++r) buf[wr ^ r] = v[r];
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = buf[8 * m + (t ^ m)];
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], twl[r * t]);
            bfly8<false>(v);
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = cmul(v[m], ginv[t + 8 * m]); // equalise (receiver.rs:68-70)
            cf rot = make_float2(1.f, 0.f);
            if (GUARD) {
                cf pv = make_float2(1.f, 0.f);
                pv = (t == 6) ? v[0] : pv;
                pv = (t == 1) ? v[3] : pv;
                pv = (t == 7) ? v[4] : pv;
                pv = (t == 2) ? v[7] : pv;
                const float trn = sum8(__ocml_atan2pi_f32(pv.y, pv.x)) * 0.125f; // mean pilot angle in turns -> hardware sin / cos
                rot = make_float2(__builtin_amdgcn_cosf(trn), -__builtin_amdgcn_sinf(trn)); // applied inside the demapper
            }
            for (int i = lane; i < REGION_DW; i += 64) img[i] = 0u;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int bo = bofftab[m * 64 + lane];
                if (bo >= 0) {
                    const unsigned idx = GUARD ? demap_point_rot(v[m], rot, BPS) : demap_point(v[m], BPS);
                    const int wd = bo >> 5, sh = bo & 31;
                    atomicOr(&img[wd], idx << sh);
                    if (BPS > 1 && (32 % BPS) != 0) {
                        if (sh + BPS > 32) atomicOr(&img[wd + 1], idx >> (32 - sh));
                    }
                }
            }
            // the next group's samples leave the prefetch registers BEFORE this group's bytes are stored: loads and stores share the
            // in-order VM counter, so a wait for the loads placed behind the stores would wait for the stores as well
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = vn[m];
            zm = zn;
            const int ndw = (count * SYM_BYTES + 3) / 4;   // (6-byte symbols: an odd count ends in half a dword; its upper half is zero in the image and lies inside the row)
            if (p.final_out) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the image is complete
                if (k0 == 0) { // bincode fixint little-endian u128 length (src/packets/mod.rs:20-32), then Vec::truncate
                    const unsigned long long lo = (unsigned long long)img[0] | ((unsigned long long)img[1] << 32);
                    const unsigned long long hi = (unsigned long long)img[2] | ((unsigned long long)img[3] << 32);
                    const int body = ns * SYM_BYTES - 16;
                    keep = (hi == 0 && lo < (unsigned long long)body) ? (int)lo : body;
                    if (lane == 0) p.final_len[f] = keep;
                }
                unsigned char *fo = p.final_out + f * p.final_stride;
                for (int i = lane; i < ndw; i += 64) {
                    const int ob = k0 * SYM_BYTES + 4 * i - 16; // final byte index of this dword
                    if (ob < 0 || ob >= keep) continue;
                    if (ob + 4 <= keep) *reinterpret_cast<unsigned *>(fo + ob) = img[i];
                    else for (int j = 0; ob + j < keep; ++j) fo[ob + j] = (unsigned char)(img[i] >> (8 * j));
                }
            } else {
                unsigned *dst = reinterpret_cast<unsigned *>(p.out + f * p.out_stride + (long long)k0 * SYM_BYTES);
                for (int i = lane; i < ndw; i += 64) dst[i] = img[i];
            }
        }
        };
        const bool whole = (long long)(10 + ns) * S <= avail; // wave-uniform
        if (MODE == 0) {
            if (whole) frame_body(std::false_type{});
            else if (lane == 0) { // left to the MODE 1 launch
                const int m = cutbuf[wave][0];
                cutbuf[wave][1 + m] = (int32_t)f;
                cutbuf[wave][0] = m + 1;
                if (m + 1 == 16) cut_flush();
            }
        } else if (MODE == 1) {
            if (!whole) frame_body(std::true_type{});
        } else {
            if (whole) frame_body(std::false_type{}); else frame_body(std::true_type{});
        }
    }
    if (MODE == 0 && lane == 0) cut_flush();
}

template <int BPS, int MODE> static void launch_rxframe_m(const RxFrame64Params &p, bool guard, dim3 grid, hipStream_t st) {
    if (guard) hipLaunchKernelGGL((k_rxframe64<BPS, true, MODE>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_rxframe64<BPS, false, MODE>), grid, dim3(256), 0, st, p);
}
// split = the caller gave a cut-list workspace and no frame list of its own: MODE 0 over every frame, then MODE 1 over the frames it left
template <int BPS> static hipError_t launch_rxframe(RxFrame64Params p, bool guard, dim3 grid, hipStream_t st, bool split) {
    if (!split) { launch_rxframe_m<BPS, 2>(p, guard, grid, st); return hipGetLastError(); }
    hipError_t e = hipMemsetAsync(p.cut_count, 0, sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    launch_rxframe_m<BPS, 0>(p, guard, grid, st);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    p.frame_list = p.cut_list; p.frame_count = p.cut_count;
    launch_rxframe_m<BPS, 1>(p, guard, grid, st); // the same persistent grid: a batch of captures that are ALL cut short must not crawl through 64 workgroups (an empty list costs microseconds either way)
    return hipGetLastError();
}

// Fused channel estimate + demod for N = 64 frames.  hipErrorNotSupported => caller uses run_chest + run_demod.
hipError_t run_rxframe64(const SymParams &sp, float2 *hk_out, hipStream_t st, int num_cu, unsigned char *final_out,
                         long long final_stride, int32_t *final_len, const int32_t *frame_list, const int32_t *frame_count,
                         int32_t *cut_ws) {
    const int nd = sp.guard ? 48 : 64;
    if ((nd * sp.bps) % 16 != 0 || !sp.nsym_frame || sp.soft) return hipErrorNotSupported;   // whole bytes per symbol, whole dwords per 8 symbols
    // raw rows are written with dword stores (the fused finish does not touch them)
    const bool to_final = final_out && final_len && (reinterpret_cast<uintptr_t>(final_out) & 3) == 0 && (final_stride & 3) == 0;
    if (!to_final && ((reinterpret_cast<uintptr_t>(sp.out_bytes) & 3) || (sp.out_stride & 3))) return hipErrorNotSupported;
    if (sp.n_frames <= 0) return hipSuccess;
    RxFrame64Params p;
    p.in = sp.in; p.n_frames = sp.n_frames; p.frame_stride = sp.frame_stride; p.frame_len = sp.frame_len;
    p.offset = sp.offset; p.f_delta = sp.f_delta; p.nsym = sp.nsym_frame; p.tw = sp.tw; p.inv_training = sp.inv_training;
    p.out = sp.out_bytes; p.out_stride = sp.out_stride; p.hk = hk_out;
    p.final_out = nullptr; p.final_stride = 0; p.final_len = nullptr;
    if (to_final) { p.final_out = final_out; p.final_stride = final_stride; p.final_len = final_len; }
    else if (final_out) return hipErrorNotSupported;
    p.frame_list = frame_list; p.frame_count = frame_count;
    // The split pays when cut captures are the exception (measured on 1 M config-3 frames, same box: chain 4.98 -> 4.68 ms / 4.98 -> 4.88 ms on
    // two boxes; with EVERY capture cut short the pair costs 18 % more than the one kernel: a skip pass plus the list): it is used when the
    // capture has room for the longest frame the caller asks for plus a 64-sample start offset -- what a slotted capture looks like.
    const bool roomy = sp.frame_len >= (long long)(10 + sp.syms_per_frame) * 80 + 64;
    const bool split = cut_ws != nullptr && frame_list == nullptr && roomy && !tuning_or_default(sp.tune).no_rxframe64_split;
    p.cut_count = split ? cut_ws : nullptr; p.cut_list = split ? cut_ws + 4 : nullptr;
    long long blocks = (sp.n_frames + 3) / 4, cap = frame_list ? 64 : (long long)num_cu * 8;
    { const long long gc = tuning_or_default(sp.tune).grid_cap; if (gc > 0 && gc < cap) cap = gc; }
    const dim3 grid((unsigned)(blocks < cap ? blocks : cap));
    trace_add(sp.trace, frame_list ? "k_rxframe64<list>" : (p.final_out ? "k_rxframe64<finish>" : "k_rxframe64"));
    if (split) trace_add(sp.trace, "k_rxframe64<cut,list>");
    switch (sp.bps) {
    case 2: return launch_rxframe<2>(p, sp.guard != 0, grid, st, split);
    case 4: return launch_rxframe<4>(p, sp.guard != 0, grid, st, split);
    case 6: return launch_rxframe<6>(p, sp.guard != 0, grid, st, split);
    case 8: return launch_rxframe<8>(p, sp.guard != 0, grid, st, split);
    case 1: return launch_rxframe<1>(p, sp.guard != 0, grid, st, split);
    default: return hipErrorNotSupported;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_txframe64: encode (src/transmitter.rs:11-58) for N = 64, one 256-thread workgroup per frame, ONE pass over HBM.
//   The data symbols are built in LDS (D x 80 samples): per group of 8 symbols a wavefront maps the byte stream to
//   constellation points straight into the Stockham input pattern (modulate + encode_block, transmitter.rs:108-165),
//   runs the inverse FFT64 in the k_demod64 layout (the symbol's own LDS slot doubles as the transpose slab), adds the
//   cyclic prefix (prefix_block, transmitter.rs:168-181) and tracks max(re, im).  After one barrier the workgroup knows
//   the frame's signed maximum (normalize, transmitter.rs:183-194, the constant header's maximum comes from the host)
//   and streams [header | data] / max to HBM with 16-byte stores.  HBM traffic: payload in, frame out, nothing else
//   (the two-kernel path wrote the frame, read it back and wrote it again).
struct TxFrame64Params {
    const uint8_t *payload;
    long long payload_stride;
    const int32_t *payload_len;
    int payload_bytes;
    long long n_frames;
    int n_sym;                 // data symbols per frame
    const float2 *tw;          // exp(-2 pi i m / 64)
    const float2 *header;      // 10 constant blocks (800 samples), unnormalised
    float header_max;
    float2 *out;
    long long out_stride;      // samples
    int debug;                 // OFDM_TX_DEBUG=1: the first two samples of every frame carry section times (s_memtime ticks)
};

// Workgroup barrier for LDS hand-offs that leaves global loads in flight (__syncthreads() also drains vmcnt, which
// would turn the next frame's payload prefetch into a stall).
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int BPS, bool GUARD>
__global__ __launch_bounds__(256, 4) void k_txframe64(TxFrame64Params p) {
    constexpr int S = 80, CP = 16, HDR = 10 * S;
    constexpr int ND = GUARD ? 48 : 64;
    constexpr int SYM_BITS = ND * BPS;
    extern __shared__ __align__(16) unsigned char smem[];
    cf *hd = reinterpret_cast<cf *>(smem);                      // [800] the constant header, staged once
    cf *fb = hd + HDR;                                          // [ceil8(n_sym) * 80] data symbols, unnormalised
    const int groups = (p.n_sym + 7) >> 3;
    unsigned *mxw = reinterpret_cast<unsigned *>(fb + (size_t)groups * 8 * S); // [2] frame maximum (float bits, >= 0), by frame parity
    unsigned *sbw_all = mxw + 4;                                               // [2][groups * 128 + 8] the frame's byte stream as dwords, by frame parity
    const int sbw_dw = groups * 128 + 8;
    const int stream_dw = groups * 8 * (SYM_BITS / 8) / 4 + 1;  // dwords the mapper may touch (one dword of slack)

    const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x, nwaves = nthr >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane >> 3, t = lane & 7;
    cf w[7];
#pragma unroll
    for (int r = 1; r < 8; ++r) { const cf x = p.tw[r * t]; w[r - 1] = make_float2(x.x, -x.y); } // conjugate: inverse transform
    int qoff[8]; // bit offset of bin (t + 8m)'s field inside one symbol's stream, -1 = null, -2 = pilot
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int c = t + 8 * m, cls = carrier_class64(c, GUARD);
        qoff[m] = cls == 0 ? (GUARD ? data_classes_below64(c) : c) * BPS : (cls == 2 ? -2 : -1);
    }
    const int wr = swz(8 * t);
    for (int i = tid; i < HDR; i += nthr) hd[i] = p.header[i];
    __shared__ float lvl[16]; // constellation levels by raw bit field, from axis_level itself (bit-identical): 2 LDS reads per point
    if (tid < 16) lvl[tid] = BPS > 1 && tid < (1 << (BPS >> 1)) ? axis_level((unsigned)tid, BPS >> 1) : 0.f;

    // The byte stream = 16-byte little-endian length (src/packets/mod.rs:20-32) + payload + zeros.  Dword i of the
    // PAYLOAD is fetched by thread (i mod nthr); the first two per thread are prefetched one frame ahead.
    const bool aligned = ((reinterpret_cast<uintptr_t>(p.payload) | (uintptr_t)p.payload_stride) & 3) == 0;
    auto pay_dword = [&](const uint8_t *pay, long long len, int i) -> unsigned { // payload bytes 4i .. 4i+3, zero past len
        const long long b0 = 4LL * i;
        if (b0 + 4 <= len && aligned) return reinterpret_cast<const unsigned *>(pay)[i];
        unsigned v = 0;
        for (int j = 0; j < 4; ++j) if (b0 + j < len) v |= (unsigned)pay[b0 + j] << (8 * j);
        return v;
    };
    // The prefetch is issued without a branch and without knowing the frame's length (a load under a branch, or behind the
    // scalar load of payload_len[f], is a synchronous one): dwords wholly inside the payload ROW (payload_bytes, mapped
    // whatever the frame's own length) are loaded as they are and cut to the frame's length when they are taken out of
    // the registers; the ragged last dword of a row whose size is not a multiple of 4 is rebuilt byte by byte there.
    // payload_len[f] itself travels with them as a per-lane load (a VGPR zero the compiler cannot fold keeps it off the
    // scalar unit, whose loads are waited for at the next LDS wait).
    int vzero = 0;
    asm volatile("" : "+v"(vzero));
    const bool row0 = aligned && 4LL * tid + 4 <= p.payload_bytes, row1 = aligned && 4LL * (tid + nthr) + 4 <= p.payload_bytes;
    auto issue = [&](long long fr, unsigned &d0, unsigned &d1, int &ln) {
        const long long fc = fr < p.n_frames ? fr : p.n_frames - 1; // past the batch: any mapped row, the values are never used
        const uint8_t *row = p.payload + fc * p.payload_stride;
        d0 = *reinterpret_cast<const unsigned *>(row0 ? row + 4 * tid : reinterpret_cast<const uint8_t *>(p.tw));
        d1 = *reinterpret_cast<const unsigned *>(row1 ? row + 4 * (tid + nthr) : reinterpret_cast<const uint8_t *>(p.tw));
        ln = p.payload_len ? p.payload_len[fc + vzero] : p.payload_bytes;
    };
    auto settle = [&](unsigned raw, bool whole, const uint8_t *pay, long long len, int i) -> unsigned {
        const long long keep = len - 4LL * i;                    // payload bytes from this dword on
        if (whole) return keep >= 4 ? raw : (keep <= 0 ? 0u : raw & ((1u << (8 * (int)keep)) - 1u));
        return pay_dword(pay, len, i);
    };
    // Per frame: [barrier] map + IFFT out of this frame's byte stream -> next frame's bytes out of the prefetch registers
    // into the OTHER stream buffer, the frame after next's loads issued -> [barrier] normalise + store.  The prefetched
    // loads are waited for BEFORE this frame's stores are issued: loads and stores share the in-order VM counter, and a
    // wait placed after the stores (the round-2 layout took the registers at the top of the next frame) drains them all.
    auto fill = [&](unsigned *sbw, long long fr, unsigned d0, unsigned d1, long long len) {
        const uint8_t *pay = p.payload + fr * p.payload_stride;
        if (tid < 4) sbw[tid] = tid < 2 ? (unsigned)((unsigned long long)len >> (32 * tid)) : 0u;
        if (tid + 4 < stream_dw) sbw[4 + tid] = settle(d0, row0, pay, len, tid);
        if (tid + nthr + 4 < stream_dw) sbw[4 + tid + nthr] = settle(d1, row1, pay, len, tid + nthr);
        for (int i = tid + 2 * nthr; i + 4 < stream_dw; i += nthr) sbw[4 + i] = pay_dword(pay, len, i); // long payloads
    };
    long long f = blockIdx.x;
    unsigned pre0 = 0, pre1 = 0;
    int pre_len = 0;
    long long len = 0;
    int cur = 0;
    if (f < p.n_frames) {
        issue(f, pre0, pre1, pre_len);
        len = row_len(__builtin_amdgcn_readfirstlane(pre_len), p.payload_bytes);
        fill(sbw_all, f, pre0, pre1, len);
        issue(f + gridDim.x, pre0, pre1, pre_len);
    }
    if (tid < 2) mxw[tid] = 0u;
    for (; f < p.n_frames; f += gridDim.x, cur ^= 1) {
        lds_only_barrier(); // this frame's byte stream is complete; the previous frame has left LDS (and the header is staged)
        const long long t0 = kProfile && p.debug ? (long long)__builtin_amdgcn_s_memtime() : 0;
        const unsigned char *sb = reinterpret_cast<const unsigned char *>(sbw_all + cur * sbw_dw);
        long long t1 = 0;
        const int npoints = (int)(((16 + len) * 8 + BPS - 1) / BPS); // points that carry stream bits; the rest are 0
        float lmax = 0.f;
        for (int g = wave; g < groups; g += nwaves) {
            const int k = 8 * g + s;       // this lane's symbol
            cf *buf = fb + k * S;          // its LDS slot (80 >= 72 entries: also the transpose slab)
            cf v[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                cf z = make_float2(0.f, 0.f);
                if (qoff[m] == -2) z = make_float2(1.f, 0.f);
                else if (qoff[m] >= 0) {
                    const int bit = k * SYM_BITS + qoff[m];
                    if (bit < npoints * BPS && k < p.n_sym) {
                        const unsigned two = (unsigned)sb[bit >> 3] | ((unsigned)sb[(bit >> 3) + 1] << 8);
                        const unsigned idx = (two >> (bit & 7)) & ((1u << BPS) - 1u);
                        z = BPS == 1 ? map_point(idx, 1) : make_float2(lvl[idx & ((1u << (BPS >> 1)) - 1u)], lvl[idx >> (BPS >> 1)]);
                    }
                }
                v[m] = z;
            }
            bfly8<true>(v);
#pragma unroll
            for (int r = 0; r < 8; ++r) buf[wr ^ r] = v[r];
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = buf[8 * m + (t ^ m)];
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], w[r - 1]);
            bfly8<true>(v);
            // v[q] = 64 x[t + 8q]; prefix_block: out = [x[48..64), x[0..64)]
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const cf y = make_float2(v[q].x * (1.0f / 64), v[q].y * (1.0f / 64));
                v[q] = y;
                lmax = fmaxf(lmax, fmaxf(y.x, y.y));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the slab reads above precede the overwrites below
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                buf[CP + t + 8 * q] = v[q];
                if (q >= 6) buf[t + 8 * q - 48] = v[q];
            }
        }
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, sh, 64));
        if (lane == 0) atomicMax(&mxw[cur], __float_as_uint(lmax)); // non-negative floats order like their bit patterns
        if (kProfile && p.debug) t1 = (long long)__builtin_amdgcn_s_memtime();
        long long len_next = 0;
        if (f + gridDim.x < p.n_frames) { // workgroup-uniform
            len_next = row_len(__builtin_amdgcn_readfirstlane(pre_len), p.payload_bytes);
            fill(sbw_all + (cur ^ 1) * sbw_dw, f + gridDim.x, pre0, pre1, len_next);
        }
        if (tid == 0) mxw[cur ^ 1] = 0u; // last read in the previous frame's store phase, which every thread left before this frame's first barrier
        issue(f + 2LL * gridDim.x, pre0, pre1, pre_len);
        lds_only_barrier();
        const long long t2 = kProfile && p.debug ? (long long)__builtin_amdgcn_s_memtime() : 0;
        const float inv = 1.0f / fmaxf(p.header_max, __uint_as_float(mxw[cur])); // one divide per thread, then multiplies (<= 1 ulp)
        len = len_next;
        // stream [header | data] out: header and data are contiguous in LDS, two samples per 16-byte store
        const int total2 = (HDR + p.n_sym * S) >> 1;
        float4 *dst = reinterpret_cast<float4 *>(p.out + f * p.out_stride);
        const float4 *src4 = reinterpret_cast<const float4 *>(hd);
        int i = tid;
        for (; i + 3 * nthr < total2; i += 4 * nthr) {
            const float4 a = src4[i], b = src4[i + nthr], c = src4[i + 2 * nthr], d = src4[i + 3 * nthr];
            dst[i] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
            dst[i + nthr] = make_float4(b.x * inv, b.y * inv, b.z * inv, b.w * inv);
            dst[i + 2 * nthr] = make_float4(c.x * inv, c.y * inv, c.z * inv, c.w * inv);
            dst[i + 3 * nthr] = make_float4(d.x * inv, d.y * inv, d.z * inv, d.w * inv);
        }
        for (; i < total2; i += nthr) { const float4 a = src4[i]; dst[i] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv); }
        if (kProfile && p.debug) {
            __syncthreads();
            if (tid == 0) {
                const long long t3 = (long long)__builtin_amdgcn_s_memtime();
                dst[0] = make_float4((float)(t1 - t0), (float)(t2 - t1), (float)(t3 - t2), 0.f); // map + IFFT, next frame's bytes, stores
            }
        }
    }
}

template <int BPS> static hipError_t launch_txframe(const TxFrame64Params &p, bool guard, dim3 grid, size_t lds, hipStream_t st) {
    const int groups = (p.n_sym + 7) / 8;
    const dim3 block(64u * (unsigned)(groups < 4 ? groups : 4)); // one wavefront per 8-symbol group, at most four
    if (guard) hipLaunchKernelGGL((k_txframe64<BPS, true>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((k_txframe64<BPS, false>), grid, block, lds, st, p);
    return hipGetLastError();
}
// Fused TX for N = 64 frames of up to 56 data symbols.  hipErrorNotSupported => caller uses k_sym<M_TX> + k_tx_finish.
hipError_t run_txframe64(const SymParams &sp, const float2 *header, float header_max, hipStream_t st, int num_cu) {
    const int n_sym = sp.syms_per_frame;
    if (n_sym <= 0 || n_sym > 56 || (!sp.payload && sp.payload_bytes)) return hipErrorNotSupported;
    if ((reinterpret_cast<uintptr_t>(sp.out) & 15) || (sp.out_stride_s & 1)) return hipErrorNotSupported; // 16-byte stores
    if (sp.n_frames <= 0) return hipSuccess;
    TxFrame64Params p;
    p.payload = sp.payload; p.payload_stride = sp.payload_stride; p.payload_len = sp.payload_len; p.payload_bytes = sp.payload_bytes;
    p.n_frames = sp.n_frames; p.n_sym = n_sym; p.tw = sp.tw; p.header = header; p.header_max = header_max;
    p.out = sp.out; p.out_stride = sp.out_stride_s;
    const Tuning &tu = tuning_or_default(sp.tune);
    p.debug = kProfile ? tu.debug_tx : 0;
    const int groups = (n_sym + 7) / 8;
    const size_t lds = (size_t)(800 + groups * 8 * 80) * sizeof(float2) + 16 + 2 * ((size_t)groups * 8 * 64 + 32); // header + frame + max + two byte streams
    long long per_cu = (long long)(160 * 1024) / (long long)lds;
    const int waves_per_cu = tu.tx_waves > 0 ? tu.tx_waves : 16; // tuning knob (measured best: 16)
    long long wave_cap = waves_per_cu / (groups < 4 ? groups : 4); // wavefronts per CU
    if (wave_cap < 1) wave_cap = 1;
    if (per_cu > wave_cap) per_cu = wave_cap;
    long long grid = (long long)num_cu * per_cu;
    if (tu.grid_cap > 0 && tu.grid_cap < grid) grid = tu.grid_cap;
    if (grid > sp.n_frames) grid = sp.n_frames;
    trace_add(sp.trace, "k_txframe64");
    switch (sp.bps) {
    case 1: return launch_txframe<1>(p, sp.guard != 0, dim3((unsigned)grid), lds, st);
    case 2: return launch_txframe<2>(p, sp.guard != 0, dim3((unsigned)grid), lds, st);
    case 4: return launch_txframe<4>(p, sp.guard != 0, dim3((unsigned)grid), lds, st);
    case 6: return launch_txframe<6>(p, sp.guard != 0, dim3((unsigned)grid), lds, st);
    case 8: return launch_txframe<8>(p, sp.guard != 0, dim3((unsigned)grid), lds, st);
    }
    return hipErrorNotSupported;
}

// ---------------------------------------------------------------------------------------------------------------
// k_demod4096: RX demod of N = 4096 symbols (BASELINE config 5) as 64 x 64: one 512-thread workgroup per symbol.
//     X[c + 64 d] = sum_b W64^(b d) * [ W4096^(b c) * sum_a x[64 a + b] W64^(a c) ]
//   stage A  wavefront w, 8-lane group s: the FFT64 over a for column b = 8 w + s, straight from HBM in the Stockham
//            pattern (the next symbol's loads are already in flight), in the k_demod64 layout: wave-local, no barrier;
//   twiddle  W4096^(b c): eight loop-invariant registers per lane;
//   transpose through LDS ([c][b], one barrier);
//   stage B  the FFT64 over b for row c = 8 w + s, again wave-local;
//   epilogue equalise, mean pilot angle over the 256 pilots (wave sums + one LDS step), hard decisions, LSB-first packing
//            through LDS, dword stores -- the same arithmetic as k_sym<4096, M_DEMOD>.
// Four radix-8 butterflies per point like the generic kernel, but ONE workgroup-wide exchange instead of three, four
// barriers per symbol instead of nine, and ~110 VGPRs (two workgroups per CU instead of one).
struct Big4096Params {
    const float2 *in;
    long long frame_stride;
    long long total;          // symbols
    int syms_per_frame, first_symbol;
    long long step_f;         // frames / symbols that one grid step (gridDim.x symbols) advances: no division in the loop
    int step_k;
    const float2 *tw;         // exp(-2 pi i m / 4096), m < 4096
    const float2 *hk;         // optional channel, hk_stride = 0 (shared) or 4096 (per frame)
    long long hk_stride;
    unsigned char *out;
    long long out_stride;
    int bps, guard;
    // FRAME = true (the decode chain after timing, src/receiver.rs:20-83): per-frame start of the trimmed frame, CFO and live
    // symbol count; samples at or past frame_len read as zero (pad_chunk, receiver.rs:203-210)
    const int32_t *offset;
    const double *f_delta;
    const int32_t *nsym_frame;
    long long frame_len;
};

template <int BPS, bool GUARD, bool FRAME>
__global__ __launch_bounds__(512, 4) void k_demod4096(Big4096Params p) { // 4 waves per SIMD = two workgroups per CU: never more than 128 VGPRs
    constexpr int N = 4096, S = 5120, CP = 1024, TS = 72, SLAB = 8 * 72;
    constexpr int ND = GUARD ? 48 * 64 : N;
    constexpr int IMG_DW = ND * BPS / 32;                        // packed bytes of one symbol, in dwords (<= 1024)
    extern __shared__ __align__(16) unsigned char smem[];
    cf *slab_all = reinterpret_cast<cf *>(smem);                 // [8 waves][8 x 72] FFT64 transpose slabs (stages A and B)
    cf *T = slab_all + 8 * SLAB;                                 // [64][72]  Z[c][b]
    unsigned *img = reinterpret_cast<unsigned *>(T + 64 * TS);   // [IMG_DW] the symbol's packed output image
    float *red = reinterpret_cast<float *>(img + 1024);          // [8] wave sums of the pilot angles

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane >> 3, t = lane & 7;
    const int col = 8 * wave + s;                 // b in stage A, c in stage B
    cf *buf = slab_all + wave * SLAB + s * 72;
    const int wr = swz(8 * t);

    // W64^(r t), r = 1 .. 7, read from a 56-entry LDS table at use: as fourteen loop-invariant registers they pushed the kernel over its
    // 128 VGPRs, and a spilled register comes back through scratch -- a VMEM load, in order behind the next symbol's prefetch, so that every
    // reload waited for the prefetch (round-5 ISA scan: 3 to 10 spilled registers in every instantiation, each reloaded right after a barrier)
    cf *wtab = reinterpret_cast<cf *>(reinterpret_cast<unsigned char *>(red) + 64 + 256);   // [8][7], behind the pilot sums and the frame-mode offset table
    if (tid < 56) wtab[tid] = p.tw[64 * (tid % 7 + 1) * (tid / 7)];
    __syncthreads();
    const cf *w = wtab + 7 * t;
    cf z[8];                                      // W4096^(b c), c = t + 8 q
#pragma unroll
    for (int q = 0; q < 8; ++q) z[q] = p.tw[col * (t + 8 * q)];
    // bit offset of bin c + 64 d (d = t + 8 q) in the image, -1 = not a data bin.  Stream mode keeps the eight of them in registers;
    // frame mode (offsets, CFO phasors, the zero-fill masks of the branch-free fetch on top) spilled 8-17 registers at 128 VGPRs, so
    // there the offset is rebuilt from a 64-entry LDS table of its column-independent part (a spill reload waits for the prefetch)
    int boff_r[8];
    int *btab = reinterpret_cast<int *>(red + 16); // [64] (FRAME only; red + 8 is the spare dword img[1024 + 8] of the packing below)
    if (FRAME) {
        if (tid < 64) { const int d = (tid & 7) + 8 * (tid >> 3); btab[tid] = carrier_class64(d, GUARD) == 0 ? (GUARD ? data_classes_below64(d) : d) * 64 * BPS : -1; }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int d = t + 8 * q;
        boff_r[q] = FRAME ? 0 : (carrier_class64(d, GUARD) == 0 ? ((GUARD ? data_classes_below64(d) : d) * 64 + col) * BPS : -1);
    }
    auto boff = [&](int q) -> int {
        if (!FRAME) return boff_r[q];
        const int b0 = btab[8 * q + t];
        return b0 < 0 ? -1 : b0 + col * BPS;
    };
    constexpr int nbytes = ND * BPS / 8;

    // (frame, symbol) of the symbol one step ahead (the prefetch); advanced by the host's per-step increments -- the two 64-bit
    // divisions per symbol this replaces were a third of the loop's instruction stream (408 of 1214, all scalar)
    long long fn = blockIdx.x / p.syms_per_frame;
    int kn = (int)(blockIdx.x - fn * p.syms_per_frame);
    // Loads are issued without a branch (a load under `if (in range)` is followed by s_waitcnt vmcnt(0) at the join, i.e. is
    // synchronous): out-of-range elements read the twiddle table instead and are zeroed when they leave the prefetch registers.
    // room = samples from this lane's first one to the end of the capture (FRAME), 0 past the batch.
    // FRAME: the frame's scalars (trimmed start, live symbols, CFO) are requested a step ahead (FS tq) and taken where the next
    // prefetch starts: read where they were used, each was followed by s_waitcnt vmcnt(0) -- the start offset ahead of the sample
    // prefetch, the symbol count right behind it (draining it), the CFO, then the channel in four more round trips (round-5 ISA scan).
    struct FS { int off, ns; double fd; };
    int vzero = 0;
    asm volatile("" : "+v"(vzero));   // keeps the workgroup-uniform scalar loads vector loads (as scalar loads they are waited for where they are issued)
    auto load_scalars = [&](bool in, long long fr_) -> FS {
        const long long fr = (in ? fr_ : 0) + vzero;
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
        const long long n0 = off + (long long)(p.first_symbol + kk) * S + CP + col;
        const cf *src = p.in + fr * p.frame_stride + n0;
        long long rm = FRAME ? p.frame_len - n0 : (long long)N;
        rm = in ? rm : 0;
        room = (int)(rm < 0 ? 0 : (rm > N ? N : rm));
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int i = 64 * (t + 8 * m);
            const cf *a = FRAME ? (i < room ? src + i : p.tw + i) : (in ? src : p.tw) + i; // p.tw: N mapped entries
            dst[m] = *a;
        }
    };
    // The image of symbol j is stored to HBM in iteration j + 1, behind that iteration's FIRST barrier (which is what makes the
    // image complete: no barrier of its own) and long before the next wait for prefetched samples: loads and stores share the
    // in-order VM counter, so stores issued just before a wait-for-loads would be waited for as well.
    // image -> global, and clear it for the next symbol (same thread, same dwords).  Straight-line code: written as a loop over the
    // dwords, the compiler put s_waitcnt vmcnt(0) in front of it -- i.e. the whole workgroup waited here, once per symbol, for the NEXT
    // symbol's samples, which had been requested a few hundred instructions earlier (round-5 ISA scan).  IMG_DW is a multiple of 4 and at
    // most 1024: with 16-byte aligned rows one predicated 16-byte store per thread, else at most two dword stores.
    // (Whole-byte / nibble fields are written with plain stores, every dword of the image by exactly one lane: nothing to clear there;
    //  only the atomic-OR packing of 1-, 2- and 6-bit fields needs a zeroed image -- and a zero 4-vector was one more register than the
    //  kernel has: it came back from scratch, a VMEM load the prefetch had to be waited for behind.)
    const bool wide_out = ((reinterpret_cast<uintptr_t>(p.out) | (uintptr_t)p.out_stride) & 15) == 0;   // (nbytes is a multiple of 16)
    constexpr bool CLEAR = !(BPS == 8 || BPS == 4);
    auto flush = [&](unsigned *dst) {
        if (wide_out) {
            if (tid < IMG_DW / 4) {
                unsigned *i1 = img + 4 * tid;
                reinterpret_cast<uint4 *>(dst)[tid] = *reinterpret_cast<const uint4 *>(i1);
                if (CLEAR) { i1[0] = 0u; i1[1] = 0u; i1[2] = 0u; i1[3] = 0u; }
            }
        } else {
            if (tid < IMG_DW) { dst[tid] = img[tid]; if (CLEAR) img[tid] = 0u; }
            if (IMG_DW > 512 && tid + 512 < IMG_DW) { dst[tid + 512] = img[tid + 512]; if (CLEAR) img[tid + 512] = 0u; }
        }
    };
    for (int i = tid; i < IMG_DW; i += 512) img[i] = 0u;
    cf pre[8];
    int room_pre = 0;
    auto advance = [&](long long &ff, int &kk) { ff += p.step_f; kk += p.step_k; if (kk >= p.syms_per_frame) { kk -= p.syms_per_frame; ++ff; } };
    // positions: the symbol being transformed (f0, k0), the one whose samples are prefetched (fn, kn), the one whose scalars are (f2, k2)
    long long f0 = fn, f2; int k0 = kn, k2;
    advance(fn, kn);
    f2 = fn; k2 = kn; advance(f2, k2);
    FS tq = FS{0, 0, 0.0};
    int ns_cur = 0; double fd_cur = 0.0;
    {
        FS s0 = FS{0, 0, 0.0};
        if (FRAME) s0 = load_scalars(blockIdx.x < p.total, f0);
        fetch(blockIdx.x, f0, k0, s0.off, pre, room_pre);
        ns_cur = __builtin_amdgcn_readfirstlane(s0.ns);
        fd_cur = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(s0.fd)), __builtin_amdgcn_readfirstlane(__double2loint(s0.fd)));
        if (FRAME) tq = load_scalars((long long)blockIdx.x + gridDim.x < p.total, fn);
    }
    unsigned *pending = nullptr; // where the image currently in LDS belongs

    for (long long sg = blockIdx.x; sg < p.total; sg += gridDim.x) {
        const long long f = f0;
        const int k = k0;
        cf v[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = (!FRAME || 64 * (t + 8 * m) < room_pre) ? pre[m] : make_float2(0.f, 0.f);
        // the next symbol's scalars (requested a step ago, right behind this symbol's samples): workgroup-uniform -> SGPRs
        const int off_n = FRAME ? __builtin_amdgcn_readfirstlane(tq.off) : 0;
        const int ns_n = FRAME ? __builtin_amdgcn_readfirstlane(tq.ns) : 0;
        const double fd_n = FRAME ? __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(tq.fd)), __builtin_amdgcn_readfirstlane(__double2loint(tq.fd))) : 0.0;
        if (!FRAME) fetch(sg + gridDim.x, fn, kn, 0, pre, room_pre);   // stream mode: nothing else is loaded, the prefetch goes out first
        bool live = true;
        if (FRAME) {
            if (k >= ns_cur) live = false; // fewer symbols in this frame: nothing is written (workgroup-uniform)
            if (!live) {
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = make_float2(0.f, 0.f);
            } else if (p.f_delta) { // CFO derotation, sample ids count from the trimmed start (receiver.rs:44-50); phase reduced in f64
                const double turns = fd_cur * 0.15915494309189533577;
                cf ph = cfo_phasor(turns, (long long)(p.first_symbol + k) * S + CP + col + 64 * t);
                const cf st = cfo_phasor(turns, 512);
#pragma unroll
                for (int m = 0; m < 8; ++m) { v[m] = cmul(v[m], ph); ph = cmul(ph, st); }
            }
        }
        unsigned *const mine = live ? reinterpret_cast<unsigned *>(p.out + f * p.out_stride + (long long)k * nbytes) : nullptr;
        // ---- stage A: FFT64 over a (wave-local)
        bfly8<false>(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) buf[wr ^ r] = v[r];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = buf[8 * m + (t ^ m)];
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], w[r - 1]);
        bfly8<false>(v);
        // v[q] = Y_b[c = t + 8 q]; twiddle and transpose
#pragma unroll
        for (int q = 0; q < 8; ++q) T[(t + 8 * q) * TS + (col ^ (t & 6))] = cmul(v[q], z[q]); // column XOR-swizzled by the row: conflict-free both ways
        __syncthreads(); // T complete; the PREVIOUS symbol's image complete (two barriers per symbol, not three: see below)
        if (pending) flush(pending); // ... so it leaves for HBM here, cleared for this symbol's fields, which are written after the next barrier
        pending = mine;
        // FRAME: this symbol's channel is requested here and used behind stage B; only THEN does the next symbol's prefetch go out (its
        // registers are free while the channel's are live: no register more than before) -- requested at its use the channel came in four
        // serialized round trips, requested behind the prefetch its wait would drain the prefetch
        cf hkv[8];
        if (FRAME) {
            const cf *h = p.hk ? p.hk + f * p.hk_stride + col : p.tw;
#pragma unroll
            for (int q = 0; q < 8; ++q) hkv[q] = h[64 * (t + 8 * q)];
            __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler sinks these loads to their use, behind stage B: a round trip in the open)
        }
        // ---- stage B: FFT64 over b for row c = col
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = T[col * TS + (t ^ (col & 6)) + 8 * m]; // (t + 8 m) ^ s = (t ^ s) + 8 m for s < 8: offsets stay immediates
        bfly8<false>(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) buf[wr ^ r] = v[r];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = buf[8 * m + (t ^ m)];
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], w[r - 1]);
        bfly8<false>(v);
        // v[q] = X[col + 64 (t + 8 q)]
        if (p.hk) { // equalise: Y /= H (src/receiver.rs:68-70)
            const cf *h = p.hk + f * p.hk_stride;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const cf hh = FRAME ? hkv[q] : h[col + 64 * (t + 8 * q)];
                const float rn = __builtin_amdgcn_rcpf(hh.x * hh.x + hh.y * hh.y);
                const cf e = cmulc(v[q], hh);
                v[q] = make_float2(e.x * rn, e.y * rn);
            }
        }
        if (FRAME) {   // the next symbol's samples from the start offset that arrived with this symbol's, then the scalars of the one after
            fetch(sg + gridDim.x, fn, kn, off_n, pre, room_pre);
            tq = load_scalars(sg + 2 * (long long)gridDim.x < p.total, f2);
        }
        f0 = fn; k0 = kn; fn = f2; kn = k2; advance(f2, k2);
        ns_cur = ns_n; fd_cur = fd_n;
        cf rot = make_float2(1.f, 0.f);
        if (GUARD) { // decode_block (src/receiver.rs:106-145): mean angle of the 4 x 64 pilots, rotate by -phase
            // pilot classes 6, 25, 39, 58 = (t, q) = (6, 0), (1, 3), (7, 4), (2, 7); other lanes feed (1, 0) -> angle 0
            cf pv = make_float2(1.f, 0.f);
            pv = (t == 6) ? v[0] : pv;
            pv = (t == 1) ? v[3] : pv;
            pv = (t == 7) ? v[4] : pv;
            pv = (t == 2) ? v[7] : pv;
            float a = __ocml_atan2pi_f32(pv.y, pv.x);
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) a += __shfl_xor(a, sh, 64);
            if (lane == 0) red[wave] = a;
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) tot += red[i];
            const float trn = tot * (0.5f / 256.0f); // mean of the 256 pilot angles, in turns -> hardware sin / cos
            rot = make_float2(__builtin_amdgcn_cosf(trn), -__builtin_amdgcn_sinf(trn)); // applied inside the demapper
        }
        // demodulate (src/receiver.rs:147-190) and pack LSB-first (src/utils.rs:30-36).  The fields go into the image behind the pilot
        // barrier, which also orders them after the flush of the previous image above; without guard bands there is no pilot
        // barrier, so one stands here.  No barrier closes the step: the image is read (flushed) only behind the next step's first
        // barrier, T is rewritten only by waves that are past the pilot barrier (every stage-B read of T lies before it), and the
        // pilot sums are rewritten only behind the next step's first barrier.
        if (!GUARD) __syncthreads();
        if (BPS == 8 || BPS == 4) {
            // Whole-byte / nibble fields: the 32 / BPS lane groups s that share a dword (same t, so the same carrier class)
            // merge their fields in registers -- lane ^ 8 by DPP, lane ^ 16 by ds_swizzle, lane ^ 32 by a shuffle -- and ONE
            // lane writes the dword with a plain store: no atomics, no 4-way same-dword serialisation (round-2 ablation on
            // noise input: loads 0.49 ms, + stage A / transpose 0.06, + stage B / pilots 0.19, + demap / packing 0.28, + stores 0.10), and the image needs no clearing
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                // branch-free: rows q in {1, 2, 5, 6} hold data bins in every lane; elsewhere non-data lanes contribute 0 and
                // non-writing lanes store to a spare dword behind the image (exec-mask branches cost more than the stores)
                const bool all_data = !GUARD || q == 1 || q == 2 || q == 5 || q == 6;
                const int bo = boff(q);
                const bool data = live && (all_data || bo >= 0); // a dead symbol writes nothing into the image
                unsigned val = (GUARD ? demap_point_rot(v[q], rot, BPS) : demap_point(v[q], BPS)) << (BPS * (s & (32 / BPS - 1)));
                val = data ? val : 0u;
                val |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)val, 0x128, 0xF, 0xF, true);          // row_ror:8   : s ^ 1
                val |= (unsigned)__builtin_amdgcn_ds_swizzle((int)val, 0x401F);                               // xor 16      : s ^ 2
                if (BPS == 4) val |= (unsigned)__shfl_xor((int)val, 32, 64);                                  // s ^ 4
                img[(data && (s & (32 / BPS - 1)) == 0) ? (bo >> 5) : 1024 + 8] = val;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) { // OR every field into the image; a dead symbol (k >= nsym_frame[f]) must leave it clear:
                const int bo = boff(q);
                if (live && bo >= 0) { // nothing flushes the image after such a step, and demap_point(0) != 0 for BPS >= 2
                    const unsigned idx = GUARD ? demap_point_rot(v[q], rot, BPS) : demap_point(v[q], BPS);
                    const int wd = bo >> 5, sh = bo & 31;
                    atomicOr(&img[wd], idx << sh);
                    if (BPS > 1 && (32 % BPS) != 0) { // a field may straddle two dwords (only for 6-bit fields)
                        if (sh + BPS > 32) atomicOr(&img[wd + 1], idx >> (32 - sh));
                    }
                }
            }
        }
    }
    __syncthreads(); // the last image is complete
    if (pending) flush(pending);
}

// N = 4096 RX demod fast path: regular symbol streams, and the data symbols of frames after timing (per-frame offset, CFO,
// live-symbol count, zero-fill past the capture).  hipErrorNotSupported => caller uses k_sym<4096, M_DEMOD>.
hipError_t run_demod4096(const SymParams &sp, hipStream_t st, int num_cu) {
    if (sp.soft || sp.syms_per_frame <= 0) return hipErrorNotSupported;
    if (sp.in_sym_stride != 5120 || sp.in_skip != 1024) return hipErrorNotSupported;
    const bool frame = sp.offset || sp.f_delta || sp.nsym_frame ||
                       (long long)(sp.first_symbol + sp.syms_per_frame) * 5120 > sp.frame_len; // tail padding needs the bounds checks
    const int nd = sp.guard ? 48 * 64 : 4096;
    if ((nd * sp.bps / 8) % 4 != 0 || (reinterpret_cast<uintptr_t>(sp.out_bytes) & 3) || (sp.out_stride & 3)) return hipErrorNotSupported;
    if (sp.hk && sp.hk_stride != 0 && sp.hk_stride != 4096) return hipErrorNotSupported;
    Big4096Params p;
    p.in = sp.in; p.frame_stride = sp.frame_stride; p.total = sp.n_frames * (long long)sp.syms_per_frame;
    p.syms_per_frame = sp.syms_per_frame; p.first_symbol = sp.first_symbol; p.tw = sp.tw; p.hk = sp.hk; p.hk_stride = sp.hk_stride;
    p.out = sp.out_bytes; p.out_stride = sp.out_stride; p.bps = sp.bps; p.guard = sp.guard;
    p.offset = sp.offset; p.f_delta = sp.f_delta; p.nsym_frame = sp.nsym_frame; p.frame_len = sp.frame_len;
    if (p.total <= 0) return hipSuccess;
    const size_t lds = (size_t)(8 * 8 * 72 + 64 * 72) * sizeof(float2) + 4096 + 64 + 256 + 448; // slabs, T, image, pilot sums + spare dword, frame-mode offset table, stage twiddles
    long long grid = (long long)num_cu * 2;
    { const long long cap = tuning_or_default(sp.tune).grid_cap; if (cap > 0 && cap < grid) grid = cap; } // test hook, as kernels_mid.hip
    if (grid > p.total) grid = p.total;
    trace_add(sp.trace, frame ? "k_demod4096<frame>" : "k_demod4096");
    p.step_f = grid / p.syms_per_frame; p.step_k = (int)(grid - p.step_f * p.syms_per_frame);
    // > 64 KB of dynamic LDS: a per-device attribute, so set on every call (one process may drive several GPUs); once per batch
#define OFDM_LAUNCH_4096_F(B, G, F)                                                                                         \
    {                                                                                                                       \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_demod4096<B, G, F>),                            \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        if (e != hipSuccess) return e;                                                                                      \
        hipLaunchKernelGGL((k_demod4096<B, G, F>), dim3((unsigned)grid), dim3(512), lds, st, p);                            \
        return hipGetLastError();                                                                                           \
    }
#define OFDM_LAUNCH_4096(B, G) { if (frame) OFDM_LAUNCH_4096_F(B, G, true) else OFDM_LAUNCH_4096_F(B, G, false) }
    switch (sp.bps * 2 + (sp.guard ? 1 : 0)) {
    case 2: OFDM_LAUNCH_4096(1, false) case 3: OFDM_LAUNCH_4096(1, true)
    case 4: OFDM_LAUNCH_4096(2, false) case 5: OFDM_LAUNCH_4096(2, true)
    case 8: OFDM_LAUNCH_4096(4, false) case 9: OFDM_LAUNCH_4096(4, true)
    case 12: OFDM_LAUNCH_4096(6, false) case 13: OFDM_LAUNCH_4096(6, true)
    case 16: OFDM_LAUNCH_4096(8, false) case 17: OFDM_LAUNCH_4096(8, true)
    }
#undef OFDM_LAUNCH_4096_F
#undef OFDM_LAUNCH_4096
    return hipErrorNotSupported;
}

// ---------------------------------------------------------------------------------------------------------------
// k_tx4096: modulate + encode_block + prefix_block (src/transmitter.rs:108-181) for a continuous stream of N = 4096
// symbols (BASELINE config 5 TX), the mirror image of k_demod4096:
//     x[c + 64 d] = 1/N sum_b W64^(-b d) * [ W4096^(-b c) * sum_a X[64 a + b] W64^(-a c) ]
// The symbol's bytes are staged in LDS from dwords prefetched one symbol ahead; lane (s, t) of wavefront w builds bins
// 64 (t + 8 m) + b, b = 8 w + s (the carrier class depends on t + 8 m only), runs the inverse FFT64 over a, the twiddle,
// the LDS transpose and the inverse FFT64 over b, and stores samples c + 64 (t + 8 q) behind the cyclic prefix.
struct Tx4096Params {
    const uint8_t *bytes;
    long long n_bytes, n_sym;
    const float2 *tw;   // exp(-2 pi i m / 4096)
    float2 *out;        // n_sym x 5120 samples
    int bps, guard;
};

template <bool GUARD>
__global__ __launch_bounds__(512, 4) void k_tx4096(Tx4096Params p) {
    constexpr int N = 4096, S = 5120, CP = 1024, TS = 72, SLAB = 8 * 72;
    extern __shared__ __align__(16) unsigned char smem[];
    cf *slab_all = reinterpret_cast<cf *>(smem);
    cf *T = slab_all + 8 * SLAB;
    unsigned *sbw = reinterpret_cast<unsigned *>(T + 64 * TS);          // [1024 + 2] the symbol's bytes as dwords
    cf *ptab = reinterpret_cast<cf *>(sbw + 1024 + 4);                  // [256] map_point by raw bit field (transmitter.rs:108-140)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane >> 3, t = lane & 7;
    const int col = 8 * wave + s;
    cf *buf = slab_all + wave * SLAB + s * 72;
    const int wr = swz(8 * t);
    if (tid < (1 << p.bps)) ptab[tid] = map_point((unsigned)tid, p.bps);
    const unsigned fmask = (1u << p.bps) - 1u;
    cf w[7];
#pragma unroll
    for (int r = 1; r < 8; ++r) { const cf x = p.tw[64 * r * t]; w[r - 1] = make_float2(x.x, -x.y); }
    cf z[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const cf x = p.tw[col * (t + 8 * q)]; z[q] = make_float2(x.x, -x.y); }
    int boff[8]; // bit offset of bin 64 (t + 8 m) + col inside the symbol's stream, -1 = null, -2 = pilot
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int a = t + 8 * m, cls = carrier_class64(a, GUARD);
        boff[m] = cls == 0 ? ((GUARD ? data_classes_below64(a) : a) * 64 + col) * p.bps : (cls == 2 ? -2 : -1);
    }
    const int nd = GUARD ? 48 * 64 : N;
    const int sym_bytes = nd * p.bps / 8;   // <= 4096, multiple of 4 (checked by the launcher)
    const bool aligned = (reinterpret_cast<uintptr_t>(p.bytes) & 3) == 0;
    // stream bytes of symbol sg, two dwords per thread: issued without a branch (paydw_issue), settled where they are used
    const bool want0 = 4 * tid < sym_bytes, want1 = 4 * (tid + 512) < sym_bytes;
    auto fetch = [&](long long sg, unsigned &d0, unsigned &d1) {
        const long long base = sg * sym_bytes;
        const bool in = sg < p.n_sym;
        d0 = paydw_issue(p.bytes, base + 4 * tid, p.n_bytes, in && want0, aligned, p.tw);
        d1 = paydw_issue(p.bytes, base + 4 * (tid + 512), p.n_bytes, in && want1, aligned, p.tw);
    };
    auto settle = [&](long long sg, unsigned d0, unsigned d1) {
        const long long base = sg * sym_bytes;
        const bool in = sg < p.n_sym;
        sbw[tid] = paydw_settle(d0, p.bytes, base + 4 * tid, p.n_bytes, in && want0, aligned);
        sbw[tid + 512] = paydw_settle(d1, p.bytes, base + 4 * (tid + 512), p.n_bytes, in && want1, aligned);
    };
    // The bytes of symbol j+1 are taken out of the prefetch registers (and symbol j+2's loads issued) BEFORE symbol j's
    // samples are stored: loads and stores share the in-order VM counter, so a wait for prefetched loads placed after the
    // stores would wait for the stores as well.
    unsigned d0, d1;
    fetch(blockIdx.x, d0, d1);
    settle(blockIdx.x, d0, d1);
    if (tid < 2) sbw[1024 + tid] = 0u; // slack for the two-byte window
    fetch((long long)blockIdx.x + gridDim.x, d0, d1);
    __syncthreads();

    for (long long sg = blockIdx.x; sg < p.n_sym; sg += gridDim.x) {
        long long left = p.n_bytes - sg * sym_bytes;             // stream bytes that belong to this symbol
        left = left < 0 ? 0 : (left < sym_bytes ? left : sym_bytes);
        const int live_bits = (int)(((unsigned)left * 8u + (unsigned)p.bps - 1u) / (unsigned)p.bps) * p.bps; // fields that carry stream bits (left <= 4096); the rest are 0
        cf v[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            // constellation points from an LDS table filled with map_point itself (bit-identical values) instead of ~20
            // instructions of Gray decoding per axis (round-2 ablation: 0.21 of 1.09 ms); branch-free
            v[m] = tx_point<true>(sbw, ptab, boff[m], live_bits, fmask);
        }
        bfly8<true>(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) buf[wr ^ r] = v[r];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = buf[8 * m + (t ^ m)];
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], w[r - 1]);
        bfly8<true>(v);
#pragma unroll
        for (int q = 0; q < 8; ++q) T[(t + 8 * q) * TS + (col ^ (t & 6))] = cmul(v[q], z[q]); // column XOR-swizzled by the row: conflict-free both ways
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = T[col * TS + (t ^ (col & 6)) + 8 * m]; // (t + 8 m) ^ s = (t ^ s) + 8 m for s < 8: offsets stay immediates
        bfly8<true>(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) buf[wr ^ r] = v[r];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = buf[8 * m + (t ^ m)];
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], w[r - 1]);
        bfly8<true>(v);
        // v[q] = N x[col + 64 (t + 8 q)]; prefix_block: out = [x[N - CP .. N), x[0 .. N)]
        settle(sg + gridDim.x, d0, d1); // next symbol's bytes (every wavefront left the mapping stage two barriers ago)
        fetch(sg + 2 * (long long)gridDim.x, d0, d1);
        // transpose once more through T ([n >> 6][n & 63], the conflict-free layout of the first transpose) so that every
        // store is a full 16 bytes per lane and 1 KiB per wavefront
        __syncthreads(); // every wavefront has read its stage-B inputs out of T
#pragma unroll
        for (int q = 0; q < 8; ++q) T[(t + 8 * q) * TS + (col ^ (t & 6))] = make_float2(v[q].x * (1.0f / N), v[q].y * (1.0f / N));
        __syncthreads();
        {
            float4 *dst4 = reinterpret_cast<float4 *>(p.out + sg * S);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = tid + 512 * j, n = 2 * i;                      // sample pair (n, n + 1)
                const float4 y = *reinterpret_cast<const float4 *>(T + (n >> 6) * TS + ((n & 63) ^ ((n >> 6) & 6))); // an even XOR keeps the pair adjacent
                dst4[(CP >> 1) + i] = y;
                if (j == 3) dst4[i - ((N - CP) >> 1)] = y;                   // n >= N - CP: the cyclic prefix
            }
        }
        __syncthreads(); // sbw / T are reused by the next symbol
    }
}

// Continuous-stream TX for N = 4096.  hipErrorNotSupported => caller uses k_sym<4096, M_TX>.
hipError_t run_tx4096(const SymParams &sp, hipStream_t st, int num_cu) {
    if (sp.tx_raw_total < 0 || sp.syms_per_frame != 1 || sp.payload_len) return hipErrorNotSupported;
    const int nd = sp.guard ? 48 * 64 : 4096;
    const int sym_bytes = nd * sp.bps / 8;
    if ((sym_bytes & 3) || sp.payload_stride != sym_bytes || sp.out_stride_s != 5120) return hipErrorNotSupported;
    if (reinterpret_cast<uintptr_t>(sp.out) & 15) return hipErrorNotSupported; // 16-byte stores
    if (sp.n_frames <= 0) return hipSuccess;
    Tx4096Params p;
    p.bytes = sp.payload; p.n_bytes = sp.tx_raw_total; p.n_sym = sp.n_frames; p.tw = sp.tw; p.out = sp.out; p.bps = sp.bps; p.guard = sp.guard;
    const size_t lds = (size_t)(8 * 8 * 72 + 64 * 72) * sizeof(float2) + 4096 + 16 + 256 * sizeof(float2); // slabs, T, byte window + slack, point table
    {   // > 64 KB of dynamic LDS: per device, so set on every call (one process may drive several GPUs); once per batch
        hipError_t e = sp.guard ? hipFuncSetAttribute(reinterpret_cast<const void *>(k_tx4096<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                : hipFuncSetAttribute(reinterpret_cast<const void *>(k_tx4096<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    long long grid = (long long)num_cu * 2;
    { const long long cap = tuning_or_default(sp.tune).grid_cap; if (cap > 0 && cap < grid) grid = cap; }
    if (grid > p.n_sym) grid = p.n_sym;
    trace_add(sp.trace, "k_tx4096");
    if (sp.guard) hipLaunchKernelGGL(k_tx4096<true>, dim3((unsigned)grid), dim3(512), lds, st, p);
    else hipLaunchKernelGGL(k_tx4096<false>, dim3((unsigned)grid), dim3(512), lds, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// k_txframe4096: encode (src/transmitter.rs:11-58) for N = 4096 in ONE pass over HBM, the frame-level sibling of k_tx4096 and
// the R = 64 member of k_txframe_mid's scheme (kernels_mid.hip): one 512-thread workgroup per frame builds the frame's data
// symbols TWICE -- pass 0 only for the signed maximum normalize needs (transmitter.rs:184-188), pass 1 to store them divided
// by it -- instead of writing, reading back and rewriting them (k_sym<4096, M_TX> + k_tx_finish: 24 B of traffic per sample).
// One instance of the symbol builder, the pass is a uniform branch around the two epilogues.
struct TxFrame4096Params {
    const uint8_t *payload;
    long long payload_stride;
    const int32_t *payload_len;
    int payload_bytes;
    long long n_frames;
    int D;               // data symbols per frame
    const float2 *tw;    // exp(-2 pi i m / 4096)
    const float2 *header; // 10 S samples
    float header_max;
    float2 *out;
    long long out_stride; // samples
    int bps;
    int optimistic;      // as k_txframe_mid: symbols leave divided by the header maximum; a frame that exceeds it is built again
};

template <bool GUARD>
__global__ __launch_bounds__(512, 4) void k_txframe4096(TxFrame4096Params p) {
    constexpr int N = 4096, S = 5120, CP = 1024, TS = 72, SLAB = 8 * 72;
    extern __shared__ __align__(16) unsigned char smem[];
    cf *slab_all = reinterpret_cast<cf *>(smem);
    cf *T = slab_all + 8 * SLAB;
    unsigned *sbw = reinterpret_cast<unsigned *>(T + 64 * TS);          // [1024 + 4] the symbol's bytes as dwords + slack
    cf *ptab = reinterpret_cast<cf *>(sbw + 1024 + 4);                  // [256] map_point by raw bit field
    unsigned *fmax = reinterpret_cast<unsigned *>(ptab + 256);          // [1] max(0, re, im) of the frame's data symbols, float bits

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane >> 3, t = lane & 7;
    const int col = 8 * wave + s;
    cf *buf = slab_all + wave * SLAB + s * 72;
    const int wr = swz(8 * t);
    if (tid < (1 << p.bps)) ptab[tid] = map_point((unsigned)tid, p.bps);
    if (tid < 4) sbw[1024 + tid] = 0u;
    const unsigned fmask = (1u << p.bps) - 1u;
    cf w[7];
#pragma unroll
    for (int r = 1; r < 8; ++r) { const cf x = p.tw[64 * r * t]; w[r - 1] = make_float2(x.x, -x.y); }
    cf z[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const cf x = p.tw[col * (t + 8 * q)]; z[q] = make_float2(x.x, -x.y); }
    int boff[8]; // bit offset of bin 64 (t + 8 m) + col inside the symbol's stream, -1 = null, -2 = pilot
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int a = t + 8 * m, cls = carrier_class64(a, GUARD);
        boff[m] = cls == 0 ? ((GUARD ? data_classes_below64(a) : a) * 64 + col) * p.bps : (cls == 2 ? -2 : -1);
    }
    const int nd = GUARD ? 48 * 64 : N;
    const int sym_bytes = nd * p.bps / 8;   // <= 4096, a multiple of 4
    const bool aligned = ((reinterpret_cast<uintptr_t>(p.payload) | (uintptr_t)p.payload_stride) & 3) == 0;
    // The stream bytes of symbol (f, k) -- two dwords per lane -- and the frame's length are REQUESTED one symbol ahead and taken out of
    // the registers when the symbol is built (paydw_issue / paydw_settle, as k_txframe_mid): read where they are used they cost a
    // dependent round trip to HBM per symbol.  The barriers of this kernel are LDS-only (lds_barrier) so that the request stays in flight.
    struct Pre { unsigned d0, d1; int len_raw; };
    auto issue_sym = [&](bool valid, long long f, int k, Pre &pr) {
        const long long fc = valid ? f : 0;
        const uint8_t *pay = p.payload + fc * p.payload_stride;
        const long long by0 = (long long)k * sym_bytes + 4 * tid - 16, by1 = by0 + 2048;   // payload byte of this lane's two dwords
        pr.d0 = paydw_issue(pay, by0, p.payload_bytes, valid && 4 * tid < sym_bytes, aligned, p.tw);
        pr.d1 = paydw_issue(pay, by1, p.payload_bytes, valid && 4 * (tid + 512) < sym_bytes, aligned, p.tw);
        pr.len_raw = p.payload_len ? p.payload_len[fc] : p.payload_bytes;
    };
    Pre pre;
    issue_sym(blockIdx.x < p.n_frames, blockIdx.x, 0, pre);

    for (long long f = blockIdx.x; f < p.n_frames; f += gridDim.x) {
        if (tid == 0) *fmax = 0u;
        const uint8_t *pay = p.payload + f * p.payload_stride;
        cf *row = p.out + f * p.out_stride;
        // Optimistic scheme (kernels_mid.hip, k_txframe_mid): pass 0 builds every symbol once, notes its maximum and stores it
        // divided by the header maximum; pass 1 -- every symbol again, divided by the true maximum -- runs only for a frame whose data
        // exceeded the header (crafted payloads).  Without p.optimistic pass 0 only forms the maximum (the round-4 scheme, the A/B).
        const bool opt = p.optimistic != 0;
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) { // header blocks (src/transmitter.rs:22-34), divided by the frame maximum like the data
                lds_barrier();
                const float inv = 1.0f / fmaxf(p.header_max, __uint_as_float(*fmax));
                float4 *dst4 = reinterpret_cast<float4 *>(row);
                const float4 *h4 = reinterpret_cast<const float4 *>(p.header);
                // four table reads in flight per lane, then their four stores; unconditional reads from a clamped index
                for (int i0 = tid; i0 < 5 * S; i0 += 4 * 512) {
                    float4 h[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int i = i0 + 512 * j; h[j] = h4[i < 5 * S ? i : 5 * S - 1]; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = i0 + 512 * j;
                        if (i < 5 * S) dst4[i] = make_float4(h[j].x * inv, h[j].y * inv, h[j].z * inv, h[j].w * inv);
                    }
                }
                if (opt && !(__uint_as_float(*fmax) > p.header_max)) break;
            }
            for (int k = 0; k < p.D; ++k) {
                const long long sb0 = (long long)k * sym_bytes;
                Pre cur;
                if (opt && pass == 1) issue_sym(true, f, k, cur);   // the rare rebuild asks where it uses; `pre` keeps the next frame's
                else {
                    cur = pre;
                    // the next item: the next symbol, the first symbol of pass 1 (two-pass scheme), or the first of the workgroup's next frame
                    if (k + 1 < p.D) issue_sym(true, f, k + 1, pre);
                    else if (pass == 0 && !opt) issue_sym(true, f, 0, pre);
                    else issue_sym(f + gridDim.x < p.n_frames, f + gridDim.x, 0, pre);
                }
                const long long len = row_len(cur.len_raw, p.payload_bytes);
                // stream bytes of a frame: [16-byte little-endian length | payload | zeros] (src/packets/mod.rs:20-32)
                auto word = [&](unsigned raw, long long sb, bool want) -> unsigned {
                    if (!want) return 0u;
                    if (sb < 16) return sb < 8 ? (unsigned)((unsigned long long)len >> (8 * sb)) : 0u;
                    return paydw_settle(raw, pay, sb - 16, len, true, aligned);
                };
                sbw[tid] = word(cur.d0, sb0 + 4 * tid, 4 * tid < sym_bytes);
                sbw[tid + 512] = word(cur.d1, sb0 + 4 * (tid + 512), 4 * (tid + 512) < sym_bytes);
                lds_barrier(); // the byte window is complete (and, first time round, fmax / ptab are set)
                long long left = 16 + len - sb0;                     // stream bytes that belong to this symbol
                left = left < 0 ? 0 : (left < sym_bytes ? left : sym_bytes);
                const int live_bits = (int)(((unsigned)left * 8u + (unsigned)p.bps - 1u) / (unsigned)p.bps) * p.bps;
                cf v[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = tx_point<true>(sbw, ptab, boff[m], live_bits, fmask);
                bfly8<true>(v);
#pragma unroll
                for (int r = 0; r < 8; ++r) buf[wr ^ r] = v[r];
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = buf[8 * m + (t ^ m)];
#pragma unroll
                for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], w[r - 1]);
                bfly8<true>(v);
#pragma unroll
                for (int q = 0; q < 8; ++q) T[(t + 8 * q) * TS + (col ^ (t & 6))] = cmul(v[q], z[q]);
                lds_barrier();
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = T[col * TS + (t ^ (col & 6)) + 8 * m];
                bfly8<true>(v);
#pragma unroll
                for (int r = 0; r < 8; ++r) buf[wr ^ r] = v[r];
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = buf[8 * m + (t ^ m)];
#pragma unroll
                for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], w[r - 1]);
                bfly8<true>(v);
                // v[q] = N x[col + 64 (t + 8 q)]
                if (pass == 0) {
                    float mine = 0.f;   // max first, one scaling behind it (x -> x / N is monotone: the same bits), three-operand maxima
#pragma unroll
                    for (int q = 0; q < 8; ++q) mine = __builtin_fmaxf(mine, __builtin_fmaxf(v[q].x, v[q].y));
                    mine *= 1.0f / N;
#pragma unroll
                    for (int sh = 32; sh >= 1; sh >>= 1) mine = fmaxf(mine, __shfl_xor(mine, sh, 64));
                    if (lane == 0) atomicMax(fmax, __float_as_uint(mine));
                    if (!opt) {
                        lds_barrier(); // every wavefront has read its stage-B inputs out of T; the byte window is free
                        continue;
                    }
                }
                // one division, then multiplies (<= 1 ulp)
                const float sc = (1.0f / N) / (pass == 0 ? p.header_max : fmaxf(p.header_max, __uint_as_float(*fmax)));
                lds_barrier(); // every wavefront has read its stage-B inputs out of T
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    T[(t + 8 * q) * TS + (col ^ (t & 6))] = make_float2(v[q].x * sc, v[q].y * sc);
                lds_barrier();
                {   // prefix_block: out = [x[N - CP .. N), x[0 .. N)], 16 bytes per lane
                    float4 *dst4 = reinterpret_cast<float4 *>(row + (long long)(10 + k) * S);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = tid + 512 * j, n = 2 * i;
                        const float4 y = *reinterpret_cast<const float4 *>(T + (n >> 6) * TS + ((n & 63) ^ ((n >> 6) & 6)));
                        dst4[(CP >> 1) + i] = y;
                        if (j == 3) dst4[i - ((N - CP) >> 1)] = y;
                    }
                }
                lds_barrier(); // the byte window / T are reused by the next symbol
            }
        }
        lds_barrier(); // fmax is reset for the next frame
    }
}

// encode for N = 4096 in one pass.  hipErrorNotSupported => caller runs k_sym<4096, M_TX> + k_tx_finish.
hipError_t run_txframe4096(const SymParams &sp, const float2 *header, float header_max, hipStream_t st, int num_cu) {
    if (sp.tx_raw_total >= 0 || sp.syms_per_frame <= 0) return hipErrorNotSupported;
    if ((reinterpret_cast<uintptr_t>(sp.out) & 15) || (sp.out_stride_s & 1)) return hipErrorNotSupported;
    const int nd = sp.guard ? 48 * 64 : 4096;
    if ((nd * sp.bps / 8) & 3) return hipErrorNotSupported;
    if (sp.n_frames <= 0) return hipSuccess;
    TxFrame4096Params p;
    p.payload = sp.payload; p.payload_stride = sp.payload_stride; p.payload_len = sp.payload_len; p.payload_bytes = sp.payload_bytes;
    p.n_frames = sp.n_frames; p.D = sp.syms_per_frame; p.tw = sp.tw; p.header = header; p.header_max = header_max;
    p.out = sp.out; p.out_stride = sp.out_stride_s; p.bps = sp.bps;
    p.optimistic = tuning_or_default(sp.tune).no_txframe_optimistic ? 0 : 1;
    const size_t lds = (size_t)(8 * 8 * 72 + 64 * 72) * sizeof(float2) + 4096 + 16 + 256 * sizeof(float2) + 16;
    hipError_t e = sp.guard ? hipFuncSetAttribute(reinterpret_cast<const void *>(k_txframe4096<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                            : hipFuncSetAttribute(reinterpret_cast<const void *>(k_txframe4096<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    long long grid = (long long)num_cu * 2;
    { const long long cap = tuning_or_default(sp.tune).grid_cap; if (cap > 0 && cap < grid) grid = cap; } // test hook, as kernels_mid.hip
    if (grid > p.n_frames) grid = p.n_frames;
    trace_add(sp.trace, "k_txframe4096");
    if (sp.guard) hipLaunchKernelGGL(k_txframe4096<true>, dim3((unsigned)grid), dim3(512), lds, st, p);
    else hipLaunchKernelGGL(k_txframe4096<false>, dim3((unsigned)grid), dim3(512), lds, st, p);
    return hipGetLastError();
}

// Persistent grid: resident workgroups per CU from the occupancy API (per instantiation and device, cached), doubled up
// to the 8 the round-1 launcher used -- measured (OFDM_DEMOD64_WG_PER_CU sweep, 1 M frames): 3 -> 1.77 ms, 4 -> 1.70,
// 5 -> 1.81, 8 -> 1.68: a second, queued round of workgroups evens out the tail.
// The cache is keyed by the kernel's ADDRESS: every k_demod64 instantiation has the same function type, so a static inside a
// template over that type would be shared by all of them (the BURST = 16 variant holds 37 KB of LDS per workgroup, the others 11).
template <typename K> static int resident_blocks(K kernel, int block) {
    struct Entry { const void *fn; int dev, n; };
    static Entry cache[64];
    static int used = 0;
    static std::mutex mtx;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const void *fn = reinterpret_cast<const void *>(kernel);
    std::lock_guard<std::mutex> lock(mtx);
    for (int i = 0; i < used; ++i) if (cache[i].fn == fn && cache[i].dev == dev) return cache[i].n;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block, 0) != hipSuccess || n < 1) n = 4;
    n = n > 8 ? 8 : n;
    if (used < 64) cache[used++] = Entry{fn, dev, n};
    return n;
}
template <int BPS, bool GUARD, bool HK> static hipError_t launch_demod64(Fast64Params p, hipStream_t st, int num_cu, const Tuning &tu, Trace *trace) {
    p.debug = kProfile ? tu.debug_demod64 : 0;
    constexpr int region_bytes = (GUARD ? 48 : 64) * BPS;
    {   // 16-byte stores need 16-byte aligned group regions: base, frame stride and the 8-symbol region itself
        p.store_policy = tu.demod64_store_policy;
        p.wide_stores = !tu.demod64_narrow_stores && region_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0 && (p.out_stride & 15) == 0;
    }
    // burst mode (config-2 shape only: the template is instantiated once): 4 groups per step, whole-line stores
    const bool burst = BPS == 6 && GUARD && !HK && tu.demod64_burst >= 4 && p.wide_stores && p.n_groups % 4 == 0 &&
                       p.out_stride == (long long)p.groups_per_frame * region_bytes && (reinterpret_cast<uintptr_t>(p.out) & 127) == 0;
    // 16 groups per burst where the batch divides (4.6 KB of stores per wavefront step; 37 KB of LDS per workgroup still leaves four
    // resident): 1.784 -> 1.761 ms per 1 M frames against bursts of 4 on a box of the slow population; Tuning::demod64_burst caps it (A/B)
    const int burst_cap = tu.demod64_burst;
    const int bl = !burst ? 1 : (burst_cap >= 16 && p.n_groups % 16 == 0) ? 16 : (burst_cap >= 8 && p.n_groups % 8 == 0) ? 8 : 4;
    auto kernel = !burst ? k_demod64<BPS, GUARD, HK, 1> : bl == 16 ? k_demod64<6, true, false, 16> : bl == 8 ? k_demod64<6, true, false, 8> : k_demod64<6, true, false, 4>;
    const long long units = burst ? p.n_groups / bl : p.n_groups;
    long long waves = (units + 3) / 4 * 4;
    const int knob = tu.demod64_wg_per_cu; // tuning knob
    long long cap = (long long)num_cu * (knob > 0 ? knob : (resident_blocks(kernel, 256) >= 4 ? 8 : 2 * resident_blocks(kernel, 256))) * 4;
    if (tu.grid_cap > 0 && tu.grid_cap * 4 < cap) cap = tu.grid_cap * 4;
    if (waves > cap) waves = cap;
    const int grid = (int)(waves / 4);
    trace_add(trace, !burst ? "k_demod64" : bl == 16 ? "k_demod64<burst16>" : bl == 8 ? "k_demod64<burst8>" : "k_demod64<burst4>");
    p.stride_groups = (long long)grid * 4;
    const int gpf = p.groups_per_frame;
    p.f0 = 0; p.k0 = 0;
    p.blk_df = 4 / gpf; p.blk_dk = 4 % gpf;    // a block advances the group index by 4
    p.wave_df = 0; p.wave_dk = 1;              // a wave by 1 (normalised in the kernel)
    p.step_df = p.stride_groups / gpf; p.step_dk = (int)(p.stride_groups % gpf);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, st, p);
    return hipGetLastError();
}
template <int BPS> static hipError_t launch_bps(const Fast64Params &p, bool guard, hipStream_t st, int num_cu, const Tuning &tu, Trace *trace) {
    const bool hk = p.hk != nullptr;
    if (guard && hk) return launch_demod64<BPS, true, true>(p, st, num_cu, tu, trace);
    if (guard) return launch_demod64<BPS, true, false>(p, st, num_cu, tu, trace);
    if (hk) return launch_demod64<BPS, false, true>(p, st, num_cu, tu, trace);
    return launch_demod64<BPS, false, false>(p, st, num_cu, tu, trace);
}

// Returns hipErrorNotSupported when the request is outside the fast path's envelope (caller falls back to k_sym).
hipError_t run_demod64_fast(const SymParams &sp, hipStream_t st, int num_cu) {
    if (sp.offset || sp.f_delta || sp.nsym_frame || sp.soft) return hipErrorNotSupported;
    if (sp.syms_per_frame <= 0 || (sp.syms_per_frame & 7) != 0) return hipErrorNotSupported;
    if (sp.hk && sp.hk_stride != 0) return hipErrorNotSupported;
    if ((long long)(sp.first_symbol + sp.syms_per_frame) * 80 > sp.frame_len) return hipErrorNotSupported; // no tail padding
    if ((reinterpret_cast<uintptr_t>(sp.out_bytes) & 3) || (sp.out_stride & 3)) return hipErrorNotSupported;
    if (reinterpret_cast<uintptr_t>(sp.in) & 7) return hipErrorNotSupported;
    Fast64Params p;
    p.in = sp.in; p.frame_stride = sp.frame_stride; p.first_symbol = sp.first_symbol;
    p.hk = sp.hk; p.tw = sp.tw; p.out = sp.out_bytes; p.out_stride = sp.out_stride;
    p.groups_per_frame = sp.syms_per_frame / 8;
    p.n_groups = sp.n_frames * (long long)p.groups_per_frame;
    if (p.n_groups <= 0) return hipSuccess;
    const Tuning &tu = tuning_or_default(sp.tune);
    switch (sp.bps) {
    case 1: return launch_bps<1>(p, sp.guard != 0, st, num_cu, tu, sp.trace);
    case 2: return launch_bps<2>(p, sp.guard != 0, st, num_cu, tu, sp.trace);
    case 4: return launch_bps<4>(p, sp.guard != 0, st, num_cu, tu, sp.trace);
    case 6: return launch_bps<6>(p, sp.guard != 0, st, num_cu, tu, sp.trace);
    case 8: return launch_bps<8>(p, sp.guard != 0, st, num_cu, tu, sp.trace);
    default: return hipErrorNotSupported;
    }
}

} // namespace ofdm
