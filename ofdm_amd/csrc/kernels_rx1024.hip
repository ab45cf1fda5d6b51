// kernels_rx1024.hip -- k_rxframe1024: the per-frame receive body for N = 1024 after timing (BASELINE config 4).
//
// One 128-thread workgroup per frame, persistent over the batch:
//   estimate_channel (src/receiver.rs:212-229) on the 5 training blocks -- derotated, summed in the time domain and transformed
//   ONCE (the transform is linear) -- then every live data symbol: CFO derotation (receiver.rs:44-50), CP strip + FFT1024
//   (receiver.rs:99-104), equalise (receiver.rs:68-70), mean angle of the 64 pilots (receiver.rs:106-145), hard demap + LSB-first
//   packing (receiver.rs:147-190), and -- when the frame's packed bytes fit the LDS image -- the finish: length header, truncate,
//   Hamming(7,4) decode (receiver.rs:85-95) straight from LDS to the caller's rows (no raw-byte round trip, no k_rx_finish launch).
//
// What changed in round 3 (the round-2 kernel waited for its loads 51 % of the time at 8 wavefronts per CU):
//   * samples arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) into a ring of three 9 KB slots; the DMA of
//     item j + 2 is issued as soon as item j has landed, so two symbols (16 KB) per workgroup are in flight at all times at no
//     cost in registers, and the ring runs ACROSS frames (the next frame's offset / CFO / symbol count are read while the
//     current frame's last symbols are transformed);
//   * the slot an item landed in is reused as its 16 x 64 transpose buffer, and every wavefront runs its FFT64 transposes in
//     place in its own 8 rows of it: 32 KB of LDS per workgroup instead of 44 KB -> 5 workgroups (10 wavefronts) per CU;
//   * stage-A twiddles live in registers again (the prefetch registers are gone).
// A frame whose start is an odd sample (8-byte aligned) is staged from the even sample below it and read with a one-sample shift;
// items that touch the end of the capture (pad_chunk, receiver.rs:203-210) take a synchronous
// load-and-store path into the same slot image.
//
// FFT1024 = 16 x 64:  X[c + 16 d] = sum_b W64^(b d) * [ W1024^(b c) * sum_a x[64 a + b] W16^(a c) ],  c < 16, d < 64
//   stage A  lane pair (b, u): an 8-point butterfly over a = u + 2 m, then the radix-2 step across the pair (one DPP swap per value);
//   twiddle, transpose through LDS ([c][b]);  stage B  the FFT64 over b for row c in the k_demod64 layout (wave-local).
// Roofline: HBM -- the training and data symbols after their prefixes (9 x 8 KB per 4-symbol frame) + the decoded bytes.
#include "device_common.hpp"
#include "kernels.hpp"

extern "C" __device__ float __ocml_atan2pi_f32(float, float); // atan2(y, x) / pi (ROCm device library)

namespace ofdm {

namespace {

template <int CTRL> __device__ __forceinline__ float dpp_x(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}

struct RxFrame1024Params {
    const float2 *in;
    long long n_frames, frame_stride, frame_len;
    const int32_t *offset;
    const double *f_delta;
    const int32_t *nsym;
    const float2 *tw;            // exp(-2 pi i m / 1024)
    const float2 *inv_training;  // 1 / training[k]
    unsigned char *out;          // raw decoded bytes (unfused mode), nsym * bytes_per_symbol per frame
    long long out_stride;
    float2 *hk;                  // optional: channel estimate per frame (1024 bins)
    int aligned;                 // `in` has its natural 8-byte alignment: LDS-DMA staging
    // fused finish (optional): final payload rows, 4-byte aligned; ecc = 1: Hamming(7,4)
    unsigned char *final_out;
    long long final_stride;
    int32_t *final_len;
    int ecc;
};

constexpr int RX_RAW_DW = 896;   // LDS image of a frame's packed bytes (3.5 KB): fused finish for frames up to that size

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct Cur {          // a position in this workgroup's stream of (frame, step) items; step -5 .. -1 = training blocks, 0 .. ns-1 = data
    long long f;      // n_frames = end of the stream
    long long rel;    // offset[f]
    int ns, step;
    double turns;
};

} // namespace

template <int BPS, bool GUARD>
__global__ __launch_bounds__(128, 3) void k_rxframe1024(RxFrame1024Params p) {
    constexpr int N = 1024, S = 1280, CP = 256, TS = 72, SLOT = 16 * TS;  // 1152 samples per ring slot (1026 staged)
    constexpr int ND = GUARD ? 48 * 16 : N;
    constexpr int IMG_DW = ND * BPS / 32;            // packed bytes of one symbol, in dwords (<= 256)
    constexpr int nbytes = ND * BPS / 8;
    constexpr int RING = 2;                          // slots: the item being transformed + the next one in flight (a third slot -- two in
                                                     // flight -- costs a workgroup per CU: measured slower)
    __shared__ __align__(16) cf ring[RING * SLOT];
    __shared__ unsigned raw[RX_RAW_DW];              // fused: the frame's packed bytes; unfused: one symbol's image at [0, IMG_DW)
    __shared__ float red[2];
    __shared__ cf w16tab[8 * 2];
    __shared__ cf w64tab[7 * 8];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = tid >> 1, u = tid & 1;     // stage A roles
    const int s = lane >> 3, t = lane & 7;   // stage B roles
    const int row = 8 * wave + s;
    const int wr = swz(8 * t);
    const bool fused = p.final_out != nullptr;

    if (tid < 16) w16tab[tid] = (tid & 1) ? p.tw[64 * (tid >> 1)] : make_float2(1.f, 0.f);
    if (tid < 56) w64tab[tid] = p.tw[16 * (tid / 8 + 1) * (tid & 7)];
    cf z[8];                                 // W1024^(b c), c = c' + 8 u
#pragma unroll
    for (int c = 0; c < 8; ++c) z[c] = p.tw[b * (c + 8 * u)];
    for (int i = tid; i < RX_RAW_DW; i += 128) raw[i] = 0u;
    __syncthreads();
    int bo8[8];                              // bit offset of bin row + 16 d (d = t + 8 q) in the symbol's image, -1 = not a data bin
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int d = t + 8 * q;
        bo8[q] = carrier_class64(d, GUARD) == 0 ? ((GUARD ? data_classes_below64(d) : d) * 16 + row) * BPS : -1;
    }

    // Per-frame scalars: the three loads of a frame go out together and are waited for once (as wave-uniform scalar values each
    // would be moved to SGPRs, and waited for, right where it is issued: three dependent round trips to HBM per frame).
    int vzero = 0;
    asm volatile("" : "+v"(vzero));   // a VGPR zero the compiler cannot fold: keeps the loads per-lane
    auto open_frame = [&](Cur &c, long long f) {
        for (;;) {
            c.f = f; c.step = -5; c.rel = 0; c.ns = 0; c.turns = 0.0;
            if (f >= p.n_frames) return;
            const long long fi = f + vzero;
            const int ns_v = p.nsym[fi];
            const int off_v = p.offset ? p.offset[fi] : 0;
            const double fd_v = p.f_delta ? p.f_delta[fi] : 0.0;
            c.ns = __builtin_amdgcn_readfirstlane(ns_v);
            c.rel = __builtin_amdgcn_readfirstlane(off_v);
            c.turns = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(fd_v)), __builtin_amdgcn_readfirstlane(__double2loint(fd_v))) * 0.15915494309189533577;
            if (c.ns > 0) return;
            if (fused && tid == 0) p.final_len[f] = 0;   // no sync / short capture: nothing decoded (every skipped frame is visited exactly once)
            f += gridDim.x;
        }
    };
    // The consume cursor never loads: when it leaves a frame the issue cursor (one item ahead, and a frame has at least six) is
    // already inside the next live one.
    Cur ic, cc;
    auto advance_issue = [&]() { if (++ic.step >= ic.ns) open_frame(ic, ic.f + gridDim.x); };
    auto advance_consume = [&]() { if (++cc.step >= cc.ns) { cc = ic; cc.step = -5; } };
    // Stage item c into ring slot `slot`: x[j] = frame[rel + chunk S + CP + j] lands at slot[j + ((rel + ...) & 1)].
    // Returns true when the LDS-DMA path was taken (its pieces are then outstanding on this wavefront's VM counter).
    auto issue = [&](const Cur &c, int slot) -> bool {
        const long long r0 = c.rel + (long long)(10 + c.step) * S + CP;  // frame-relative index of the symbol's first FFT sample
        const long long a = r0 & ~1LL;
        cf *dst = ring + slot * SLOT;
        if (p.aligned && a + 1026 <= p.frame_len) {
            const char *sb = reinterpret_cast<const char *>(p.in + c.f * p.frame_stride + a);
            const unsigned l0 = lds_addr(dst);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned piece = (unsigned)(wave + 2 * j);
                glds16(sb, piece * 1024u + (unsigned)lane * 16u, l0 + piece * 1024u);
            }
            if (wave == 0) { if (lane == 0) glds16(sb, 8192u, l0 + 8192u); } // samples 1024, 1025: the one-sample shift's tail
            return true;
        }
        const int sh = (int)(r0 & 1);
        const cf *src = p.in + c.f * p.frame_stride + r0;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int i = 64 * u + 128 * m + b;
            dst[sh + i] = (r0 + i) < p.frame_len ? src[i] : make_float2(0.f, 0.f); // zero past the capture: pad_chunk
        }
        return false;
    };

    open_frame(ic, blockIdx.x);       // issue and consume cursors start at the workgroup's first live frame
    cc = ic;
    bool dma_cur = false, dma_next = false;   // how the item under the consume cursor / the one after it were staged
    bool have_next = false;
    long long k = 0;                  // items consumed so far: item k lives in slot k % RING
    if (ic.f < p.n_frames) {
        dma_cur = issue(ic, 0);
        advance_issue();
        if (RING == 3 && ic.f < p.n_frames) { dma_next = issue(ic, 1); have_next = true; advance_issue(); }
    }
    unsigned *pending = nullptr;      // unfused mode: where the symbol image currently in LDS belongs
    cf g[8];                          // first the time-domain sum of the derotated training blocks, then 1 / H
#pragma unroll
    for (int q = 0; q < 8; ++q) g[q] = make_float2(0.f, 0.f);

    while (cc.f < p.n_frames) {
        // ---- item k has landed: every piece this wavefront issued for it, then everyone's
        if (dma_cur) {
            if (RING == 3 && have_next && dma_next) { if (wave == 0) wait_vm<5>(); else wait_vm<4>(); }   // the next item's pieces stay in flight
            else wait_vm<0>();
        }
        lds_barrier();                // B1: item k's samples are visible; everyone is done with item k - 1 (its slot, T, red)
        const int slot = (int)(k % RING);
        // ---- the item two ahead goes into the slot item k - 1 has just left
        bool dma_issued = false, issued = false;
        if (ic.f < p.n_frames) { dma_issued = issue(ic, (int)((k + RING - 1) % RING)); issued = true; advance_issue(); }
        if (!fused && pending) {      // unfused: the previous symbol's image leaves for HBM, the LDS image is cleared
            for (int i = tid; i < IMG_DW; i += 128) { pending[i] = raw[i]; raw[i] = 0u; }
            pending = nullptr;
        }
        const long long f = cc.f;
        const int step = cc.step, ns = cc.ns;
        const long long r0 = cc.rel + (long long)(10 + step) * S + CP;
        const int sh = (int)(r0 & 1);
        cf *slotp = ring + slot * SLOT;
        cf v[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = slotp[sh + 64 * u + 128 * m + b];
        const double turns = cc.turns;
        if (p.f_delta && step >= 0) { // CFO derotation, sample ids count from the trimmed start (receiver.rs:44-50); phase reduced in f64
            cf ph = cfo_phasor(turns, (long long)(10 + step) * S + CP + 64 * u + b);
            const cf st = cfo_phasor(turns, 128);
#pragma unroll
            for (int m = 0; m < 8; ++m) { v[m] = cmul(v[m], ph); ph = cmul(ph, st); }
        }
        // rotate the bookkeeping now: everything below refers to (f, step, slotp) only
        const bool last_of_frame = step == ns - 1;
        advance_consume();
        if (RING == 3) { dma_cur = dma_next; dma_next = dma_issued; have_next = issued; }
        else dma_cur = dma_issued;
        ++k;
        if (step < 0) {
            // estimate_channel (receiver.rs:212-229) averages the spectra of the 5 training blocks; the transform is linear, so
            // the derotated blocks are summed in the time domain and transformed ONCE.  The derotation of sample i of block b,
            // exp(-j phi ((5 + b) S + CP + i)), splits into the block's own factor exp(-j phi S b) -- one uniform phasor per block --
            // and a per-sample factor that is applied once, to the sum.
            if (p.f_delta) {
                const cf rb = cfo_phasor(turns, (long long)S * (step + 5));
#pragma unroll
                for (int m = 0; m < 8; ++m) g[m] = cadd(g[m], cmul(v[m], rb));
            } else {
#pragma unroll
                for (int m = 0; m < 8; ++m) g[m] = cadd(g[m], v[m]);
            }
            if (step < -1) continue; // workgroup-uniform
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = g[m];
            if (p.f_delta) {
                cf ph = cfo_phasor(turns, (long long)5 * S + CP + 64 * u + b);
                const cf st = cfo_phasor(turns, 128);
#pragma unroll
                for (int m = 0; m < 8; ++m) { v[m] = cmul(v[m], ph); ph = cmul(ph, st); }
            }
        }
        lds_barrier();                // B2: every thread has taken its samples out of the slot: it becomes the transpose buffer
        // ---- stage A: FFT16 over a = u + 2 m  (8-point butterfly, then radix 2 across the lane pair)
        bfly8<false>(v);
        cf *T = slotp;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const cf e = cmul(v[c], w16tab[2 * c + u]);
            const cf o = make_float2(dpp_x<0xB1>(e.x), dpp_x<0xB1>(e.y));   // the partner's value (lane ^ 1)
            const cf y = u ? make_float2(o.x - e.x, o.y - e.y) : make_float2(e.x + o.x, e.y + o.y);
            T[(c + 8 * u) * TS + b] = cmul(y, z[c]);                          // Y_b[c' + 8 u] * W1024^(b c)
        }
        lds_barrier();                // B3: T complete
        // ---- stage B: FFT64 over b for row c; the wavefront's own 8 rows of T double as its transpose slab
        cf *buf = T + row * TS;
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = buf[t + 8 * m];
        bfly8<false>(v);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the row reads above precede the in-place writes below (in-order LDS pipe)
#pragma unroll
        for (int r = 0; r < 8; ++r) buf[wr ^ r] = v[r];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = buf[8 * m + (t ^ m)];
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], w64tab[(r - 1) * 8 + t]);
        bfly8<false>(v);
        // v[q] = X[row + 16 (t + 8 q)]
        if (step < 0) { // step == -1: H = FFT(sum of the training blocks) / 5 / training
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int bin = row + 16 * (t + 8 * q);
                cf h = cmul(v[q], p.inv_training[bin]);
                h = make_float2(h.x * 0.2f, h.y * 0.2f);
                if (p.hk) p.hk[f * N + bin] = h;
                const float rn = __builtin_amdgcn_rcpf(h.x * h.x + h.y * h.y);
                g[q] = make_float2(h.x * rn, -h.y * rn); // 1 / H
            }
            continue;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = cmul(v[q], g[q]); // equalise (receiver.rs:68-70)
        cf rot = make_float2(1.f, 0.f);
        if (GUARD) { // decode_block (receiver.rs:106-145): mean angle of the 4 x 16 pilots, rotate by -phase
            cf pv = make_float2(1.f, 0.f);
            pv = (t == 6) ? v[0] : pv;
            pv = (t == 1) ? v[3] : pv;
            pv = (t == 7) ? v[4] : pv;
            pv = (t == 2) ? v[7] : pv;
            float a = __ocml_atan2pi_f32(pv.y, pv.x);
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) a += __shfl_xor(a, sft, 64);
            if (lane == 0) red[wave] = a;
            lds_barrier();            // B4
            const float trn = (red[0] + red[1]) * (0.5f / 64.0f); // mean of the 64 pilot angles, in turns
            rot = make_float2(__builtin_amdgcn_cosf(trn), -__builtin_amdgcn_sinf(trn)); // applied inside the demapper
        }
        unsigned *img = fused ? raw + step * IMG_DW : raw;
#pragma unroll
        for (int q = 0; q < 8; ++q) { // demodulate + LSB-first packing: OR every field into the image
            const int bo = bo8[q];
            if (bo >= 0) {
                const unsigned idx = GUARD ? demap_point_rot(v[q], rot, BPS) : demap_point(v[q], BPS);
                const int wd = bo >> 5, shf = bo & 31;
                atomicOr(&img[wd], idx << shf);
                if (BPS > 1 && (32 % BPS) != 0) {
                    if (shf + BPS > 32) atomicOr(&img[wd + 1], idx >> (32 - shf));
                }
            }
        }
        if (!fused) {
            pending = reinterpret_cast<unsigned *>(p.out + f * p.out_stride + (long long)step * nbytes);
            if (last_of_frame) {
                lds_barrier();
                for (int i = tid; i < IMG_DW; i += 128) { pending[i] = raw[i]; raw[i] = 0u; }
                pending = nullptr;
#pragma unroll
                for (int q = 0; q < 8; ++q) g[q] = make_float2(0.f, 0.f);
            }
            continue;
        }
        if (!last_of_frame) continue;
        // ---- fused finish (receiver.rs:85-95): bincode fixint little-endian u128 length (src/packets/mod.rs:20-32), Vec::truncate,
        //      then the payload -- Hamming(7,4)-decoded if the outer code is on -- from the LDS image to its final row
        lds_barrier();                // the frame's image is complete
        {
            const unsigned long long lo = (unsigned long long)raw[0] | ((unsigned long long)raw[1] << 32);
            const unsigned long long hi = (unsigned long long)raw[2] | ((unsigned long long)raw[3] << 32);
            const int body = ns * nbytes - 16;
            const int keep = (hi == 0 && lo < (unsigned long long)body) ? (int)lo : body;
            const unsigned char *srcb = reinterpret_cast<const unsigned char *>(raw) + 16;
            unsigned char *dstb = p.final_out + f * p.final_stride;
            if (!p.ecc) {
                const int nd = keep >> 2;
                for (int i = tid; i < nd; i += 128) reinterpret_cast<unsigned *>(dstb)[i] = raw[4 + i];
                for (int i = (nd << 2) + tid; i < keep; i += 128) dstb[i] = srcb[i];
                if (tid == 0) p.final_len[f] = keep;
            } else {
                unsigned fixed = 0;
                const int blocks = keep / 7;
                for (int bl = tid; bl < blocks; bl += 128) {
                    unsigned char o4[4];
                    ham_decode_block(srcb + bl * 7, o4, fixed);
                    reinterpret_cast<unsigned *>(dstb)[bl] = (unsigned)o4[0] | ((unsigned)o4[1] << 8) | ((unsigned)o4[2] << 16) | ((unsigned)o4[3] << 24);
                }
                if (tid == 0) p.final_len[f] = blocks * 4;
            }
        }
        lds_barrier();                // everyone has read the image
        for (int i = tid; i < ns * IMG_DW; i += 128) raw[i] = 0u;   // clear it for the next frame (visible after the next B1)
#pragma unroll
        for (int q = 0; q < 8; ++q) g[q] = make_float2(0.f, 0.f);
    }
}

// Fused channel estimate + demod [+ finish] for N = 1024 frames.  hipErrorNotSupported => caller uses run_chest + run_demod.
// final_out / final_stride / final_len / ecc (optional, rows 4-byte aligned): the kernel also does the finish
// (length header, truncate, Hamming decode) when a frame's packed bytes fit its LDS image; *fused_out says whether it did.
// CONTRACT of the fused finish: a frame that must not be decoded has nsym_frame[f] == 0 -- that, not a status array, is what
// zeroes its final_len (k_rx_prepare / k_rx_prepare_ref write nsym = 0 whenever status != 0).
hipError_t run_rxframe1024(const SymParams &sp, float2 *hk_out, hipStream_t st, int num_cu, unsigned char *final_out,
                           long long final_stride, int32_t *final_len, int ecc, bool *fused_out) {
    if (fused_out) *fused_out = false;
    if (!sp.nsym_frame || sp.soft) return hipErrorNotSupported;
    const int nd = sp.guard ? 48 * 16 : 1024;
    const int nbytes = nd * sp.bps / 8;
    if (nbytes % 4 != 0 || (reinterpret_cast<uintptr_t>(sp.out_bytes) & 3) || (sp.out_stride & 3)) return hipErrorNotSupported;
    if (sp.n_frames <= 0) return hipSuccess;
    RxFrame1024Params p;
    p.in = sp.in; p.n_frames = sp.n_frames; p.frame_stride = sp.frame_stride; p.frame_len = sp.frame_len;
    p.offset = sp.offset; p.f_delta = sp.f_delta; p.nsym = sp.nsym_frame; p.tw = sp.tw; p.inv_training = sp.inv_training;
    p.out = sp.out_bytes; p.out_stride = sp.out_stride; p.hk = hk_out;
    p.aligned = (reinterpret_cast<uintptr_t>(sp.in) & 7) == 0;   // LDS-DMA takes any 4-byte aligned source (tools/lab/glds_align.hip); rows of odd stride too
    p.final_out = nullptr; p.final_stride = 0; p.final_len = nullptr; p.ecc = ecc;
    // the frame's packed bytes (at most max_symbols symbols: out_stride of the raw rows) must fit the LDS image
    const bool fuse = final_out && final_len && (reinterpret_cast<uintptr_t>(final_out) & 3) == 0 && (final_stride & 3) == 0 &&
                      sp.out_stride <= (long long)RX_RAW_DW * 4 && !tuning_or_default(sp.tune).no_rx1024_finish;
    if (fuse) { p.final_out = final_out; p.final_stride = final_stride; p.final_len = final_len; }
    if (fused_out) *fused_out = fuse;
    long long grid = (long long)num_cu * 6;   // 22.6 KB of LDS and 167 VGPRs per 2-wavefront workgroup: six (three wavefronts per SIMD) per CU
    { const long long cap = tuning_or_default(sp.tune).grid_cap; if (cap > 0 && cap < grid) grid = cap; }
    if (grid > p.n_frames) grid = p.n_frames;
    trace_add(sp.trace, fuse ? "k_rxframe1024<finish>" : "k_rxframe1024");
#define OFDM_LAUNCH_RX1024(B, G) { hipLaunchKernelGGL((k_rxframe1024<B, G>), dim3((unsigned)grid), dim3(128), 0, st, p); return hipGetLastError(); }
    switch (sp.bps * 2 + (sp.guard ? 1 : 0)) {
    case 2: OFDM_LAUNCH_RX1024(1, false) case 3: OFDM_LAUNCH_RX1024(1, true)
    case 4: OFDM_LAUNCH_RX1024(2, false) case 5: OFDM_LAUNCH_RX1024(2, true)
    case 8: OFDM_LAUNCH_RX1024(4, false) case 9: OFDM_LAUNCH_RX1024(4, true)
    case 12: OFDM_LAUNCH_RX1024(6, false) case 13: OFDM_LAUNCH_RX1024(6, true)
    case 16: OFDM_LAUNCH_RX1024(8, false) case 17: OFDM_LAUNCH_RX1024(8, true)
    }
#undef OFDM_LAUNCH_RX1024
    return hipErrorNotSupported;
}

} // namespace ofdm
