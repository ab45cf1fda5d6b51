// ofdm_ctx.hpp -- the context behind `ofdm_ctx *` and the helpers every translation unit of the C ABI shares (internal to
// libofdm_hip.so): ofdm_abi.hip (device-buffer entry points) and ofdm_host_path.hip (host-buffer pipelines, long captures).
#pragma once
#include "../../include/ofdm_hip.h"
#include "kernels.hpp"

#include <cstddef>

struct Workspace {
    void *ptr = nullptr;
    size_t cap = 0;
};

struct HostPipe; // pinned staging, copy streams and slot buffers of the host-buffer entry points (ofdm_host_path.hip)

struct ofdm_ctx {
    ofdm_params prm;
    int device = 0;
    int num_cu = 256;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int last_hip = 0;
    // constant device tables
    float2 *d_tw = nullptr;       // exp(-2 pi i m / N)
    float2 *d_inv_trn = nullptr;  // 1 / training[k]
    float2 *d_header = nullptr;   // 10 * S un-normalised header samples
    float header_max = 0.f;
    ofdm::Tuning tune;                  // ofdm_set_tuning: per-context A/B switches and grid shapes (no environment variable is read)
    ofdm::Trace trace;                  // ofdm_last_dispatch: the kernels the last entry point launched
    ofdm::ScStats sc_stats;             // list counters of the last Schmidl-Cox search (ofdm_get_tuning "stat_sc_*")
    int32_t *d_stats = nullptr;         // [2] their home on the device (owned by the context)
    // workspaces (grown on demand, never inside a captured region)
    Workspace ws[10];
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    HostPipe *pipe = nullptr;     // created by the first host-buffer call, freed by ofdm_destroy

    int S() const { return prm.n_fft + prm.cp_len; }
    int carriers() const { return prm.guard_bands ? 48 * (prm.n_fft / 64) : prm.n_fft; }
    int bytes_per_symbol() const { return carriers() * prm.modulation / 8; }
};

// Scoped device selection: every entry point that touches HIP runs on its context's device and leaves the calling
// thread's current device as it found it (one thread may hold contexts on several GPUs; torch shares the thread's device).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

#define HIP_TRY(ctx, expr)                                   \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) { (ctx)->last_hip = (int)_e; return OFDM_ERR_HIP; } \
    } while (0)

inline int ws_get(ofdm_ctx *c, int slot, size_t bytes, void **out) {
    Workspace &w = c->ws[slot];
    if (bytes > w.cap) {
        if (w.ptr) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(w.ptr)); w.ptr = nullptr; w.cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&w.ptr, want);
        if (e != hipSuccess) { c->last_hip = (int)e; w.ptr = nullptr; return OFDM_ERR_NOMEM; }
        w.cap = want;
    }
    *out = w.ptr;
    return OFDM_OK;
}


// internal cross-file helpers (C linkage only because their definitions sit inside the extern "C" blocks; not in ofdm_hip.h)
extern "C" {
__attribute__((visibility("hidden"))) void ofdm_host_pipe_destroy(ofdm_ctx *c); // ofdm_host_path.hip; the context's device is current
// Schmidl-Cox over a batch (the dispatcher behind ofdm_sc_correlate_batch and step 1 of ofdm_rx_decode_batch; ofdm_abi.hip)
__attribute__((visibility("hidden"))) int ofdm_abi_sc_run(ofdm_ctx *c, const float2 *in, int64_t n_frames, int64_t frame_stride, int64_t frame_len, int64_t n_lags,
                    int32_t *d_hat, double *f_delta, float *metric);
// ofdm_rx_decode_batch for ONE capture whose timing is already known (peak lag d_hat, arg P / L and metric at it): steps 2-5 of the
// chain without its own search (ofdm_rx_decode_long with lag_lo > 0: the search must not look at lags in front of lag_lo)
__attribute__((visibility("hidden"))) int ofdm_abi_rx_decode_known(ofdm_ctx *c, const ofdm_fc32 *in, int64_t frame_len, int32_t d_hat, double f_delta, float metric,
                    int32_t max_symbols, uint8_t *out, int64_t out_stride, int32_t *out_len, int32_t *status, int32_t *offset, double *f_delta_out,
                    float *metric_out);
}
