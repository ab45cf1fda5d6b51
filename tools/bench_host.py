"""Host-buffer and long-capture measurements (bench.py, rank 0 of a single-GPU run).

  h2d_inclusive   the drop-in's own operating point: the reference's decode owns a HOST Vec (src/receiver.rs:9-13), so a host that
                  cannot keep captures resident pays PCIe.  Pinned host batches go through ofdm_rx_demod_host (config 2) and
                  ofdm_rx_decode_host (config 3): H2D(k + 1) || kernels(k) || D2H(k - 1); reported next to the box's own pinned
                  H2D rate measured in the same run, plus the latency of ONE 2080-sample decode call.  NEVER the headline `value`.
  long_capture    ONE 2 000 000-sample buffer per decode (examples/jetson_rx.rs:15-17,48-49,84-86) through ofdm_rx_decode_long
                  (device-resident and from pinned host memory), against the same capture as a single frame of the batch path.
"""
import math
import time

import numpy as np


def _wall(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) * 1e3 / reps


def h2d_inclusive(api, torch, device, frames2=131072, frames3=65536):
    from tools import bench_cfg3

    out = {"note": "inputs start in pinned HOST memory; times are wall clock around the synchronous host-buffer entry points"}
    ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True, device=device)
    # --- config 2 shape: frames of 16 symbols, RX demod only
    syms = 16
    g = torch.Generator(device=ctx.device); g.manual_seed(77)
    pay = torch.randint(0, 256, (frames2 * syms * ctx.bytes_per_symbol,), dtype=torch.uint8, device=ctx.device, generator=g)
    x_dev = ctx.tx_symbols(pay).view(frames2, syms * ctx.S)
    x = api.pinned_empty((frames2, syms * ctx.S), np.complex64)
    x[:] = x_dev.cpu().numpy()
    # the box's own pinned H2D / D2H rates (one large copy each, HIP events on the context's stream)
    scratch = torch.empty_like(x_dev)
    import ctypes as C

    def h2d():
        ctx._ck(ctx.lib.ofdm_memcpy_h2d(ctx.h, C.c_void_p(scratch.data_ptr()), C.c_void_p(x.ctypes.data), x.nbytes), "h2d")

    h2d(); ctx.synchronize(); ctx.timer_start()
    for _ in range(3):
        h2d()
    h2d_ms = ctx.timer_stop_ms() / 3
    ceiling = x.nbytes / h2d_ms / 1e6  # GB/s
    out["pinned_h2d_ceiling_gbs"] = ceiling
    want = ctx.rx_demod(x_dev, syms).cpu().numpy()
    res = api.pinned_empty(want.shape, np.uint8)
    ms = _wall(lambda: ctx.demod_host(x, syms, out=res), 3)
    out["cfg2_rx_demod_host"] = {"frames": frames2, "ms": ms, "msamples_per_s": frames2 * syms * ctx.S / ms / 1e3,
                                 "h2d_gbs": x.nbytes / ms / 1e6, "frac_of_pinned_h2d_ceiling": x.nbytes / ms / 1e6 / ceiling,
                                 "bytes_equal_resident_path": bool(np.array_equal(res, want)), "dispatch": ctx.last_dispatch()}
    xp = np.array(x)  # pageable copy: staged through the library's pinned bounce slots by the calling thread
    ms_p = _wall(lambda: ctx.demod_host(xp, syms), 2)
    out["cfg2_rx_demod_host_pageable"] = {"ms": ms_p, "msamples_per_s": frames2 * syms * ctx.S / ms_p / 1e3,
                                          "frac_of_pinned_h2d_ceiling": x.nbytes / ms_p / 1e6 / ceiling}
    del x_dev, scratch, pay
    # --- config 3 shape: full chain
    caps_dev, payload = bench_cfg3.synth(api, torch, ctx, frames3, seed=11)
    D = ctx.data_symbols(bench_cfg3.NBYTES)
    want3 = {k: v.cpu().numpy() for k, v in ctx.decode_batch(caps_dev, max_symbols=D).items()}
    caps = api.pinned_empty(tuple(caps_dev.shape), np.complex64)
    caps[:] = caps_dev.cpu().numpy()
    got = ctx.decode_host(caps, max_symbols=D)
    same = all(np.array_equal(got[k], want3[k]) for k in ("status", "len", "offset", "f_delta")) and \
        all(bytes(got["bytes"][f, : got["len"][f]]) == bytes(want3["bytes"][f, : want3["len"][f]]) for f in range(0, frames3, 97))
    ms3 = _wall(lambda: ctx.decode_host(caps, max_symbols=D, out=got), 3)
    out["cfg3_rx_decode_host"] = {"frames": frames3, "ms": ms3, "msamples_per_s": frames3 * caps.shape[1] / ms3 / 1e3,
                                  "h2d_gbs": caps.nbytes / ms3 / 1e6, "frac_of_pinned_h2d_ceiling": caps.nbytes / ms3 / 1e6 / ceiling,
                                  "results_equal_resident_path": bool(same), "dispatch": ctx.last_dispatch()}
    # --- TX: payload in, frames out (D2H-bound)
    pay_h = payload[:frames3].cpu().numpy()
    frames_out = api.pinned_empty((frames3, ctx.frame_samples(bench_cfg3.NBYTES)), np.complex64)
    mst = _wall(lambda: ctx.encode_host(pay_h, out=frames_out), 3)
    out["cfg3_tx_encode_host"] = {"frames": frames3, "ms": mst, "msamples_per_s": frames_out.size / mst / 1e3,
                                  "d2h_gbs": frames_out.nbytes / mst / 1e6}
    # --- latency of ONE frame per call, context reused (the reference's examples call decode! once per frame)
    one = np.array(caps[5, :2144])
    lat = []
    for _ in range(300):
        t0 = time.perf_counter()
        r = ctx.decode_long_host(one, D)
        lat.append((time.perf_counter() - t0) * 1e6)
    lat = np.sort(np.array(lat[20:]))
    t0 = time.perf_counter()
    for _ in range(20):
        c2 = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True, device=device)
        c2.decode_long_host(one, D)
        c2.close()
    fresh_us = (time.perf_counter() - t0) * 1e6 / 20
    out["single_frame_decode_call"] = {"samples": int(one.size), "status": int(r["status"]), "median_us": float(np.median(lat)),
                                       "p90_us": float(lat[int(0.9 * lat.size)]),
                                       "with_a_fresh_context_per_call_us": fresh_us,
                                       "note": "ofdm_rx_decode_long_host on a cached context (pageable input); the last figure is "
                                               "round 3's create / decode / destroy per call"}
    return out


def long_capture(api, torch, device, n=2_000_000, reps=5):
    from tools import bench_cfg3

    ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True, device=device)
    g = torch.Generator(device=ctx.device); g.manual_seed(5)
    pay = torch.randint(0, 256, (1, bench_cfg3.NBYTES), dtype=torch.uint8, device=ctx.device, generator=g)
    tx = ctx.encode_batch(pay)
    D = ctx.data_symbols(bench_cfg3.NBYTES)
    out = {"samples": n, "workload": "one 2080-sample 64QAM frame somewhere in a 2 000 000-sample capture (examples/jetson_rx.rs:15-17,84-86), "
                                     "FIR CHANNEL, CFO, 30 dB noise over the whole capture; max_symbols = 16"}
    rows = {}
    for name, delay in (("packet_at_75_percent", 1_500_001), ("packet_early", 4_001), ("no_packet", -1)):
        cap = torch.empty((1, n), dtype=torch.complex64, device=ctx.device)
        noise = torch.randn((n, 2), dtype=torch.float32, device=ctx.device, generator=g) * 0.004
        if delay >= 0:
            d = torch.tensor([delay], dtype=torch.int32, device=ctx.device)
            fd = torch.tensor([0.4 * math.pi / ctx.S], dtype=torch.float64, device=ctx.device)
            ctx.channel_batch(tx, snr_db=60.0, seed=3, delay=d, f_delta=fd, out=cap)
            cap += torch.view_as_complex(noise).view(1, n)
        else:
            cap.copy_(torch.view_as_complex(noise).view(1, n))
        torch.cuda.synchronize()
        r = ctx.decode_long(cap, D)
        ms = _wall(lambda: ctx.decode_long(cap, D), reps)
        disp = ctx.last_dispatch()
        one = ctx.decode_batch(cap, max_symbols=D)
        ctx.synchronize()
        ms_one = _wall(lambda: (ctx.decode_batch(cap, max_symbols=D), ctx.synchronize()), 2)
        pin = api.pinned_empty((n,), np.complex64)
        pin[:] = cap.cpu().numpy().ravel()
        rh = ctx.decode_long_host(pin, D)
        ms_h = _wall(lambda: ctx.decode_long_host(pin, D), reps)
        same = r["status"] == int(one["status"][0]) == rh["status"] and (r["status"] != 0 or (
            r["offset"] == int(one["offset"][0]) == rh["offset"] and r["len"] == int(one["len"][0]) == rh["len"]
            and bool((r["bytes"][: r["len"]] == one["bytes"][0, : r["len"]]).all())
            and bytes(rh["bytes"][: rh["len"]]) == bytes(r["bytes"][: r["len"]].cpu().numpy())))
        rows[name] = {"status": r["status"], "offset": r["offset"], "decoded_len": r["len"],
                      "ms": ms, "msamples_per_s": n / ms / 1e3, "gb_per_s": n * 8 / ms / 1e6, "dispatch": disp,
                      "as_one_frame_of_the_batch_path_ms": ms_one, "speedup_vs_one_frame": ms_one / ms,
                      "from_pinned_host_ms": ms_h, "from_pinned_host_msamples_per_s": n / ms_h / 1e3,
                      "identical_to_the_one_frame_result": bool(same)}
        del cap, noise, pin
    out.update(rows)
    out["ms"] = rows["packet_at_75_percent"]["ms"]
    out["msamples_per_s"] = rows["packet_at_75_percent"]["msamples_per_s"]
    out["note"] = ("wall clock per synchronous call (search as a batch of 1000 overlapping slices, pick, decode); a 16 MB capture is "
                   "launch- and sync-latency bound, not HBM bound")
    return out


def run(api, torch, device):
    res = {}
    for name, fn in (("h2d_inclusive", h2d_inclusive), ("long_capture", long_capture)):
        try:
            res[name] = fn(api, torch, device)
        except Exception as e:  # the headline must survive
            import traceback
            res[name] = {"error": repr(e), "trace": traceback.format_exc()[-800:]}
        torch.cuda.empty_cache()
    return res
