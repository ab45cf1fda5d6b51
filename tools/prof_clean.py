"""One timed configuration per process, for `rocprofv3 --kernel-trace --stats` (tools/profile_round.sh): the kernels of the chosen
block are launched ONLY in the measured configuration, so the profiler's per-kernel average is the number the bench block divides
by (in the full bench run several legs share k_sc80 and k_rxframe64).

    python tools/prof_clean.py <which> [frames]
      sc_every_lag   cfg3 frames, ofdm_sc_correlate_batch over every lag (k_sc80, the exact streaming detector, + k_sc_post)
      cfg3_chain     cfg3 frames, ofdm_rx_decode_batch over every lag (k_sc80 + k_sc_post + k_rx_prepare + k_rxframe64)
      cfg3_late      the late-packet / noise-only layout of tools/bench_cfg3.py, same call
      cfg4_chain     cfg4 ring, ofdm_rx_decode_batch over every lag (k_sc_stream + k_rxframe1024)
      cfg4_late      the late-packet layout of tools/bench_large_n.py, same call
      cfg5           k_tx4096 / k_demod4096 on 65 536 continuous symbols
Prints one JSON line: HIP-event milliseconds per call of the same launches."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ofdm_amd import api
from tools import bench_cfg3, bench_large_n

which = sys.argv[1]
reps = 5
out = {"which": which}


def timed(ctx, fn):
    fn(); torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop_ms() / reps


if which in ("sc_every_lag", "cfg3_chain", "cfg3_late"):
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
    if which == "cfg3_late":
        x, _ = bench_cfg3.synth(api, torch, ctx, n, span=bench_cfg3.LATE_SPAN, seed=31, max_delay=bench_cfg3.LATE_SPAN - 2080 - 63,
                                noise_only=bench_cfg3.LATE_NOISE_ONLY)
    else:
        x, _ = bench_cfg3.synth(api, torch, ctx, n, seed=3)
    span = int(x.shape[1])
    if which == "sc_every_lag":
        ctx.set_tuning("sc_first_lags", 0)
        out["ms"] = timed(ctx, lambda: ctx.sc_correlate(x))
        disp = ctx.last_dispatch()
        # k_sc80 stops reading a slot once its decision is determined: slot bytes / time is a throughput, the roofline is taken on the
        # bytes the decision requires, as in the chain blocks
        d_all, _, _ = ctx.sc_correlate(x)
        L, W = ctx.S, ctx.params.sync_window_reps * ctx.S
        out["required_bytes_per_frame"] = bench_cfg3.required_sync_bytes(torch, d_all, span, W, L) / n
        out["bytes_per_frame"] = out["required_bytes_per_frame"]
        out["capture_throughput_of_hbm_peak"] = n * (span * 8 + 16) / (out["ms"] / 1e3) / 8e12
    else:
        out["ms"] = timed(ctx, lambda: ctx.decode_batch(x, max_symbols=16))
        disp = ctx.last_dispatch()
        d_all, _, _ = ctx.sc_correlate(x)                            # (one more, untimed search: d_hat for the required-bytes count)
        L, W = ctx.S, ctx.params.sync_window_reps * ctx.S
        found = int((d_all >= 0).sum())
        out["required_bytes_per_frame"] = (bench_cfg3.required_sync_bytes(torch, d_all, span, W, L) + found * ((5 + 16) * 64 * 8 + bench_cfg3.NBYTES)) / n
        out["bytes_per_frame"] = out["required_bytes_per_frame"]
        out["capture_throughput_of_hbm_peak"] = n * (span * 8 + bench_cfg3.NBYTES) / (out["ms"] / 1e3) / 8e12
    out["roofline_frac"] = n * out["bytes_per_frame"] / (out["ms"] / 1e3) / 8e12   # of 8 TB/s, on the bytes above (tools/bench_cfg3.py's definition)
    out.update(frames=n, slot_samples=span, dispatch=disp)
elif which in ("cfg4_chain", "cfg4_late"):
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    ctx = api.Context(n_fft=1024, modulation=api.QAM64, guard_bands=True, ecc=api.ECC_HAMMING74)
    g = torch.Generator(device=ctx.device); g.manual_seed(4)
    flen = ctx.frame_samples(bench_large_n.CFG4_NBYTES)
    if which == "cfg4_late":
        x, _ = bench_large_n._cfg4_ring(ctx, g, n, flen + 4096 + 256, 40.0, 0, max_delay=4096, noise_only=0.10)
    else:
        x, _ = bench_large_n._cfg4_ring(ctx, g, n, flen + 256, 40.0, 0)
    D = ctx.data_symbols(bench_large_n.CFG4_NBYTES)
    out["ms"] = timed(ctx, lambda: ctx.decode_batch(x, max_symbols=D))
    disp = ctx.last_dispatch()
    span = int(x.shape[1])
    d_all, _, _ = ctx.sc_correlate(x)
    L, W = ctx.S, ctx.params.sync_window_reps * ctx.S
    found = int((d_all >= 0).sum())
    out["required_bytes_per_frame"] = (bench_cfg3.required_sync_bytes(torch, d_all, span, W, L) + found * ((5 + D) * 1024 * 8 + bench_large_n.CFG4_NBYTES)) / n
    out["capture_throughput_of_hbm_peak"] = n * (span * 8 + bench_large_n.CFG4_NBYTES) / (out["ms"] / 1e3) / 8e12
    out["roofline_frac"] = n * out["required_bytes_per_frame"] / (out["ms"] / 1e3) / 8e12   # on the bytes the decision requires
    out.update(frames=n, slot_samples=span, dispatch=disp)
elif which == "cfg5":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    ctx = api.Context(n_fft=4096, modulation=api.QAM256, guard_bands=True)
    g = torch.Generator(device=ctx.device); g.manual_seed(5)
    pay = torch.randint(0, 256, (n * ctx.bytes_per_symbol,), dtype=torch.uint8, device=ctx.device, generator=g)
    xf = torch.empty((n, ctx.S), dtype=torch.complex64, device=ctx.device)
    ob = torch.empty((1, pay.numel()), dtype=torch.uint8, device=ctx.device)
    out["tx_ms"] = timed(ctx, lambda: ctx.tx_symbols(pay, out=xf))
    out["rx_ms"] = timed(ctx, lambda: ctx.rx_demod(xf.view(1, -1), syms_per_frame=n, out=ob))
    by = n * (ctx.S * 8 + ctx.bytes_per_symbol)
    out.update(symbols=n, tx_roofline_frac=by / (out["tx_ms"] / 1e3) / 8e12, rx_roofline_frac=by / (out["rx_ms"] / 1e3) / 8e12)
else:
    raise SystemExit(f"unknown configuration {which}")
print(json.dumps(out))
