"""Profiling aid: per-section time of the Schmidl-Cox filter kernel (s_memtime ticks per frame, median over frames)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import _lib, api
from tools import bench_cfg3
_lib.use_profile_build()  # the section timers live in libofdm_hip_profile.so only
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
lags = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # 0: every lag (k_sc_cf<256,2,4,0>); e.g. 384: the 128-chunk kernel of bounded / first-lags searches
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True, tuning={"sc_first_lags": 0})   # one launch: the timers replace the results
x, _ = bench_cfg3.synth(api, torch, ctx, n, 2176)
names = ["dma_wait", "(unused)", "phase1", "coarse", "fine", "fine.slide", "fine.select", "fine.exact"]
for k, nm in enumerate(names):
    ctx.set_tuning("debug_sc", 10 + k)
    dh, _, _ = ctx.sc_correlate(x, n_lags=lags)
    torch.cuda.synchronize()
    d = dh.to(torch.float64)
    print(nm, "median", float(d.median()), "mean", float(d.mean()), "p90", float(d.quantile(0.9)))
ctx.set_tuning("debug_sc", 0)
ctx.timer_start()
for _ in range(5): ctx.sc_correlate(x, n_lags=lags)
print("ms per", n, "frames:", ctx.timer_stop_ms() / 5, ctx.last_dispatch())
