import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api
from tools import bench_cfg3
n = 131072
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
x, payload = bench_cfg3.synth(api, torch, ctx, n, 2176)
ctx.sc_correlate(x); torch.cuda.synchronize()
ctx.timer_start()
for _ in range(5): d, fd, m = ctx.sc_correlate(x)
ms = ctx.timer_stop_ms() / 5
print(os.environ.get("OFDM_SC_DEBUG", "0"), "ms", ms, "GB/s", n * 2176 * 8 / ms / 1e6, "found", int((d >= 0).sum()))
