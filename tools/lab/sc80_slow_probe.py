"""Which frames does k_sc80 hand to the slow list, and why?  Binary search on the batch with the stat counter, then prints the
frame's delay and the oracle's view of the lags around the untrusted window."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
from ofdm_amd import api
from tools import bench_cfg3
n = 262144
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
g = torch.Generator(device="cuda"); g.manual_seed(9)
pay = torch.randint(0, 256, (n, bench_cfg3.NBYTES), dtype=torch.uint8, device="cuda", generator=g)
tx = ctx.encode_batch(pay)
span = bench_cfg3.LATE_SPAN
for lo, hi in ((1, 64), (1, 64), (64, 240), (240, 401)):
    x = torch.empty((n, span), dtype=torch.complex64, device="cuda")
    d = torch.randint(lo, hi + 1, (n,), device="cuda", generator=g, dtype=torch.int32)
    fd = (torch.rand((n,), device="cuda", generator=g, dtype=torch.float64) * 1.9 - 0.95) * math.pi / 80
    ctx.channel_batch(tx, snr_db=30.0, seed=77, delay=d, f_delta=fd, out=x)
def slow(a, b):
    ctx.sc_correlate(x[a:b]); torch.cuda.synchronize()
    return ctx.get_tuning("stat_sc_slow_frames")
a, b = 0, n
print("slow in batch", slow(a, b))
while b - a > 1:
    m = (a + b) // 2
    if slow(a, m): b = m
    else: a = m
print("frame", a, "delay", int(d[a]))
c = x[a].cpu().numpy().astype(np.complex128)
e = np.abs(c) ** 2
se = np.concatenate([[0], np.cumsum(e)])
E = se[240:] - se[:-240]
print("first samples |x|", np.abs(c[:8]))
print("min E", E.min(), "at", int(E.argmin()), "total", se[-1])
r = ctx.sc_correlate(x[a:a+1]); print([t.cpu().numpy() for t in r])
