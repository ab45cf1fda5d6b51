"""Output-buffer populations of k_demod64 against the size of its store bursts (lab key demod64_burst: 16 / 8 / 4 / 1 groups of 288 bytes per
wavefront and store) and against narrow stores: does a slow buffer care how the bytes arrive?   python tools/lab/out_pop_burst.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import api
import bench
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
F, syms = 1_000_000, 16
nb = syms * ctx.bytes_per_symbol
outs = [torch.empty((F, nb), dtype=torch.uint8, device=ctx.device) for _ in range(2)]
x, payload = bench.synth_cfg2(ctx, torch, F, syms, 30.0, seed=0)
torch.cuda.synchronize()
for i in range(4): outs.append(torch.empty((F, nb), dtype=torch.uint8, device=ctx.device))
def t(oo, reps=5):
    ctx.rx_demod(x, syms_per_frame=syms, out=oo); torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(reps): ctx.rx_demod(x, syms_per_frame=syms, out=oo)
    return round(ctx.timer_stop_ms() / reps, 4)
for bi, o in enumerate(outs):
    row = {"buffer": bi}
    for burst in (16, 8, 4, 1):
        ctx.set_tuning("demod64_burst", burst)
        row[f"burst{burst}"] = t(o)
    ctx.set_tuning("demod64_burst", 16)
    ctx.set_tuning("demod64_narrow_stores", 1); row["narrow"] = t(o); ctx.set_tuning("demod64_narrow_stores", 0)
    row["dispatch"] = ctx.last_dispatch()
    print(json.dumps(row), flush=True)
