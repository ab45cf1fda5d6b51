"""Which allocations draw the slow population?  Variants (one per process, argv[1]):
  out_x          output first, then the 10 GB input (bench.py's round-1..4 order)
  x_out          input first
  dummy_out_x    a 2 GB allocation that stays, then output, then input
  freed_out_x    a 2 GB allocation that is freed again (torch's cache emptied), then output, then input
  small_out_x    256 allocations of 8 MB that stay, then output, then input
  hip_out_x      output from hipMalloc through the library (ofdm_dev_alloc), then input"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import api
import bench
mode = sys.argv[1]
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
F, syms = 1_000_000, 16
nb = syms * ctx.bytes_per_symbol
keep = []
mk = lambda: torch.empty((F, nb), dtype=torch.uint8, device=ctx.device)
if mode == "dummy_out_x": keep.append(torch.empty(2 << 30, dtype=torch.uint8, device=ctx.device))
if mode == "freed_out_x":
    t = torch.empty(2 << 30, dtype=torch.uint8, device=ctx.device); del t; torch.cuda.empty_cache()
if mode == "small_out_x": keep += [torch.empty(8 << 20, dtype=torch.uint8, device=ctx.device) for _ in range(256)]
out = None
if mode != "x_out": out = mk()
x, payload = bench.synth_cfg2(ctx, torch, F, syms, 30.0, seed=0)
if mode == "x_out": out = mk()
ts = []
for rnd in range(3):
    ctx.rx_demod(x, syms_per_frame=syms, out=out); torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(10): ctx.rx_demod(x, syms_per_frame=syms, out=out)
    ts.append(round(ctx.timer_stop_ms() / 10, 4))
print(json.dumps({"mode": mode, "ms": ts, "out_ptr": hex(out.data_ptr()), "x_ptr": hex(x.data_ptr())}))
