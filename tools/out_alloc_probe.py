"""Does the headline kernel's time depend on WHICH output buffer it writes?  (The pool shows 1.55 vs 1.75 ms per 1 M frames between
processes on one box; the difference is the 576 MB of stores.)  Same input, several output allocations, several rounds."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api
import bench
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
F, syms = 1_000_000, 16
x, payload = bench.synth_cfg2(ctx, torch, F, syms, 30.0, seed=0)
outs = []
pad = []
for i in range(6):
    outs.append(torch.empty((F, syms * ctx.bytes_per_symbol), dtype=torch.uint8, device=ctx.device))
    pad.append(torch.empty((37 + 11 * i) << 20, dtype=torch.uint8, device=ctx.device))   # shift the next allocation
rows = []
for rnd in range(3):
    t = []
    for o in outs:
        ctx.rx_demod(x, syms_per_frame=syms, out=o); torch.cuda.synchronize()
        ctx.timer_start()
        for _ in range(10): ctx.rx_demod(x, syms_per_frame=syms, out=o)
        t.append(round(ctx.timer_stop_ms() / 10, 4))
    rows.append(t)
print(json.dumps({"ms_per_buffer_by_round": rows, "ptr_mod_2MB": [o.data_ptr() % (2 << 20) for o in outs]}))
