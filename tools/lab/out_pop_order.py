"""Does the headline kernel's time depend on whether its output buffer was allocated BEFORE or AFTER the 10 GB input (bench.py allocates
it before)?  One process: two buffers before the input, six after; each timed over 10 launches, three rounds."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import api
import bench
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
F, syms = 1_000_000, 16
nb = syms * ctx.bytes_per_symbol
before = [torch.empty((F, nb), dtype=torch.uint8, device=ctx.device) for _ in range(2)]
x, payload = bench.synth_cfg2(ctx, torch, F, syms, 30.0, seed=0)
after = [torch.empty((F, nb), dtype=torch.uint8, device=ctx.device) for _ in range(4)]
torch.cuda.empty_cache()
late = [torch.empty((F, nb), dtype=torch.uint8, device=ctx.device) for _ in range(2)]
rows = []
for rnd in range(3):
    t = []
    for o in before + after + late:
        ctx.rx_demod(x, syms_per_frame=syms, out=o); torch.cuda.synchronize()
        ctx.timer_start()
        for _ in range(10): ctx.rx_demod(x, syms_per_frame=syms, out=o)
        t.append(round(ctx.timer_stop_ms() / 10, 4))
    rows.append(t)
print(json.dumps({"order": "2 before the input, 4 after, 2 after empty_cache", "x_ptr": hex(x.data_ptr()),
                  "ptrs": [hex(o.data_ptr()) for o in before + after + late], "ms_by_round": rows}))
