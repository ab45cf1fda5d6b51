"""The two populations of the headline kernel's output buffers (1.57 / 1.73 ms per 1 M frames) under PMC counters: ONE
process, eight 576 MB output buffers, every buffer timed (HIP events, 6 launches) and then written by exactly 2 more launches, in
buffer order -- so that the per-dispatch counter rows of `rocprofv3 --pmc ...` (tools/lab/out_pop_pmc.sh) can be laid beside the times.
Prints one JSON line: pointers, times and the dispatch order."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import api
import bench
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
F, syms = 1_000_000, 16
# the FIRST large allocations of a process draw the slow population (tools/lab/out_pop_order.py): two buffers before the 10 GB input, as
# bench.py allocates its output, then six after it
outs = [torch.empty((F, syms * ctx.bytes_per_symbol), dtype=torch.uint8, device=ctx.device) for _ in range(2)]
x, payload = bench.synth_cfg2(ctx, torch, F, syms, 30.0, seed=0)
torch.cuda.synchronize()
for i in range(6):
    outs.append(torch.empty((F, syms * ctx.bytes_per_symbol), dtype=torch.uint8, device=ctx.device))
rows = []
for o in outs:
    ctx.rx_demod(x, syms_per_frame=syms, out=o); torch.cuda.synchronize()      # launch 1 of this buffer (warm-up)
    ctx.timer_start()
    for _ in range(6): ctx.rx_demod(x, syms_per_frame=syms, out=o)              # launches 2..7
    ms = ctx.timer_stop_ms() / 6
    for _ in range(2): ctx.rx_demod(x, syms_per_frame=syms, out=o)              # launches 8, 9: the rows to read the counters from
    torch.cuda.synchronize()
    rows.append({"ptr": hex(o.data_ptr()), "ptr_mod_2MB": o.data_ptr() % (2 << 20), "ms": round(ms, 4)})
print(json.dumps({"launches_per_buffer": 9, "counter_rows": "the last two of every nine k_demod64 dispatches", "x_ptr": hex(x.data_ptr()),
                  "buffers": rows}))
