"""The output-buffer populations of k_demod64 against the cache-policy bits of its image stores (lab key
demod64_store_policy: 0 default, 1 nt, 2 sc1, 3 sc0 sc1): two output buffers allocated before the 10 GB input (the ones that draw the slow
population about half the time), four after it; every buffer timed under every policy, twice.   python tools/lab/out_pop_policy.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import api
import bench
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
F, syms = 1_000_000, 16
outs = [torch.empty((F, syms * ctx.bytes_per_symbol), dtype=torch.uint8, device=ctx.device) for _ in range(2)]
x, payload = bench.synth_cfg2(ctx, torch, F, syms, 30.0, seed=0)
torch.cuda.synchronize()
for i in range(4):
    outs.append(torch.empty((F, syms * ctx.bytes_per_symbol), dtype=torch.uint8, device=ctx.device))
ref = None
rows = []
for bi, o in enumerate(outs):
    row = {"buffer": bi, "allocated": "before the input" if bi < 2 else "after the input"}
    for rep in range(2):
        for pol in (0, 1, 2, 3):
            ctx.set_tuning("demod64_store_policy", pol)
            ctx.rx_demod(x, syms_per_frame=syms, out=o); torch.cuda.synchronize()
            ctx.timer_start()
            for _ in range(5): ctx.rx_demod(x, syms_per_frame=syms, out=o)
            row.setdefault(f"policy{pol}_ms", []).append(round(ctx.timer_stop_ms() / 5, 4))
            if ref is None: ref = o.clone()
            row.setdefault("bytes_equal", True)
            row["bytes_equal"] = row["bytes_equal"] and bool((o == ref).all())
    rows.append(row)
    print(json.dumps(row), flush=True)
