"""Headline kernel time against the OFFSET of its output buffer inside one arena (same input): which address bits matter?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api
import bench
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
F, syms = 1_000_000, 16
x, payload = bench.synth_cfg2(ctx, torch, F, syms, 30.0, seed=0)
nb = F * syms * ctx.bytes_per_symbol
arena = torch.empty(nb + (1 << 31), dtype=torch.uint8, device=ctx.device)
base = arena.data_ptr()
offs = [0, 4096, 65536, 1 << 20] + [k << 21 for k in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 640, 768, 896, 1000)]
rows = []
for off in offs:
    o = arena[off: off + nb].view(F, syms * ctx.bytes_per_symbol)
    ctx.rx_demod(x, syms_per_frame=syms, out=o); torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(8): ctx.rx_demod(x, syms_per_frame=syms, out=o)
    rows.append((off >> 20, round(ctx.timer_stop_ms() / 8, 3)))
print(json.dumps({"x_ptr": hex(x.data_ptr()), "arena": hex(base), "delta_mb": (base - x.data_ptr()) >> 20, "off_mb_ms": rows}))
