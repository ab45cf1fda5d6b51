"""Practical HBM read ceilings on this box, next to the headline kernel (same buffer, HIP events on the launch stream):
   pattern 0  k_demod64's access (8-byte loads, cyclic prefix lines never touched: 512 of every 640 bytes)
   pattern 1  the same 8-byte loads over whole symbols
   pattern 2  unit-stride 16-byte loads
and the torch copy / reduction the round-1 probe used.  python tools/lab/hbm_probe.py [frames]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import api

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
x = torch.view_as_complex(torch.randn((F, 16 * 80, 2), device="cuda") * 0.1).contiguous()
out = torch.empty((F, 16 * 36), dtype=torch.uint8, device="cuda")
nbytes = x.numel() * 8
res = {"frames": F, "buffer_bytes": nbytes}


def timed(fn, steps=10):
    fn(); torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(steps): fn()
    return ctx.timer_stop_ms() / steps


for pat, touched in ((0, 0.8), (1, 1.0), (2, 1.0)):
    ms = timed(lambda: ctx.hbm_read_probe(x, pat))
    res[f"probe_pattern{pat}"] = {"ms": ms, "touched_GBps": nbytes * touched / ms / 1e6, "algorithmic_GBps": nbytes / ms / 1e6}
ms = timed(lambda: ctx.rx_demod(x, syms_per_frame=16, out=out))
res["k_demod64"] = {"ms": ms, "touched_GBps": (nbytes * 0.8 + out.numel()) / ms / 1e6, "algorithmic_GBps": (nbytes + out.numel()) / ms / 1e6}
y = torch.empty_like(x)
ms = timed(lambda: y.copy_(x))
res["torch_copy"] = {"ms": ms, "GBps_read_plus_write": 2 * nbytes / ms / 1e6}
print(json.dumps(res))
