"""Per-frame section times (s_memtime ticks) of k_sc_stream on config-4 frames (profile build, tuning debug_sc = 40 + i).
Producer wavefront: tile waits | sums + scans | barrier waits.  Consumer wavefront: bounds + lists | evaluations before the
crossing | barrier waits | closing the window | whole frame."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import math
import torch
from ofdm_amd import _lib, api
_lib.use_profile_build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = api.Context(n_fft=1024, modulation=api.QAM64, guard_bands=True, ecc=api.ECC_HAMMING74)
g = torch.Generator(device=ctx.device); g.manual_seed(4)
pay = torch.randint(0, 256, (n, 1304), dtype=torch.uint8, device=ctx.device, generator=g)
tx = ctx.encode_batch(pay)
d = torch.randint(1, 65, (n,), device=ctx.device, generator=g, dtype=torch.int32)
fd = (torch.rand((n,), device=ctx.device, generator=g, dtype=torch.float64) * 1.9 - 0.95) * math.pi / ctx.S
x = ctx.channel_batch(tx, snr_db=40.0, seed=4_000_003, delay=d, f_delta=fd, span=tx.shape[1] + 256)
names = ["prod_tile_wait", "prod_sums_scans", "prod_barrier_wait", "cons_bounds_lists", "cons_evaluations", "cons_barrier_wait", "cons_close", "frame"]
out = {}
for i, nm in enumerate(names):
    ctx.set_tuning("debug_sc", 40 + i)
    dh, f_, m = ctx.sc_correlate(x)
    torch.cuda.synchronize()
    out[nm] = float(m.double().mean())
ctx.set_tuning("debug_sc", 0)
ctx.timer_start()
for _ in range(5): ctx.sc_correlate(x)
out["ms_per_call"] = ctx.timer_stop_ms() / 5
print(json.dumps(out))
