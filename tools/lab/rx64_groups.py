"""k_rxframe64's time against the number of 8-symbol groups per frame (D = 8, 16, 32, 48 data symbols): the slope is the cost of a
group, the intercept the per-frame part (scalars, training blocks, channel estimate, exposed round trips).  One D per process under
rocprofv3 --kernel-trace --stats:   python tools/lab/rx64_groups.py D [frames]"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import api

D = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
nbytes = D * ctx.bytes_per_symbol - 16
g = torch.Generator(device="cuda"); g.manual_seed(D)
pay = torch.randint(0, 256, (n, nbytes), dtype=torch.uint8, device="cuda", generator=g)
tx = ctx.encode_batch(pay)
d = torch.randint(1, 65, (n,), device="cuda", generator=g, dtype=torch.int32)
fd = (torch.rand((n,), device="cuda", generator=g, dtype=torch.float64) * 1.9 - 0.95) * math.pi / 80
x = ctx.channel_batch(tx, snr_db=30.0, seed=7, delay=d, f_delta=fd, span=tx.shape[1] + 96)
r = ctx.decode_batch(x, max_symbols=D); torch.cuda.synchronize()
ctx.timer_start()
for _ in range(5): ctx.decode_batch(x, max_symbols=D)
ms = ctx.timer_stop_ms() / 5
ok = int((r["len"] == nbytes).sum())
print(json.dumps({"D": D, "frames": n, "chain_ms": round(ms, 4), "decoded_full_length": ok, "dispatch": ctx.last_dispatch()}))
