"""TX throughput: ofdm_tx_encode_batch on config-3 shaped frames (560-byte payloads, 64-QAM, guard bands, N = 64)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api
from tools import tune_env
tune_env.install()   # OFDM_PROFILE=1 OFDM_TUNE=debug_tx=1: section times of k_txframe64

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
g = torch.Generator(device="cuda"); g.manual_seed(1)
pay = torch.randint(0, 256, (n, 560), dtype=torch.uint8, device="cuda", generator=g)
out = ctx.encode_batch(pay)
torch.cuda.synchronize()
ctx.timer_start()
for _ in range(5):
    ctx.encode_batch(pay, out=out)
ms = ctx.timer_stop_ms() / 5
samples = n * out.shape[-1]
print(json.dumps({"frames": n, "samples_per_frame": out.shape[-1], "tx_ms": ms, "tx_msamples_per_s": samples / ms / 1e3,
                  "hbm_frac_of_one_write": samples * 8 / (ms / 1e3) / 1e9 / 8000.0}))

if ctx.get_tuning("profile_build") and ctx.get_tuning("debug_tx"):
    o = ctx.encode_batch(pay)
    torch.cuda.synchronize()
    r = torch.view_as_real(o)[:, :2, :].reshape(n, 4).double()
    for i, nm in enumerate(["stage", "compute", "output"]):
        print(nm, "median", float(r[:, i].median()), "mean", float(r[:, i].mean()))
