"""Write-side HBM ceilings next to the TX kernels' one-write roofline: a fill of 2 GiB (pure stores) and a device copy (reads +
writes) through torch, timed with the library's HIP events.   python tools/lab/hbm_write_probe.py"""
import torch, time, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ofdm_amd import api
ctx = api.Context(n_fft=64, modulation=6, guard_bands=True)
n = 1 << 28  # complex64 elements: 2 GiB
x = torch.empty(n, dtype=torch.complex64, device="cuda")
res = {}
for name, fn in (("torch_zero", lambda: x.zero_()), ("torch_fill_1", lambda: torch.view_as_real(x).fill_(1.0))):
    fn(); torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(5): fn()
    ms = ctx.timer_stop_ms() / 5
    res[name] = {"ms": ms, "TBps": n * 8 / (ms / 1e3) / 1e12}
y = torch.empty(n, dtype=torch.complex64, device="cuda")
def cp(): y.copy_(x)
cp(); torch.cuda.synchronize(); ctx.timer_start()
for _ in range(5): cp()
ms = ctx.timer_stop_ms() / 5
res["torch_copy"] = {"ms": ms, "read_plus_write_TBps": 2 * n * 8 / (ms / 1e3) / 1e12}
print(json.dumps(res))
