import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api
from tools import tune_env
tune_env.install()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
x = (torch.randn((n, 2176, 2), device="cuda") * 0.1)
x = torch.view_as_complex(x).contiguous()
ctx.sc_correlate(x); torch.cuda.synchronize()
ctx.timer_start()
for _ in range(5): ctx.sc_correlate(x)
ms = ctx.timer_stop_ms() / 5
print(ctx.get_tuning("debug_sc"), "ms", ms, "GB/s", n * 2176 * 8 / ms / 1e6)
