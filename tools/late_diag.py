"""Where the one-tile Schmidl-Cox search spends its time as a function of the packet's position in its slot: per delay band the
kernel time of the single-launch search over every lag and the number of frames its f32 filter handed to the all-f64 kernel."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api
from tools import bench_cfg3, tune_env
tune_env.install()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
span = bench_cfg3.LATE_SPAN
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
g = torch.Generator(device="cuda"); g.manual_seed(9)
pay = torch.randint(0, 256, (n, bench_cfg3.NBYTES), dtype=torch.uint8, device="cuda", generator=g)
tx = ctx.encode_batch(pay)
rows = []
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1   # one band only (for a rocprofv3 kernel trace of that band)
for bi, (name, lo, hi) in enumerate((("delay 1..64", 1, 64), ("64..160", 64, 160), ("160..240", 160, 240), ("240..320", 240, 320), ("320..401", 320, 401), ("noise only", -1, -1))):
    if only >= 0 and bi != only:
        continue
    x = torch.empty((n, span), dtype=torch.complex64, device="cuda")
    if lo >= 0:
        d = torch.randint(lo, hi + 1, (n,), device="cuda", generator=g, dtype=torch.int32)
        fd = (torch.rand((n,), device="cuda", generator=g, dtype=torch.float64) * 1.9 - 0.95) * math.pi / 80
        ctx.channel_batch(tx, snr_db=30.0, seed=77, delay=d, f_delta=fd, out=x)
    else:
        x.copy_(torch.view_as_complex(torch.randn((n, span, 2), device="cuda", generator=g) * 0.004))
    row = {"band": name}
    for first in (0, 384):
        ctx.set_tuning("sc_first_lags", first)
        ctx.sc_correlate(x); torch.cuda.synchronize()
        ctx.timer_start()
        for _ in range(3): ctx.sc_correlate(x)
        ms = ctx.timer_stop_ms() / 3
        row[f"first{first}_ms"] = round(ms, 4)
        row[f"first{first}_slow"] = ctx.get_tuning("stat_sc_slow_frames")
        row[f"first{first}_redo"] = ctx.get_tuning("stat_sc_redo_frames")
        row[f"first{first}_dispatch"] = ctx.last_dispatch()
    rows.append(row)
    print(json.dumps(row), flush=True)
