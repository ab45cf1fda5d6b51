"""k_sc80 (exact streaming detector, kernels_sc80.hip) against the round-4 f32 filter pair (tuning no_sc80 = 1) on config-3
captures: per delay band the time of both searches over every lag, whether every timing index agrees, and the largest CFO / metric
difference.  Run on the GPU box: python tools/lab/sc80_ab.py [frames]"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import api
from tools import bench_cfg3, tune_env
tune_env.install()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
g = torch.Generator(device="cuda"); g.manual_seed(9)
pay = torch.randint(0, 256, (n, bench_cfg3.NBYTES), dtype=torch.uint8, device="cuda", generator=g)
tx = ctx.encode_batch(pay)
for name, span, lo, hi in (("cfg3 slot, delay 1..64", bench_cfg3.SPAN, 1, 64), ("late slot, delay 1..64", bench_cfg3.LATE_SPAN, 1, 64),
                           ("late slot, 64..240", bench_cfg3.LATE_SPAN, 64, 240), ("late slot, 240..401", bench_cfg3.LATE_SPAN, 240, 401),
                           ("late slot, noise only", bench_cfg3.LATE_SPAN, -1, -1)):
    x = torch.empty((n, span), dtype=torch.complex64, device="cuda")
    if lo >= 0:
        d = torch.randint(lo, hi + 1, (n,), device="cuda", generator=g, dtype=torch.int32)
        fd = (torch.rand((n,), device="cuda", generator=g, dtype=torch.float64) * 1.9 - 0.95) * math.pi / 80
        ctx.channel_batch(tx, snr_db=30.0, seed=77, delay=d, f_delta=fd, out=x)
    else:
        x.copy_(torch.view_as_complex(torch.randn((n, span, 2), device="cuda", generator=g) * 0.004))
    row = {"band": name}
    res = {}
    for key in (0, 2, 1):
        ctx.set_tuning("no_sc80", 1 if key == 1 else 0)
        ctx.set_tuning("sc80_depth", 2 if key == 2 else 1)
        r = ctx.sc_correlate(x); torch.cuda.synchronize()
        ctx.timer_start()
        for _ in range(3): ctx.sc_correlate(x)
        ms = ctx.timer_stop_ms() / 3
        tag = {0: "sc80", 1: "pair", 2: "sc80_d2"}[key]
        row[f"{tag}_ms"] = round(ms, 4)
        row[f"{tag}_gbs_slot"] = round(n * span * 8 / ms / 1e6, 1)
        row[f"{tag}_slow"] = ctx.get_tuning("stat_sc_slow_frames")
        row[f"{tag}_dispatch"] = ctx.last_dispatch()
        res[key] = [t.clone() for t in r]
    row["d_hat_equal"] = bool((res[0][0] == res[1][0]).all())
    row["d_hat_differs"] = int((res[0][0] != res[1][0]).sum())
    row["found"] = int((res[0][0] >= 0).sum())
    row["max_cfo_diff"] = float((res[0][1] - res[1][1]).abs().max())
    row["max_metric_diff"] = float((res[0][2] - res[1][2]).abs().max())
    print(json.dumps(row), flush=True)
    del x
