"""Sweep of sc_first_lags (the lags the first of the two Schmidl-Cox launches looks at) on the stated config-3 placement (delay 1..64) and on
the late-packet layout (2544-sample slots, delay 1..401, 10 % empty): time of the search and of the whole decode chain."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api
from tools import bench_cfg3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
sets = {"early": bench_cfg3.synth(api, torch, ctx, n, seed=3)[0],
        "late": bench_cfg3.synth(api, torch, ctx, n, span=bench_cfg3.LATE_SPAN, seed=31, max_delay=bench_cfg3.LATE_SPAN - 2080 - 63, noise_only=0.10)[0]}
for first in [int(a) for a in sys.argv[2:]] or (0, 384, 448, 512, 640, 768, 900, 948):
    row = {"first": first}
    ctx.set_tuning("sc_first_lags", first)
    for name, x in sets.items():
        for what, fn in (("sc", lambda: ctx.sc_correlate(x)), ("chain", lambda: ctx.decode_batch(x, max_symbols=16))):
            fn(); torch.cuda.synchronize(); ctx.timer_start()
            for _ in range(5): fn()
            row[f"{name}_{what}_ms"] = round(ctx.timer_stop_ms() / 5, 4)
        row[f"{name}_redo"] = ctx.get_tuning("stat_sc_redo_frames")
    print(json.dumps(row), flush=True)
