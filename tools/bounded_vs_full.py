"""Why do a few config-3 frames decode differently with n_lags = 256 than over all lags?  Lists them with the GPU's and the
oracle's timing under both settings."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ofdm_amd import api
from tools import bench_cfg3
from oracle import oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
x, payload = bench_cfg3.synth(api, torch, ctx, n)
full = ctx.sc_correlate(x)
bnd = ctx.sc_correlate(x, n_lags=256)
torch.cuda.synchronize()
df, db = full[0].cpu().numpy(), bnd[0].cpu().numpy()
idx = np.nonzero(df != db)[0]
out = {"frames": n, "differ": int(idx.size), "cases": []}
for j in idx[:12]:
    cap = x[j].cpu().numpy().astype(np.complex128)
    wf = orc.sc_sync(cap, L=80, window_reps=3, n_lags=0, threshold=0.5)
    wb = orc.sc_sync(cap, L=80, window_reps=3, n_lags=256, threshold=0.5)
    m = orc.sc_metric(cap, L=80, window_reps=3, n_lags=0)
    M = np.asarray(m[0] if isinstance(m, tuple) else m)
    d1 = int(np.argmax(M >= 0.5))
    out["cases"].append({"frame": int(j), "gpu_full": int(df[j]), "gpu_bounded": int(db[j]), "oracle_full": int(wf[0]), "oracle_bounded": int(wb[0]),
                         "first_crossing": d1, "M_at_full": float(M[df[j]]), "M_at_bounded": float(M[db[j]]) if db[j] >= 0 else None,
                         "window_end_full": d1 + 240, "window_end_bounded": min(d1 + 240, 255)})
print(json.dumps(out))
