"""k_rxframe64 as one kernel with both frame bodies (tuning no_rxframe64_split = 1: round 3, three waves per SIMD) against the
common-body / cut-body pair (four waves per SIMD for the common body), same process, same frames; results compared.
python tools/lab/rx64_ab.py [frames] [cut_every]   (cut_every > 0: every cut_every-th capture is cut short inside its data symbols)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import api
from tools import bench_cfg3

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
x, payload = bench_cfg3.synth(api, torch, ctx, n, 2176)
D = 16
out, res = {}, {}
for rep in range(2):
    for split_off in (1, 0):
        ctx.set_tuning("no_rxframe64_split", split_off)
        for lags, flen in ((256, 2176), (256, 1900)):   # whole captures; every capture cut inside its data symbols
            r = ctx.decode_batch(x, max_symbols=D, n_lags=lags, frame_len=flen)
            torch.cuda.synchronize()
            ctx.timer_start()
            for _ in range(5):
                r = ctx.decode_batch(x, max_symbols=D, n_lags=lags, frame_len=flen)
            ms = ctx.timer_stop_ms() / 5
            out.setdefault(f"{'one_kernel' if split_off else 'split'}_len{flen}_ms", []).append(round(ms, 4))
            res[(split_off, flen)] = {k: v.clone() for k, v in r.items()}
            out[f"{'one_kernel' if split_off else 'split'}_dispatch"] = ctx.last_dispatch()
for flen in (2176, 1900):
    a, b = res[(1, flen)], res[(0, flen)]
    out[f"len{flen}_identical"] = all(bool(torch.equal(a[k], b[k])) for k in ("status", "len", "offset", "f_delta")) and \
        bool(((a["bytes"] == b["bytes"]) | (torch.arange(a["bytes"].shape[1], device=a["bytes"].device)[None, :] >= a["len"][:, None])).all())
    out[f"len{flen}_decoded"] = int((b["status"] == 0).sum())
print(json.dumps(out))
