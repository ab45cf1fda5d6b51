"""A/B of the config-3 receive chain on one GPU: the one-pass kernel (k_sc_cf<..., BPS>) against the staged chain
(ofdm_params.rx_path = OFDM_RX_STAGED / OFDM_RX_ONE_PASS), same frames, outputs compared.  python tools/cfg3_ab.py [frames] [steps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api
from tools import bench_cfg3

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
from tools import tune_env
tune_env.install()
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True, rx_path=api.RX_STAGED)
ctx1 = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True, rx_path=api.RX_ONE_PASS)
x, payload = bench_cfg3.synth(api, torch, ctx, n, 2176)
D = ctx.data_symbols(560)
out = {}
res = {}
for name, ctx in (("one_pass", ctx1), ("staged", ctx)):
    for lags in (0, 256):
        r = ctx.decode_batch(x, max_symbols=D, n_lags=lags)
        torch.cuda.synchronize()
        ctx.timer_start()
        for _ in range(steps): r = ctx.decode_batch(x, max_symbols=D, n_lags=lags)
        ms = ctx.timer_stop_ms() / steps
        res[(name, lags)] = {k: v.clone() for k, v in r.items()}
        out[f"{name}_lags{lags}_ms"] = ms
        out[f"{name}_lags{lags}_hbm_frac_one_read"] = n * (2176 * 8 + 560) / (ms / 1e3) / 8e12
for lags in (0, 256):
    a, b = res[("one_pass", lags)], res[("staged", lags)]
    out[f"lags{lags}_status_equal"] = bool((a["status"] == b["status"]).all())
    out[f"lags{lags}_offset_equal"] = bool((a["offset"] == b["offset"]).all())
    out[f"lags{lags}_len_equal"] = bool((a["len"] == b["len"]).all())
    out[f"lags{lags}_frames_with_different_bytes"] = int((a["bytes"][:, :560] != b["bytes"][:, :560]).any(dim=1).sum())
    out[f"lags{lags}_max_cfo_diff"] = float((a["f_delta"] - b["f_delta"]).abs().max())
    ok = (a["status"] == 0) & (a["len"] == 560)
    out[f"lags{lags}_decoded"] = int(ok.sum())
    out[f"lags{lags}_frames_exact"] = int(((a["bytes"][:, :560] == payload).all(dim=1) & ok).sum())
print(json.dumps(out))
