"""Quick A/B on one GPU: config-4 chain with the long-period Schmidl-Cox path (kernels_scbig.hip) against the round-1
k_sc_tile path (tuning no_sc_big), and config-5 TX / RX.  python tools/lab/cfg45_ab.py [cfg4|cfg5]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools import bench_large_n as b
out = {}
which = sys.argv[1] if len(sys.argv) > 1 else "both"
def brief(r):
    return {k: {kk: r[k][kk] for kk in ("ms_per_pass", "msamples_per_s", "frames_decoded_exactly")} | {"frac": r[k]["roofline"]["frac"]} for k in ("full_chain_all_lags", "full_chain_bounded_2048_lags")}
if which in ("cfg5", "both"):
    r = b.cfg5(65536, 5, cpu=False)
    out["cfg5"] = {k: r[k] for k in ("tx_ms", "rx_ms", "rx_bytes_equal_tx_payload", "tx_fused_vs_staged_max_rel_err")} | {"tx_frac": r["roofline_tx"]["frac"], "rx_frac": r["roofline_rx"]["frac"]}
    torch.cuda.empty_cache()
if which in ("cfg4", "both"):
    out["cfg4_scbig"] = brief(b.cfg4(16384, 65536, cpu=False))
    torch.cuda.empty_cache()
    from ofdm_amd import api
    api.DEFAULT_TUNING["no_sc_big"] = 1
    out["cfg4_sc_tile"] = brief(b.cfg4(16384, 65536, cpu=False))
    api.DEFAULT_TUNING.pop("no_sc_big")
print(json.dumps(out))
