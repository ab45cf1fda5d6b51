"""k_sc_cf ablation on PACKET frames (config-3 captures), profile build: OFDM_PROFILE=1 OFDM_TUNE=debug_sc=K per process
(0 full, 2 phase 1, 3 + coarse, 5 + slide, 4 + select)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api
from tools import bench_cfg3, tune_env
tune_env.install()
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
x, _ = bench_cfg3.synth(api, torch, ctx, 262144)
ctx.sc_correlate(x); torch.cuda.synchronize()
ctx.timer_start()
for _ in range(5): ctx.sc_correlate(x)
print("debug_sc", ctx.get_tuning("debug_sc"), "sc_ms", ctx.timer_stop_ms() / 5)
