"""Per-frame section times (s_memtime ticks, wave 0) of the one-pass receive kernel (profile build, tuning debug_sc = 20..26):
DMA wait | phase 1 | coarse | fine + timing tail | channel estimate + data groups | finish | whole iteration."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import _lib, api
from tools import bench_cfg3
_lib.use_profile_build()
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True, rx_path=api.RX_ONE_PASS)
x, payload = bench_cfg3.synth(api, torch, ctx, 65536, 2176)
names = ["dma_wait", "phase1", "coarse", "fine+timing", "chest+data", "finish", "iteration"]
out = {}
for i, nm in enumerate(names):
    ctx.set_tuning("debug_sc", 20 + i)
    r = ctx.decode_batch(x, max_symbols=ctx.data_symbols(560))
    torch.cuda.synchronize()
    out[nm] = float(r["metric"].double().mean())
print(json.dumps(out))
