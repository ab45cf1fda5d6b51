"""Output-buffer populations of k_demod64: is "slow" a property of the OUTPUT buffer alone or of the (input, output) pair, and is it uniform
over the buffer?  Six output buffers (two allocated before the first input) x two copies of the input at different addresses, and the two
halves of the batch separately.   python tools/lab/out_pop_pairs.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import api
import bench
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
F, syms = 1_000_000, 16
nb = syms * ctx.bytes_per_symbol
outs = [torch.empty((F, nb), dtype=torch.uint8, device=ctx.device) for _ in range(2)]
x, payload = bench.synth_cfg2(ctx, torch, F, syms, 30.0, seed=0)
torch.cuda.synchronize()
for i in range(4): outs.append(torch.empty((F, nb), dtype=torch.uint8, device=ctx.device))
x2 = x.clone()
def t(xx, oo, reps=5):
    n = xx.shape[0]
    ctx.rx_demod(xx, syms_per_frame=syms, out=oo); torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(reps): ctx.rx_demod(xx, syms_per_frame=syms, out=oo)
    return round(ctx.timer_stop_ms() / reps * (1_000_000 / n), 4)      # per 1 M frames
print(json.dumps({"x": hex(x.data_ptr()), "x2": hex(x2.data_ptr()), "outs": [hex(o.data_ptr()) for o in outs]}))
for bi, o in enumerate(outs):
    h = F // 2
    row = {"buffer": bi, "x": t(x, o), "x2": t(x2, o), "x_again": t(x, o),
           "first_half": t(x[:h], o[:h]), "second_half": t(x[h:], o[h:]),
           "x_first_half_into_out_second_half": t(x[:h], o[h:]), "x_second_half_into_out_first_half": t(x[h:], o[:h])}
    print(json.dumps(row), flush=True)
