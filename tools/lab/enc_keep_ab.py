"""Frames whose data symbols fit one workgroup step: the keep-in-registers form (k_txframe_mid<once>, three waves per SIMD) against the
optimistic two-pass form (lab key txframe_keep_steps = 0; four waves per SIMD for R <= 8).   python tools/lab/enc_keep_ab.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import api
for n, D in ((128, 16), (128, 8), (256, 8), (512, 4), (1024, 2), (2048, 1)):
    ctx = api.Context(n_fft=n, modulation=api.QAM64, guard_bands=True)
    g = torch.Generator(device="cuda"); g.manual_seed(n)
    nbytes = D * ctx.bytes_per_symbol - 16
    fs = ctx.frame_samples(nbytes); nfr = (1 << 28) // fs
    pay = torch.randint(0, 256, (nfr, nbytes), dtype=torch.uint8, device="cuda", generator=g)
    fo = ctx.encode_batch(pay)
    row = {"n_fft": n, "D": D}
    for keep in (1, 0, 1, 0):
        ctx.set_tuning("txframe_keep_steps", keep)
        ctx.encode_batch(pay, out=fo); torch.cuda.synchronize(); ctx.timer_start()
        for _ in range(5): ctx.encode_batch(pay, out=fo)
        ms = ctx.timer_stop_ms() / 5
        row.setdefault("keep" if keep else "two_pass", []).append(round(nfr * (fs * 8 + nbytes) / ms / 1e6 / 8000, 3))
        row["dispatch_keep" if keep else "dispatch_two_pass"] = ctx.last_dispatch()
    print(json.dumps(row), flush=True)
