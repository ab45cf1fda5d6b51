"""Frame-level encode (k_txframe_mid / k_txframe4096): the optimistic one-pass scheme against the build-twice scheme (lab key
no_txframe_optimistic), for D = 16 and D = 64 data symbols per frame (the header blocks are 10 / 26 and 10 / 74 of the frame), next to
the continuous-stream TX rate of the same length.   python tools/lab/enc_ab.py [log2 samples] [N ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import api

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 27
ns = [int(a) for a in sys.argv[2:]] or [128, 256, 512, 1024, 2048, 4096]
for n in ns:
    ctx = api.Context(n_fft=n, modulation=api.QAM64, guard_bands=True)
    S = ctx.S
    g = torch.Generator(device=ctx.device); g.manual_seed(n)

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        return ctx.timer_stop_ms() / reps

    row = {"n_fft": n}
    n_sym = (1 << lg) // S
    data = torch.randint(0, 256, (n_sym * ctx.bytes_per_symbol,), dtype=torch.uint8, device=ctx.device, generator=g)
    x = ctx.tx_symbols(data, n_sym)
    ms = timed(lambda: ctx.tx_symbols(data, n_sym, out=x))
    row["tx_stream_frac"] = round(n_sym * (S * 8 + ctx.bytes_per_symbol) / ms / 1e6 / 8000, 3)
    del x, data
    for D in (16, 64):
        nbytes = D * ctx.bytes_per_symbol - 16
        fs = ctx.frame_samples(nbytes)
        nfr = max(1, (1 << lg) // fs)
        pay = torch.randint(0, 256, (nfr, nbytes), dtype=torch.uint8, device=ctx.device, generator=g)
        fo = ctx.encode_batch(pay)
        for key in (0, 1):
            ctx.set_tuning("no_txframe_optimistic", key)
            ms = timed(lambda: ctx.encode_batch(pay, out=fo))
            row[f"D{D}_{'twice' if key else 'once'}_ms"] = round(ms, 4)
            row[f"D{D}_{'twice' if key else 'once'}_frac"] = round(nfr * (fs * 8 + nbytes) / ms / 1e6 / 8000, 3)
        row[f"D{D}_dispatch"] = ctx.last_dispatch()
        del pay, fo
    print(json.dumps(row), flush=True)
    ctx.close(); torch.cuda.empty_cache()
