"""Perf-cliff audit of the decode / encode / demod entry points: many (N, modulation, payload, stride parity, outer code) shapes at a fixed
sample budget, one line each with the dispatch string and the rate in GB/s of capture bytes -- a shape that falls off a fast path shows
up as a generic kernel name and a rate several times below its neighbours.   python tools/lab/cliff_audit.py [log2 samples]"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import api

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 25
g = torch.Generator(device="cuda"); g.manual_seed(1)
rows = []
for n in (64, 256, 1024, 4096):
    for mod in (api.BPSK, api.QPSK, api.QAM16, api.QAM64, api.QAM256):
        for guard in (True, False):
            for ecc in (0, 1):
                for nbytes in (36, 300, 2000):
                    for odd in (0, 1):
                        if (ecc and (mod != api.QAM64 or not guard)) or (odd and (mod != api.QAM64 or not guard or ecc)): continue
                        if not guard and mod not in (api.BPSK, api.QAM64): continue
                        ctx = api.Context(n_fft=n, modulation=mod, guard_bands=guard, ecc=api.ECC_HAMMING74 if ecc else api.ECC_NONE)
                        S = ctx.S
                        fs = ctx.frame_samples(nbytes)
                        D = ctx.data_symbols(nbytes)
                        span = fs + S + 16 + odd
                        nfr = max(8, (1 << lg) // span)
                        pay = torch.randint(0, 256, (nfr, nbytes), dtype=torch.uint8, device="cuda", generator=g)
                        def timed(fn, reps=3):
                            fn(); torch.cuda.synchronize(); ctx.timer_start()
                            for _ in range(reps): fn()
                            return ctx.timer_stop_ms() / reps
                        tx = ctx.encode_batch(pay)
                        enc_ms = timed(lambda: ctx.encode_batch(pay, out=tx)); enc_d = ctx.last_dispatch()
                        d = torch.randint(1, S, (nfr,), device="cuda", generator=g, dtype=torch.int32)
                        fd = (torch.rand((nfr,), device="cuda", generator=g, dtype=torch.float64) * 1.9 - 0.95) * math.pi / S
                        x = ctx.channel_batch(tx, snr_db=40.0, seed=3, delay=d, f_delta=fd, span=span)
                        r = ctx.decode_batch(x, max_symbols=D)
                        dec_ms = timed(lambda: ctx.decode_batch(x, max_symbols=D)); dec_d = ctx.last_dispatch()
                        okf = float((r["len"] == nbytes).float().mean())
                        row = {"n": n, "mod": mod, "guard": guard, "ecc": ecc, "nbytes": nbytes, "odd_stride": odd, "frames": nfr, "D": D,
                               "decode_GBs": round(nfr * span * 8 / dec_ms / 1e6, 1), "encode_GBs": round(nfr * fs * 8 / enc_ms / 1e6, 1),
                               "decoded_ok": round(okf, 3), "decode": dec_d, "encode": enc_d}
                        print(json.dumps(row), flush=True)
                        ctx.close(); del pay, tx, x, r
                        torch.cuda.empty_cache()
