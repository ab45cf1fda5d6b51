"""Full decode chain (Schmidl-Cox timing over every lag, CFO, channel estimate, demod, header) for transform lengths other than the
two BASELINE frame shapes: frames from the library's TX through its GPU channel, decode_batch timed, payloads checked.
  python tools/bench_decode_n.py [N ...]      (OFDM_TUNE=no_mid_kernels=1: the generic demodulator instead of k_demod_mid<FRAME>)"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api
from tools import tune_env
tune_env.install()

for n in [int(a) for a in sys.argv[1:]] or [128, 512, 2048]:
    ctx = api.Context(n_fft=n, modulation=api.QAM64, guard_bands=True)
    D = 16
    nbytes = D * ctx.bytes_per_symbol - 16
    fs = ctx.frame_samples(nbytes)
    span = (fs + 2 * ctx.S) // 2 * 2
    nfr = max(8, (1 << 27) // span)
    g = torch.Generator(device=ctx.device); g.manual_seed(n)
    pay = torch.randint(0, 256, (nfr, nbytes), dtype=torch.uint8, device=ctx.device, generator=g)
    tx = ctx.encode_batch(pay)
    d = torch.randint(1, ctx.S, (nfr,), device=ctx.device, generator=g, dtype=torch.int32)
    fd = (torch.rand((nfr,), device=ctx.device, generator=g, dtype=torch.float64) * 1.8 - 0.9) * math.pi / ctx.S
    x = ctx.channel_batch(tx, snr_db=40.0, seed=n, delay=d, f_delta=fd, span=span)
    del tx
    r = ctx.decode_batch(x, max_symbols=D)
    torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(3):
        r = ctx.decode_batch(x, max_symbols=D)
    ms = ctx.timer_stop_ms() / 3
    ok = (r["status"] == 0) & (r["len"] == nbytes)
    exact = int(((r["bytes"][:, :nbytes] == pay).all(dim=1) & ok).sum())
    print(json.dumps({"n_fft": n, "frames": nfr, "span": span, "ms": ms, "gsamples_per_s": nfr * span / ms / 1e6,
                      "frac_of_one_read": nfr * (span * 8 + nbytes) / (ms / 1e3) / 8e12, "decoded": int(ok.sum()), "byte_exact": exact}), flush=True)
    del x, r
    torch.cuda.empty_cache()
