"""Frame-level encode, N = 128 .. 2048: the two-pass kernel (every symbol built twice) against the build-once scheme (tuning
txframe_rewrite = 1: unnormalised samples out, then a rescale sweep over what was just written).  Same process, same box, same
inputs; the outputs must be bit-identical.  python tools/enc_ab.py [log2_samples=27] [N ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api

total = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 27)
ns = [int(a) for a in sys.argv[2:]] or [128, 256, 512, 1024, 2048]
for n in ns:
    ctx = api.Context(n_fft=n, modulation=api.QAM64, guard_bands=True)
    nb = 16 * ctx.bytes_per_symbol - 16
    flen = ctx.frame_samples(nb)
    nfr = total // flen
    g = torch.Generator(device="cuda"); g.manual_seed(n)
    pay = torch.randint(0, 256, (nfr, nb), dtype=torch.uint8, device="cuda", generator=g)
    outs, row = [], {"n_fft": n, "frames": nfr}
    for rw in (0, 1, 0, 1):
        ctx.set_tuning("txframe_rewrite", rw)
        out = ctx.encode_batch(pay)
        torch.cuda.synchronize()
        ctx.timer_start()
        for _ in range(5):
            ctx.encode_batch(pay, out=out)
        ms = ctx.timer_stop_ms() / 5
        row.setdefault(f"rewrite{rw}_ms", []).append(round(ms, 4))
        row[f"rewrite{rw}_dispatch"] = ctx.last_dispatch()
        if len(outs) < 2:
            outs.append(out.clone())
        del out
    row["bit_identical"] = bool(torch.equal(torch.view_as_real(outs[0]), torch.view_as_real(outs[1])))
    b = nfr * (flen * 8 + nb)
    row["frac0"] = round(b / (min(row["rewrite0_ms"]) / 1e3) / 8e12, 3)
    row["frac1"] = round(b / (min(row["rewrite1_ms"]) / 1e3) / 8e12, 3)
    print(json.dumps(row), flush=True)
    del outs, pay
    torch.cuda.empty_cache()
