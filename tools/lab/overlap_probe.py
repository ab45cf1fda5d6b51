"""Can a store-only kernel hide inside the (compute-bound) frame encoder?  encode_batch on one stream, a fill of the header blocks' volume
(10 of 26 symbols) on another; wall time of the pair against each alone.   python tools/lab/overlap_probe.py [N]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = api.Context(n_fft=n, modulation=api.QAM64, guard_bands=True)
g = torch.Generator(device="cuda"); g.manual_seed(n)
nbytes = 16 * ctx.bytes_per_symbol - 16
fs = ctx.frame_samples(nbytes); nfr = (1 << 28) // fs
pay = torch.randint(0, 256, (nfr, nbytes), dtype=torch.uint8, device="cuda", generator=g)
fo = ctx.encode_batch(pay)
hdr = torch.empty((nfr * 10 * ctx.S,), dtype=torch.complex64, device="cuda")     # the headers' volume
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ctx._ck(ctx.lib.ofdm_set_stream(ctx.h, C.c_void_p(sa.cuda_stream)), "set_stream")
def wall(fn, reps=8):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
def enc():
    ctx.encode_batch(pay, out=fo)
def fill():
    with torch.cuda.stream(sb): hdr.fill_(1.0)
def both():
    fill(); enc()
print(json.dumps({"n_fft": n, "frames": nfr, "encode_ms": round(wall(enc), 4), "fill_ms": round(wall(fill), 4), "both_ms": round(wall(both), 4)}))
