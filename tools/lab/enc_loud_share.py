import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ofdm_amd import api
for n, mod, guard, nbytes in ((64,1,False,2000),(64,1,True,2000),(64,2,False,2000),(64,6,True,4000),(256,1,False,2000),(256,1,True,2000)):
    ctx = api.Context(n_fft=n, modulation=mod, guard_bands=guard)
    g = torch.Generator(device="cuda"); g.manual_seed(n)
    fs = ctx.frame_samples(nbytes); nfr = (1 << 26) // fs
    pay = torch.randint(0, 256, (nfr, nbytes), dtype=torch.uint8, device="cuda", generator=g)
    fo = ctx.encode_batch(pay)
    row = {"n": n, "mod": mod, "guard": guard, "D": ctx.data_symbols(nbytes), "frames": nfr}
    for key in (0, 1):
        ctx.set_tuning("no_txframe_optimistic", key)
        ctx.encode_batch(pay, out=fo); torch.cuda.synchronize(); ctx.timer_start()
        for _ in range(5): ctx.encode_batch(pay, out=fo)
        ms = ctx.timer_stop_ms() / 5
        row["once_GBs" if key == 0 else "twice_GBs"] = round(nfr * fs * 8 / ms / 1e6, 1)
    mx = torch.view_as_real(fo[:, :10 * ctx.S]).amax(dim=(1, 2))
    row["frames_louder_than_header"] = int((mx < 0.999).sum())
    row["dispatch"] = ctx.last_dispatch()
    print(json.dumps(row), flush=True)
