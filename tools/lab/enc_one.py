import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ofdm_amd import api
n = int(sys.argv[1]); key = int(sys.argv[2])
ctx = api.Context(n_fft=n, modulation=api.QAM64, guard_bands=True, tuning={"no_txframe_optimistic": key})
g = torch.Generator(device="cuda"); g.manual_seed(n)
nbytes = 16 * ctx.bytes_per_symbol - 16
fs = ctx.frame_samples(nbytes); nfr = (1 << 28) // fs
pay = torch.randint(0, 256, (nfr, nbytes), dtype=torch.uint8, device="cuda", generator=g)
fo = ctx.encode_batch(pay)
for _ in range(5): ctx.encode_batch(pay, out=fo)
torch.cuda.synchronize()
print(ctx.last_dispatch(), nfr)
