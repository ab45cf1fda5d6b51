import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import oracle as orc
from ofdm_amd import api
from util import through_channel, wide
rng = np.random.default_rng(5)
caps = []
for f in range(48):
    d = int(rng.integers(1, 65)); fd = (rng.random() * 1.9 - 0.95) * np.pi / 80
    tx = orc.encode(bytes(rng.integers(0, 256, 560, dtype=np.uint8)), True, orc.QAM64)
    caps.append(through_channel(orc, rng, tx, 2176, d, fd, 30.0))
caps = np.stack(caps)
ctx = api.Context(modulation=api.QAM64, guard_bands=True)
for nl in (256, 400, 900, 0):
    d2, fd2, m2 = ctx.sc_correlate(ctx.to_device(caps), frame_len=2000, n_lags=nl)
    d2 = d2.cpu().numpy()
    want = np.array([orc.sc_sync(wide(caps[f][:2000]), 80, 3, nl, 0.5)[0] for f in range(48)])
    print(os.environ.get("OFDM_SC_MIN_WG"), "n_lags", nl, "mismatch", int((d2 != want).sum()), "of 48", d2[:8], want[:8])
