"""One-off / on-demand stress of ofdm_rx_decode_batch against the oracle's decode_sc with the parity rules of
tests/test_gpu_parity.py (status / offset / CFO exact, bytes exact unless the ORACLE's soft value sits within 1e-5 of a
decision boundary).  python tools/decode_stress.py [frames] [seed]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_gpu_parity as T
from ofdm_amd import api
from oracle import oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tot = 0
for (nf, mod, guard, ecc, nbytes, snr) in ((64, 6, True, 0, 560, 30.0), (64, 6, True, 0, 560, 22.0), (64, 2, False, 0, 400, 15.0),
                                            (64, 4, True, 1, 300, 25.0), (64, 8, True, 0, 500, 35.0), (64, 1, True, 0, 90, 8.0)):
    exc, r, _ = T.run_decode_parity(api, orc, nf, mod, guard, ecc, nbytes, n, 96, seed=seed, snr_db=snr)
    ok = int((r["status"] == 0).sum())
    print(f"N={nf} mod={mod} guard={guard} ecc={ecc} snr={snr}: {n} frames, {ok} decoded, {exc} boundary excuses, all parity checks passed")
    tot += exc
print("total excuses", tot)
