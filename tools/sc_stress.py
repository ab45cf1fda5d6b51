"""Randomised stress of ofdm_sc_correlate_batch against the f64 oracle (test infrastructure): thousands of captures in
different regimes (SNR from -3 to 40 dB, any delay, large CFO, truncated frames, two packets, interferers).  Prints the
number of frames whose timing index / CFO / metric differ.  Run on the GPU box: python tools/sc_stress.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from ofdm_amd import api
from oracle import oracle as orc
from util import fc32, wide

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
span = 2176
taps = orc.channel_taps()
caps = []
for f in range(n):
    mod = [orc.BPSK, orc.QPSK, orc.QAM16, orc.QAM64][f % 4]
    nbytes = int(rng.integers(1, 140 if mod == orc.BPSK else 560))
    tx = orc.encode(bytes(rng.integers(0, 256, nbytes, dtype=np.uint8)), guard=bool(f & 1), modulation=mod)
    if tx.size > span - 100:
        tx = tx[: span - 100]
    y = np.convolve(tx, taps)[: tx.size + 24] if f % 3 else tx
    delay = int(rng.integers(0, max(1, span - min(y.size, 900))))
    buf = np.zeros(span, complex)
    m = min(span - delay, y.size)
    buf[delay:delay + m] = y[:m]
    if f % 11 == 0:  # a second, weaker packet earlier in the slot
        d2 = int(rng.integers(0, max(1, delay - 400))) if delay > 450 else 0
        buf[d2:d2 + min(400, span - d2)] += 0.4 * y[: min(400, span - d2)]
    buf *= np.exp(1j * float(rng.uniform(-1, 1)) * np.pi / 80 * np.arange(1, span + 1))
    p = np.mean(np.abs(y[800:]) ** 2) if y.size > 900 else np.mean(np.abs(y) ** 2)
    snr = float(rng.choice([-3, 0, 3, 6, 10, 20, 30, 40]))
    sig = np.sqrt(p / 10 ** (snr / 10) / 2)
    buf += sig * (rng.standard_normal(span) + 1j * rng.standard_normal(span))
    if f % 13 == 0:  # narrow-band interferer
        buf += 0.5 * np.sqrt(p) * np.exp(1j * 2 * np.pi * float(rng.uniform(0, 0.5)) * np.arange(span))
    caps.append(fc32(buf))
caps = np.stack(caps)
# detector: k_sc80 (exact streaming detector, the default) or k_sc_cf (the f32 filter pair, tuning no_sc80)
detector = sys.argv[3] if len(sys.argv) > 3 else "k_sc80"
ctx = api.Context(modulation=api.QAM64, guard_bands=True, tuning={"no_sc80": 1} if detector == "k_sc_cf" else {})
bad = 0
for n_lags, flen in ((0, span), (256, span), (0, 2000), (700, 1800)):
    d_hat, f_delta, metric = (t.cpu().numpy() for t in ctx.sc_correlate(ctx.to_device(caps), frame_len=flen, n_lags=n_lags))
    assert detector in ctx.last_dispatch(), ctx.last_dispatch()
    nb = nf = 0
    for f in range(n):
        wd, _, wm, wfd = orc.sc_sync(wide(caps[f][:flen]), 80, 3, n_lags, 0.5)
        if d_hat[f] != wd or (wd >= 0 and (abs(f_delta[f] - wfd) > 1e-9 or abs(metric[f] - wm) > 1e-6)):
            nb += 1
            if nb <= 5:
                print("MISMATCH frame", f, "n_lags", n_lags, "flen", flen, "gpu", d_hat[f], f_delta[f], metric[f], "oracle", wd, wfd, wm)
        nf += wd >= 0
    print(f"n_lags={n_lags} frame_len={flen}: {n} frames, {nf} with a packet, {nb} mismatches")
    bad += nb
sys.exit(1 if bad else 0)
