"""On-demand randomised stress of the shape-specialised symbol kernels (kernels_mid.hip, k_demod4096 / k_tx4096 / k_txframe4096,
k_demod64) against the oracle: random transform length, modulation, guard setting, symbols per frame, frame count, channel mode
(none / shared / per frame), first_symbol, and -- in frame mode -- per-frame offsets, CFO and a capture cut short; TX streams that
end inside a symbol; encode with ragged payload lengths.  Bytes exact unless the ORACLE's own soft value of the differing point
sits within 1e-5 of a decision boundary; samples within 1e-5.      python tools/shape_stress.py [trials] [seed]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import torch
from util import assert_bytes_match, fc32, make_symbols_np, rel_err, wide
from ofdm_amd import api
from oracle import oracle as orc

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
TOL = 1e-5
fails = excused = 0
LENGTHS = (64, 128, 256, 512, 1024, 2048, 4096)
for trial in range(trials):
    n = int(rng.choice(LENGTHS)); mod = int(rng.choice((1, 2, 4, 6, 8))); guard = bool(rng.integers(0, 2))
    S = n + n // 4
    ctx = api.Context(n_fft=n, modulation=mod, guard_bands=guard, tuning={"grid_cap": int(rng.integers(1, 6))})  # few workgroups: many steps each
    dev = lambda a: ctx.to_device(a)
    kind = int(rng.integers(0, 4))
    what = f"trial {trial}: N={n} mod={mod} guard={guard} kind={kind}"
    try:
        if kind == 0:      # RX stream: k symbols per frame, channel mode, first_symbol
            k = int(rng.integers(1, 13)) * (8 if n == 64 else 1)
            nf = int(rng.integers(1, max(2, 40 * 64 // n)))
            first = int(rng.integers(0, 3)) * (8 if n == 64 else 1)
            x, _ = make_symbols_np(orc, rng, (k + first) * nf, n, guard, mod, snr_db=36.0)
            xs = x.reshape(nf, (k + first) * S)
            hmode = int(rng.integers(0, 3)) if n != 64 else int(rng.integers(0, 2))
            hks = fc32(1.0 + 0.2 * (rng.standard_normal((nf, n)) + 1j * rng.standard_normal((nf, n))))
            hk = None if hmode == 0 else (hks[0] if hmode == 1 else hks)
            out = ctx.rx_demod(dev(xs), k, first_symbol=first, hk=None if hk is None else dev(hk)).cpu().numpy()
            for f in range(nf):
                h = None if hmode == 0 else wide(hks[0] if hmode == 1 else hks[f])
                want, soft = orc.rx_demod(wide(xs[f, first * S:]), n, guard, mod, hk=h, want_soft=True)
                excused += assert_bytes_match(bytes(out[f]), want, soft, mod, what=what + f" frame {f}")
        elif kind == 1:    # RX frame mode: offsets, CFO, per-frame channel, short capture
            k, nf = int(rng.integers(1, 9)), int(rng.integers(1, 7))
            x, _ = make_symbols_np(orc, rng, k * nf, n, guard, mod, snr_db=36.0)
            x = wide(x).reshape(nf, k * S)
            offs = rng.integers(0, 90, nf).astype(np.int32)
            fds = (rng.random(nf) * 1.8 - 0.9) * np.pi / S
            span = k * S + 96
            frames = np.zeros((nf, span), np.complex128)
            for f in range(nf):
                frames[f, offs[f]:offs[f] + k * S] = x[f] * np.exp(1j * fds[f] * (np.arange(k * S) + 1))
            hks = fc32(1.0 + 0.2 * (rng.standard_normal((nf, n)) + 1j * rng.standard_normal((nf, n))))
            cut = span - int(rng.integers(0, S))
            out = ctx.rx_demod(dev(fc32(frames)), k, offset=torch.from_numpy(offs).to(ctx.device),
                               f_delta=torch.from_numpy(fds).to(ctx.device), hk=dev(hks), frame_len=cut).cpu().numpy()
            for f in range(nf):
                seg = wide(fc32(frames[f]))[:cut][offs[f]:]
                seg = np.concatenate([seg, np.zeros(max(0, k * S - seg.size), np.complex128)])[: k * S]
                want, soft = orc.rx_demod(orc.cfo_rotate(seg, fds[f], 0), n, guard, mod, hk=wide(hks[f]), want_soft=True)
                excused += assert_bytes_match(bytes(out[f]), want, soft, mod, what=what + f" frame {f}")
        elif kind == 2:    # TX stream ending inside a symbol
            nd, bps = ctx.data_carriers, ctx.bytes_per_symbol
            n_sym = int(rng.integers(1, max(2, 60 * 64 // n)))
            nb = int(rng.integers(0, n_sym * bps + 1))
            data = rng.integers(0, 256, nb, dtype=np.uint8)
            got = ctx.tx_symbols(torch.from_numpy(data.copy()).to(ctx.device), n_sym=n_sym).cpu().numpy()
            pts = np.zeros(n_sym * nd, np.complex128)
            o = np.asarray(orc.modulate(bytes(data), mod))
            pts[: o.size] = o
            want = np.stack([orc.prefix_block(orc.encode_block(pts[i * nd:(i + 1) * nd], n, guard)[0]) for i in range(n_sym)])
            assert rel_err(got, want) < TOL, what
            for i in range(n_sym):
                assert rel_err(got[i], want[i]) < 4 * TOL, what + f" symbol {i}"
        else:              # encode with ragged lengths
            nbytes = int(rng.integers(0, 6 * ctx.bytes_per_symbol))
            nf = int(rng.integers(1, 8))
            lens = rng.integers(0, nbytes + 1, nf).astype(np.int32)
            pay = rng.integers(0, 256, (nf, max(nbytes, 1)), dtype=np.uint8)[:, :nbytes]
            got = ctx.encode_batch(torch.from_numpy(np.ascontiguousarray(pay)).to(ctx.device).reshape(nf, nbytes),
                                   lens=torch.from_numpy(lens)).cpu().numpy()
            for f in range(nf):
                want = orc.encode(bytes(pay[f, :lens[f]]), guard, mod, n)
                assert rel_err(got[f, :want.size], want) <= TOL, what + f" frame {f} (len {lens[f]})"
    except AssertionError as e:
        fails += 1
        print("FAIL", what, str(e)[:200], flush=True)
    ctx.close()
print(f"{trials} trials: parity failures {fails}, excused boundary decisions {excused}")
sys.exit(1 if fails else 0)
