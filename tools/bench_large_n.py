"""Configs 4 and 5 on ONE GPU (the multi-GPU versions are frame-index replicas of these):
   cfg4: N=1024, 64-QAM + Hamming(7,4), full RX chain on a resident ring of frames;
   cfg5: N=4096, 256-QAM, TX (map + IFFT + CP) then RX (CP strip + FFT + demap) back to back, continuous symbols."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api


def cfg5(n_sym=65536, steps=5):
    ctx = api.Context(n_fft=4096, modulation=api.QAM256, guard_bands=True)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    nb = n_sym * ctx.bytes_per_symbol
    pay = torch.randint(0, 256, (nb,), dtype=torch.uint8, device="cuda", generator=g)
    def tx():
        pts = ctx.modulate(pay)
        bins = ctx.encode_block(pts.view(-1, ctx.data_carriers))
        return ctx.prefix_block(bins)
    x = tx(); torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(steps): x = tx()
    tx_staged_ms = ctx.timer_stop_ms() / steps
    xf = ctx.tx_symbols(pay); torch.cuda.synchronize()          # the same three stages in one pass
    same = float((xf.view(-1) - x.view(-1)).abs().max() / x.view(-1).abs().max())  # 64 x 64 kernel vs staged radix-8 passes
    ctx.timer_start()
    for _ in range(steps): xf = ctx.tx_symbols(pay)
    tx_ms = ctx.timer_stop_ms() / steps
    del xf
    out = ctx.rx_demod(x.view(1, -1), syms_per_frame=n_sym)
    torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(steps): out = ctx.rx_demod(x.view(1, -1), syms_per_frame=n_sym)
    rx_ms = ctx.timer_stop_ms() / steps
    ok = bool((out.view(-1) == pay).all())
    ns = n_sym * ctx.S
    return {"workload": "cfg5: N=4096 256QAM guard, continuous symbols", "symbols": n_sym, "tx_ms": tx_ms, "rx_ms": rx_ms,
            "tx_msamples_per_s": ns / tx_ms / 1e3, "tx_staged_ms": tx_staged_ms, "tx_fused_vs_staged_max_rel_err": same,
            "rx_msamples_per_s": ns / rx_ms / 1e3,
            "rx_hbm_frac": (ns * 8 + nb) / (rx_ms / 1e3) / 8e12, "rx_bytes_equal_tx_payload": ok}


def cfg4(n_frames=8192, steps=5):
    ctx = api.Context(n_fft=1024, modulation=api.QAM64, guard_bands=True, ecc=api.ECC_HAMMING74)
    nbytes = 1536
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    pay = torch.randint(0, 256, (n_frames, nbytes), dtype=torch.uint8, device="cuda", generator=g)
    frames = ctx.encode_batch(pay)                     # [F, (10 + D) * 1280]
    D = ctx.data_symbols(nbytes)
    span = frames.shape[1] + 256
    x = torch.zeros((n_frames, span), dtype=torch.complex64, device="cuda")
    x[:, 100:100 + frames.shape[1]] = frames
    x += torch.view_as_complex(torch.randn((n_frames, span, 2), device="cuda", generator=g) * 1e-4)
    res = ctx.decode_batch(x, max_symbols=D, n_lags=2048)
    torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(steps): res = ctx.decode_batch(x, max_symbols=D, n_lags=2048)
    ms = ctx.timer_stop_ms() / steps
    ok = (res["status"] == 0) & (res["len"] >= nbytes)
    good = int(((res["bytes"][:, :nbytes] == pay).all(dim=1) & ok).sum())
    return {"workload": "cfg4: N=1024 64QAM + Hamming(7,4), full RX chain, frames of %d samples" % frames.shape[1],
            "frames": n_frames, "data_symbols": D, "chain_ms": ms, "msamples_per_s": n_frames * span / ms / 1e3,
            "hbm_frac_of_one_read": n_frames * span * 8 / (ms / 1e3) / 8e12, "frames_decoded_exactly": good}


if __name__ == "__main__":
    out = {}
    for name, fn in (("cfg5", cfg5), ("cfg4", cfg4)):
        try:
            out[name] = fn()
        except Exception as e:
            out[name] = {"error": repr(e)}
    print(json.dumps(out))
