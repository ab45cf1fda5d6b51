"""Symbol-stream TX (tx_symbols: modulate + encode_block + IFFT + CP) and RX (rx_demod: unprefix + FFT + pilot phase + demap)
for every transform length the library accepts, each against its one-pass HBM roofline.  Shows which lengths run on a
shape-specialised kernel and what the generic k_sym<N> path costs for the others.
  python tools/bench_shapes.py [total_samples_log2=26] [steps=5] [modulation=6] [N ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ofdm_amd as api

HBM_PEAK_GBS = 8000.0


def one(n_fft, mod, total, steps, k, rx=True):
    ctx = api.Context(n_fft=n_fft, modulation=mod, guard_bands=True, device=0)
    S = ctx.S
    n_sym = (total // S) // k * k
    nb = n_sym * ctx.bytes_per_symbol
    g = torch.Generator(device=ctx.device)
    g.manual_seed(n_fft)
    data = torch.randint(0, 256, (nb,), dtype=torch.uint8, device=ctx.device, generator=g)
    x = ctx.tx_symbols(data, n_sym)
    torch.cuda.synchronize()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ctx.timer_start()
        for _ in range(steps):
            fn()
        return ctx.timer_stop_ms() / steps

    tx_ms = timed(lambda: ctx.tx_symbols(data, n_sym, out=x))
    bytes_ = n_sym * (S * 8 + ctx.bytes_per_symbol)
    r = {"n_fft": n_fft, "modulation": mod, "symbols": n_sym, "symbols_per_frame": k,
         "tx_ms": tx_ms, "tx_gsamples_per_s": n_sym * S / tx_ms / 1e6, "tx_frac": bytes_ / (tx_ms / 1e3) / 1e9 / HBM_PEAK_GBS}
    if rx:
        frames = x.view(n_sym // k, k * S)
        out = ctx.rx_demod(frames, k)
        rx_ms = timed(lambda: ctx.rx_demod(frames, k, out=out))
        r.update({"rx_ms": rx_ms, "rx_gsamples_per_s": n_sym * S / rx_ms / 1e6, "rx_frac": bytes_ / (rx_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                  "rx_bytes_equal_tx_payload": bool((out.reshape(-1) == data).all())})
        del out
    del x
    # frame-level TX (encode: header blocks + D data symbols, normalised per frame) and the staged chain behind OFDM_TUNE=no_mid_kernels=1
    D = 16
    nbytes = D * ctx.bytes_per_symbol - 16
    fs = ctx.frame_samples(nbytes)
    nfr = max(1, total // fs)
    pay = torch.randint(0, 256, (nfr, nbytes), dtype=torch.uint8, device=ctx.device, generator=g)
    fo = ctx.encode_batch(pay)
    enc_ms = timed(lambda: ctx.encode_batch(pay, out=fo))
    eb = nfr * (fs * 8 + nbytes)
    r.update({"encode_frames": nfr, "encode_data_symbols": D, "encode_ms": enc_ms, "encode_gsamples_per_s": nfr * fs / enc_ms / 1e6,
              "encode_frac": eb / (enc_ms / 1e3) / 1e9 / HBM_PEAK_GBS})
    ctx.close()
    return r


if __name__ == "__main__":
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    mod = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    ns = [int(a) for a in sys.argv[4:]] or [64, 128, 256, 512, 1024, 2048, 4096]
    for n in ns:
        print(json.dumps(one(n, mod, 1 << lg, steps, 8)), flush=True)
        torch.cuda.empty_cache()
