"""Config-3 report for bench.py: full RX chain (Schmidl-Cox -> CFO -> channel estimate -> FFT -> equalise -> demap ->
length header) on synthetic frames with a random integer delay, CFO and the reference FIR channel, plus the
stand-alone Schmidl-Cox kernel's HBM roofline fraction (north-star target: >= 40 %)."""
import math

CHANNEL_TAPS = [-0.1912, 0.9316, 0.2821, -0.1990, 0.1630, -0.1017, 0.0544, -0.0261, 0.0090, 0.0000, -0.0034]  # channel.rs:26-31, taps 8..18
FIRST_TAP = 8
HBM_PEAK_GBS = 8000.0


def synth(api, torch, ctx, n_frames, span, snr_db=30.0, seed=3):
    g = torch.Generator(device=ctx.device)
    g.manual_seed(seed)
    nbytes = 560
    flen = ctx.frame_samples(nbytes)  # 2080
    x = torch.zeros((n_frames, span), dtype=torch.complex64, device=ctx.device)
    payload = torch.randint(0, 256, (n_frames, nbytes), dtype=torch.uint8, device=ctx.device, generator=g)
    taps = torch.tensor(CHANNEL_TAPS, dtype=torch.float32, device=ctx.device).flip(0).view(1, 1, -1)
    n = torch.arange(span, device=ctx.device)
    chunk = 32768
    for lo in range(0, n_frames, chunk):
        hi = min(lo + chunk, n_frames)
        m = hi - lo
        tx = ctx.encode_batch(payload[lo:hi].contiguous())  # [m, 2080]
        ri = torch.view_as_real(tx).permute(0, 2, 1).reshape(2 * m, 1, flen)
        y = torch.nn.functional.conv1d(ri, taps, padding=len(CHANNEL_TAPS) - 1)  # full convolution
        y = y.view(m, 2, -1)
        ylen = y.shape[-1]
        yc = torch.complex(y[:, 0], y[:, 1])
        d = torch.randint(1, 65, (m, 1), device=ctx.device, generator=g)
        src = n.view(1, -1) - d - FIRST_TAP
        ok = (src >= 0) & (src < ylen)
        buf = torch.gather(yc, 1, src.clamp(0, ylen - 1)) * ok
        fd = (torch.rand((m, 1), device=ctx.device, generator=g, dtype=torch.float64) * 1.9 - 0.95) * math.pi / ctx.S
        ph = fd * (n.view(1, -1) + 1).to(torch.float64)
        rot = torch.polar(torch.ones_like(ph), ph).to(torch.complex64)
        p = float((yc.real ** 2 + yc.imag ** 2).mean())
        sigma = (p / 10 ** (snr_db / 10) / 2) ** 0.5
        noise = torch.randn((m, span, 2), device=ctx.device, generator=g) * sigma
        x[lo:hi] = buf * rot + torch.view_as_complex(noise)
        del tx, ri, y, yc, buf, ph, rot, noise
    torch.cuda.synchronize()
    return x, payload


SYNC_LAGS = 256  # frames start within the first 64 samples of their slot: d_hat <= 64 + 80 + 9, so 256 lags cover it


def run(api, torch, n_frames, steps, device):
    ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True, device=device)
    span = 2176
    x, payload = synth(api, torch, ctx, n_frames, span)
    D = ctx.data_symbols(560)
    # --- full chain, timing search bounded to the slot's possible frame starts
    res = ctx.decode_batch(x, max_symbols=D, n_lags=SYNC_LAGS)
    torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(steps):
        res = ctx.decode_batch(x, max_symbols=D, n_lags=SYNC_LAGS)
    ms = ctx.timer_stop_ms() / steps
    # the same chain searching every lag of the slot (1857 lags)
    full = ctx.decode_batch(x, max_symbols=D)
    both = (full["status"] == 0) & (res["status"] == 0)
    differ = int(((full["offset"] != res["offset"]) | (full["len"] != res["len"])
                  | (full["bytes"][:, :560] != res["bytes"][:, :560]).any(dim=1))[both].sum())
    status_differ = int((full["status"] != res["status"]).sum())
    ctx.timer_start()
    for _ in range(steps):
        ctx.decode_batch(x, max_symbols=D)
    ms_all = ctx.timer_stop_ms() / steps
    ok = (res["status"] == 0) & (res["len"] == 560)
    nok = int(ok.sum())
    diff = torch.bitwise_xor(res["bytes"][:, :560], payload)[ok]
    bits = sum(int(((diff >> sh) & 1).sum()) for sh in range(8))
    chain_bytes = n_frames * (span * 8 + 560)
    out = {
        "workload": "cfg3: 2080-sample 64QAM frames at stride 2176, delay 1..64, CFO +-0.95 pi/80, FIR channel, 30 dB",
        "frames": n_frames, "sync_lags": SYNC_LAGS, "full_chain_ms": ms, "full_chain_msamples_per_s": n_frames * span / ms / 1e3,
        "full_chain_all_lags_ms": ms_all, "full_chain_all_lags_msamples_per_s": n_frames * span / ms_all / 1e3,
        "bounded_vs_full_search": {"frames_ok_in_both_but_different": differ, "frames_with_different_status": status_differ},
        "full_chain_hbm_frac_of_one_read": chain_bytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS,
        "frames_decoded": nok, "ber_decoded_frames": bits / max(1, nok * 560 * 8),
    }
    # --- TX side of the hot path: encode (modulate + encode_block + IFFT + CP + header + normalise) for the same payloads
    npay = min(n_frames, 131072)
    txo = ctx.encode_batch(payload[:npay])
    torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(steps):
        ctx.encode_batch(payload[:npay], out=txo)
    tms = ctx.timer_stop_ms() / steps
    out["tx_encode"] = {"kernel": "k_txframe64<6, true>", "frames": npay, "ms": tms, "msamples_per_s": npay * txo.shape[-1] / tms / 1e3,
                        "hbm_frac_of_one_write": npay * txo.shape[-1] * 8 / (tms / 1e3) / 1e9 / HBM_PEAK_GBS}
    del txo
    # --- Schmidl-Cox kernel alone
    ctx.sc_correlate(x)
    torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(steps):
        ctx.sc_correlate(x)
    sms = ctx.timer_stop_ms() / steps
    sc_bytes = n_frames * (span * 8 + 16)
    out["schmidl_cox"] = {"kernel": "k_sc_cf<256, 2> + k_sc_post (all 1857 lags of every 2176-sample slot)", "kernel_ms": sms, "msamples_per_s": n_frames * span / sms / 1e3,
                          "roofline": {"bound": "hbm", "achieved": sc_bytes / (sms / 1e3) / 1e9, "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "frac": sc_bytes / (sms / 1e3) / 1e9 / HBM_PEAK_GBS}}
    return out
