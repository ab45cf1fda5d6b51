"""Config 3 (BASELINE.json configs[2]): 1 M full frames with the Schmidl-Cox preamble, the reference's FIR channel, a random
integer delay, a random signed CFO and 30 dB noise; full RX chain (timing -> CFO -> channel estimate -> FFT -> equalise ->
pilot phase -> demap -> length header) on one GPU.  Reports
  * the chain searching EVERY lag of every slot (the headline of this block) and bounded to the slot's 256 possible lags,
    staged (two HBM passes) and through the one-pass kernel (ofdm_params.rx_path = OFDM_RX_ONE_PASS),
  * Schmidl-Cox alone against the HBM roofline (north-star target >= 40 %): the kernel that computes every lag, and the
    product's two-launch search (first lags decide, the rest of the slot is read only for frames they do not determine),
  * the TX side (encode) for the same payloads,
  * a CPU baseline (the oracle's decode_sc on a bounded sample, all host cores) and GPU-vs-CPU equality on that sample.
Inputs come from the library's own TX (encode_batch) and its GPU channel model (channel_batch, src/channel.rs:33-74)."""
import math
import os

import numpy as np

HBM_PEAK_GBS = 8000.0
SYNC_LAGS = 256  # frames start within the first 64 samples of their slot: d_hat <= 64 + 80 + 9, so 256 lags cover it
SPAN = 2176      # slot: 2080-sample frame + delay <= 64 + channel tail, a multiple of 256 B
NBYTES = 560     # 16 data symbols of 36 B minus the 16-byte length header


def synth(api, torch, ctx, n_frames, span=SPAN, snr_db=30.0, seed=3):
    """[n_frames, span] captures + payloads.  TX by the library, channel by the library's GPU restatement of
    src/channel.rs (FIR CHANNEL, CFO, noise) with the test-bench extensions: per-frame delay, signed CFO."""
    g = torch.Generator(device=ctx.device)
    g.manual_seed(seed)
    x = torch.empty((n_frames, span), dtype=torch.complex64, device=ctx.device)
    payload = torch.randint(0, 256, (n_frames, NBYTES), dtype=torch.uint8, device=ctx.device, generator=g)
    chunk = 65536
    for lo in range(0, n_frames, chunk):
        hi = min(lo + chunk, n_frames)
        tx = ctx.encode_batch(payload[lo:hi].contiguous())  # [m, 2080]
        d = torch.randint(1, 65, (hi - lo,), device=ctx.device, generator=g, dtype=torch.int32)
        fd = (torch.rand((hi - lo,), device=ctx.device, generator=g, dtype=torch.float64) * 1.9 - 0.95) * math.pi / ctx.S
        ctx.channel_batch(tx, snr_db=snr_db, seed=seed * 1_000_003 + lo, delay=d, f_delta=fd, out=x[lo:hi])
        del tx
    torch.cuda.synchronize()
    return x, payload


def _timed(ctx, torch, fn, steps, grp=None):
    from tools import rank_timing
    return rank_timing.timed(ctx, torch, fn, steps, grp)


def _ber(torch, res, payload):
    ok = (res["status"] == 0) & (res["len"] == NBYTES)
    nok = int(ok.sum())
    diff = torch.bitwise_xor(res["bytes"][:, :NBYTES], payload)[ok]
    bits = sum(int(((diff >> sh) & 1).sum()) for sh in range(8))
    return nok, bits / max(1, nok * NBYTES * 8)


def cpu_leg(x, payload, gpu, D, n_sample, threads):
    """oracle decode_sc (all lags) on the first n_sample frames, all cores; per-frame equality with the GPU's outputs."""
    from oracle import oracle as orc
    from tools import cpu_baseline as cb

    orc.lib()
    orc.set_fft_cache(True)
    xs = x[:n_sample].cpu().numpy()
    blocks = [(np.arange(i, n_sample, threads), None) for i in range(threads)]
    wide = [[xs[j].astype(np.complex128) for j in idx] for idx, _ in blocks]

    def work(frames):
        return [orc.decode_sc(f, True, orc.QAM64, 64, max_symbols=D) for f in frames]

    rec, outs = cb.timed(work, wide, n_sample * SPAN / 1e6, target_s=3.0)
    st = gpu["status"][:n_sample].cpu().numpy(); off = gpu["offset"][:n_sample].cpu().numpy()
    ln = gpu["len"][:n_sample].cpu().numpy(); by = gpu["bytes"][:n_sample].cpu().numpy()
    pay = payload[:n_sample].cpu().numpy()
    same = differ = sync_differ = 0
    cpu_bits = cpu_ok = 0
    for (idx, _), res in zip(blocks, outs):
        for j, w in zip(idx, res):
            if st[j] != w["status"] or (w["status"] == 0 and off[j] != w["offset"]):
                sync_differ += 1
                continue
            if w["status"] != 0:
                same += 1
                continue
            if bytes(by[j][: ln[j]]) == w["bytes"]:
                same += 1
            else:
                differ += 1
            if len(w["bytes"]) == NBYTES:
                cpu_ok += 1
                cpu_bits += int(np.unpackbits(np.frombuffer(w["bytes"], np.uint8) ^ pay[j]).sum())
    orc.set_fft_cache(False)
    return {"value": rec["value"], "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"first {n_sample} frames of the same batch x {rec['passes_over_sample']} passes, oracle decode_sc over all "
                      f"lags (f64, cached twiddles), {rec['seconds']:.1f} s wall",
            "frames_compared": n_sample, "frames_identical_to_gpu": same, "frames_with_a_differing_decision": differ,
            "frames_with_different_status_or_offset": sync_differ, "gpu_bytes_equal_cpu_bytes": differ == 0 and sync_differ == 0,
            "cpu_ber_on_sample": cpu_bits / max(1, cpu_ok * NBYTES * 8)}


def run(api, torch, n_frames, steps, device, cpu=True, grp=None):
    """grp (ofdm_amd.dist.Group, optional): every rank decodes its OWN n_frames captures (weak scaling, no data-path
    collective); times are the max over ranks, rates the aggregate over all ranks."""
    W = 1 if grp is None else grp.world
    rank = 0 if grp is None else grp.rank
    ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True, device=device)
    x, payload = synth(api, torch, ctx, n_frames, seed=3 + rank)
    D = ctx.data_symbols(NBYTES)
    chain_bytes = n_frames * (SPAN * 8 + NBYTES)  # algorithmic: the capture once + the decoded payload
    out = {"workload": "cfg3: 2080-sample 64QAM frames at stride 2176, delay 1..64, CFO +-0.95 pi/80, FIR CHANNEL, 30 dB noise "
                       "(channel_batch = src/channel.rs:33-74 on the GPU)", "frames": n_frames, "n_gpus": W,
           "frames_per_gpu": n_frames}

    ctx.set_tuning("one_pass_rx", 0)

    def leg(name, lags, one_pass):
        ctx.set_tuning("one_pass_rx", int(one_pass))
        try:
            ms, per_rank, r = _timed(ctx, torch, lambda: ctx.decode_batch(x, max_symbols=D, n_lags=lags), steps, grp)
        finally:
            ctx.set_tuning("one_pass_rx", 0)
        out[name] = {"ms": ms, "ms_per_rank": per_rank, "msamples_per_s": W * n_frames * SPAN / ms / 1e3, "hbm_passes": 1 if one_pass else 2,
                     "dispatch": ctx.last_dispatch(),
                     "roofline": {"bound": "hbm", "achieved": chain_bytes / (ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": chain_bytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, "per": "GPU (slowest rank)",
                                  "algorithmic_bytes_per_launch": chain_bytes,
                                  "kernels": "k_sc_cf<256,2,3,6,true> (one pass)" if one_pass else
                                             "k_sc_cf<128,2,4,0> (one wavefront per frame) over the first lags + k_sc_cf<256,2,4,0> over the frames they do not determine (bounded: k_sc_cf<128,2,4,0> alone) + k_sc_post + k_sc_tile<list> + k_rx_prepare + k_rxframe64<6,true> (finish fused); see dispatch"}}
        return r

    full = leg("full_chain_all_lags", 0, False)
    nok, ber = _ber(torch, full, payload)  # this rank's frames (rank 0's in the report)
    out["frames_decoded"] = nok
    out["ber_decoded_frames_vs_tx_payload"] = ber
    bounded = leg("full_chain_bounded_256_lags", SYNC_LAGS, False)
    one = leg("full_chain_all_lags_one_pass_kernel", 0, True)
    both = (full["status"] == 0) & (bounded["status"] == 0)
    out["bounded_vs_full_search"] = {
        "frames_ok_in_both_but_different": int(((full["offset"] != bounded["offset"]) | (full["len"] != bounded["len"])
                                                | (full["bytes"][:, :NBYTES] != bounded["bytes"][:, :NBYTES]).any(dim=1))[both].sum()),
        "frames_with_different_status": int((full["status"] != bounded["status"]).sum()),
        "note": "n_lags also clips the peak window [d1, d1 + W]: see DESIGN.md section 3 (EXT-3) and "
                "tests/test_gpu_parity.py::test_bounded_search_clips_the_peak_window"}
    out["one_pass_vs_staged"] = {
        "status_offset_len_equal": bool(((one["status"] == full["status"]) & (one["offset"] == full["offset"])
                                         & (one["len"] == full["len"])).all()),
        "frames_with_different_bytes": int((one["bytes"][:, :NBYTES] != full["bytes"][:, :NBYTES]).any(dim=1).sum()),
        "max_cfo_difference": float((one["f_delta"] - full["f_delta"]).abs().max())}
    if cpu and rank == 0:
        from tools import cpu_baseline as cb
        threads = cb.host_threads()
        out["cpu_baseline"] = cpu_leg(x, payload, full, D, min(n_frames, 65536, 1024 * threads), threads)
        out["speedup_vs_cpu"] = out["full_chain_all_lags"]["msamples_per_s"] / out["cpu_baseline"]["value"]
    del full, bounded, one
    # --- TX side of the hot path: encode (modulate + encode_block + IFFT + CP + header + normalise) for the same payloads
    npay = min(n_frames, 262144)
    txo = ctx.encode_batch(payload[:npay])
    tms, _, _ = _timed(ctx, torch, lambda: ctx.encode_batch(payload[:npay], out=txo), steps, grp)
    tx_bytes = npay * (txo.shape[-1] * 8 + NBYTES)
    out["tx_encode"] = {"kernel": "k_txframe64<6, true>", "frames": npay, "ms": tms, "msamples_per_s": W * npay * txo.shape[-1] / tms / 1e3,
                        "roofline": {"bound": "hbm", "achieved": tx_bytes / (tms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": tx_bytes / (tms / 1e3) / 1e9 / HBM_PEAK_GBS}}
    del txo
    # --- Schmidl-Cox alone over every lag of every slot: (a) the kernel that COMPUTES every lag (one launch, tuning sc_first_lags = 0:
    #     the north-star kernel against the HBM roofline), (b) the product path: the first 384 lags decide every frame whose crossing and
    #     peak window lie among them, the rest take the whole search (same results; it reads a third of the slot when the packet is early)
    sc_bytes = n_frames * (SPAN * 8 + 16)
    first = ctx.get_tuning("sc_first_lags")
    for name, fl in (("schmidl_cox", 0), ("schmidl_cox_two_launches", first)):
        ctx.set_tuning("sc_first_lags", fl)
        try:
            sms, _, _ = _timed(ctx, torch, lambda: ctx.sc_correlate(x), steps, grp)
            disp = ctx.last_dispatch()
        finally:
            ctx.set_tuning("sc_first_lags", first)
        out[name] = {"kernel": disp + (" (all 1857 lags of every 2176-sample slot computed)" if fl == 0 else
                                       f" (first {fl} lags, then the whole search for the frames they do not determine)"),
                     "kernel_ms": sms, "msamples_per_s": W * n_frames * SPAN / sms / 1e3,
                     "roofline": {"bound": "hbm", "achieved": sc_bytes / (sms / 1e3) / 1e9, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": sc_bytes / (sms / 1e3) / 1e9 / HBM_PEAK_GBS,
                                  "algorithmic_bytes_per_launch": sc_bytes}}
    return out
