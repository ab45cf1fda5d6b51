"""Config 3 (BASELINE.json configs[2]): 1 M full frames with the Schmidl-Cox preamble, the reference's FIR channel, a random
integer delay, a random signed CFO and 30 dB noise; full RX chain (timing -> CFO -> channel estimate -> FFT -> equalise ->
pilot phase -> demap -> length header) on one GPU.  Reports
  * the chain searching EVERY lag of every slot (the headline of this block) and bounded to the slot's 256 possible lags,
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


LATE_SPAN = 2544   # late-packet layout: about the largest slot whose every lag the one-tile kernel holds (2225 of at most 2240); the frame may start anywhere it still fits
LATE_NOISE_ONLY = 0.10


def synth(api, torch, ctx, n_frames, span=SPAN, snr_db=30.0, seed=3, max_delay=64, noise_only=0.0):
    """[n_frames, span] captures + payloads.  TX by the library, channel by the library's GPU restatement of
    src/channel.rs (FIR CHANNEL, CFO, noise) with the test-bench extensions: per-frame delay, signed CFO.
    max_delay: delays are uniform over [1, max_delay]; noise_only: that share of the slots holds no packet at all (noise of the
    packets' own level) -- the placements an early-exit search cannot decide early."""
    g = torch.Generator(device=ctx.device)
    g.manual_seed(seed)
    x = torch.empty((n_frames, span), dtype=torch.complex64, device=ctx.device)
    payload = torch.randint(0, 256, (n_frames, NBYTES), dtype=torch.uint8, device=ctx.device, generator=g)
    chunk = 65536
    for lo in range(0, n_frames, chunk):
        hi = min(lo + chunk, n_frames)
        tx = ctx.encode_batch(payload[lo:hi].contiguous())  # [m, 2080]
        d = torch.randint(1, max_delay + 1, (hi - lo,), device=ctx.device, generator=g, dtype=torch.int32)
        fd = (torch.rand((hi - lo,), device=ctx.device, generator=g, dtype=torch.float64) * 1.9 - 0.95) * math.pi / ctx.S
        ctx.channel_batch(tx, snr_db=snr_db, seed=seed * 1_000_003 + lo, delay=d, f_delta=fd, out=x[lo:hi])
        if noise_only > 0:
            empty = torch.rand((hi - lo,), device=ctx.device, generator=g) < noise_only
            k = int(empty.sum())
            if k:
                # the packets' own noise floor, from the first sample of every slot (always in front of the packet: delay >= 1);
                # channel.rs draws U(-1, 1) s per component: mean |.| = 0.765 s, same power as a Gaussian of s / sqrt(3)
                sigma = float(x[lo:hi, 0].abs().mean()) * 0.755
                nz = torch.randn((k, span, 2), dtype=torch.float32, device=ctx.device, generator=g) * sigma
                x[lo:hi][empty] = torch.view_as_complex(nz)
                del nz
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


def required_sync_bytes(torch, d_hat, span, W, L):
    """The samples ANY threshold-then-peak detector has to read, in bytes: every lag up to the reported peak d_hat must have been
    evaluated, and the last of them spans W + L samples -- the whole slot where nothing is found.  (A lower bound: the peak window
    really runs to d1 + W >= d_hat.)  This, not the slot size, is what an early-exit search is measured against."""
    need = torch.where(d_hat >= 0, torch.clamp(d_hat.to(torch.int64) + (W + L), max=span), torch.full_like(d_hat, span, dtype=torch.int64))
    return int(need.sum()) * 8


def cpu_leg(x, payload, gpu, D, n_sample, threads, span=SPAN, target_s=3.0):
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

    rec, outs = cb.timed(work, wide, n_sample * span / 1e6, target_s=target_s)
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


def _roof(bytes_, ms, **extra):
    gbs = bytes_ / (ms / 1e3) / 1e9
    return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, **extra}


def _traffic(kernels):
    """HBM bytes per frame of the named kernels from the newest committed PMC pass (tools/pmc_traffic5.py), or None."""
    import json
    for rnd in ("r05", "r04", "r03"):
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{rnd}_pmc_traffic.json")
        try:
            pm = json.load(open(path))["per_kernel"]
            tot = 0.0
            for k in kernels:
                tot += pm[k]["read_bytes_per_frame"] + pm[k].get("write_bytes_per_frame", 0.0)
            return tot, f"profiles/{rnd}_pmc_traffic.json (separate --pmc passes of tools/pmc_traffic5.py, not measured in this run)"
        except Exception:
            continue
    return None, None


def chain_block(api, torch, ctx, x, payload, n_frames, steps, grp, W_, name_suffix="", bounded=True, span=SPAN, traffic_keys=None):
    """The full RX chain over one data set: every lag, bounded; Schmidl-Cox alone (every lag computed; the product's
    two-launch early-exit search).  Every early-exit block carries TWO figures: `capture_throughput` (slot bytes / time: what a user
    sees, NOT a roofline -- an early exit skips bytes) and `roofline` (bytes the decision REQUIRES / time)."""
    out = {}
    D = ctx.data_symbols(NBYTES)
    L, Wn = ctx.S, ctx.params.sync_window_reps * ctx.S
    slot_bytes = n_frames * (span * 8 + NBYTES)
    d_all, _, _ = ctx.sc_correlate(x)
    sync_req = required_sync_bytes(torch, d_all, span, Wn, L)
    found = int((d_all >= 0).sum())
    # receive body: the (5 + D) N useful samples of every frame found (CP never needed) + the decoded payload
    chain_req = sync_req + found * ((5 + D) * ctx.n_fft * 8 + NBYTES)
    out["required_bytes"] = {"sync_per_frame": sync_req / n_frames, "chain_per_frame": chain_req / n_frames, "slot_per_frame": span * 8,
                             "frames_with_a_detection": found,
                             "definition": "sync: 8 B x min(slot, d_hat + W + L) per slot (the whole slot when nothing is found) -- what any "
                                           "threshold-then-peak detector must read; chain: + (5 + D) N useful samples and the payload of every "
                                           "frame found"}

    def leg(name, lags):
        ms, per_rank, r = _timed(ctx, torch, lambda: ctx.decode_batch(x, max_symbols=D, n_lags=lags), steps, grp)
        one = False
        early_exit = lags == 0                      # the streaming detector stops reading a slot once its decision is determined
        bytes_ = chain_req if early_exit else slot_bytes
        tr, src = _traffic(traffic_keys) if (traffic_keys and early_exit) else (None, None)
        out[name + name_suffix] = {
            "ms": ms, "ms_per_rank": per_rank, "msamples_per_s": W_ * n_frames * span / ms / 1e3, "hbm_passes": 1 if one else 2,
            "dispatch": ctx.last_dispatch(),
            "capture_throughput": {"gb_per_s": slot_bytes / (ms / 1e3) / 1e9, "of_hbm_peak": slot_bytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                                   "note": "slot bytes / time; not a roofline when the search exits early"},
            "roofline": _roof(bytes_, ms, per="GPU (slowest rank)", algorithmic_bytes_per_launch=bytes_,
                              bytes="required (see required_bytes)" if early_exit else "the whole slot once + payload",
                              traffic=None if tr is None else tr * n_frames, traffic_source=src,
                              kernels="k_sc80<2> (every lag on f64 prefix differences; stops when the peak window has closed) + k_sc_post + "
                                      "k_sc_tile<list> + k_rx_prepare + k_rxframe64<6,true> (finish fused); see dispatch")}
        return r

    full = leg("full_chain_all_lags", 0)
    res = {"full": full}
    if bounded:
        res["bounded"] = leg("full_chain_bounded_256_lags", SYNC_LAGS)
    # --- Schmidl-Cox alone: (a) the product's search, k_sc80 -- every lag it evaluates is exact, and it stops reading a slot when the
    #     peak window has closed: capture throughput + roofline on the bytes the decision requires; (b) the round-4 f32 filter kernel that
    #     computes EVERY lag of every slot (tuning no_sc80 = 1, sc_first_lags = 0): the whole slot is read, so slot bytes are its roofline
    sc_slot = n_frames * (span * 8 + 16)
    first = ctx.get_tuning("sc_first_lags")
    for name, keys in (("schmidl_cox", {}), ("schmidl_cox_f32_filter_every_lag", {"no_sc80": 1, "sc_first_lags": 0})):
        for k, v in keys.items():
            ctx.set_tuning(k, v)
        try:
            sms, _, _ = _timed(ctx, torch, lambda: ctx.sc_correlate(x), steps, grp)
            disp = ctx.last_dispatch()
        finally:
            ctx.set_tuning("sc_first_lags", first)
            ctx.set_tuning("no_sc80", 0)
        every = bool(keys)
        bytes_ = sc_slot if every else sync_req + 16 * n_frames
        out[name + name_suffix] = {
            "kernel": disp + (f" (all {span - Wn - L + 1} lags of every {span}-sample slot computed)" if every else
                              " (every lag up to the end of the peak window, exactly; the rest of the slot is never fetched)"),
            "kernel_ms": sms, "msamples_per_s": W_ * n_frames * span / sms / 1e3,
            "capture_throughput": {"gb_per_s": sc_slot / (sms / 1e3) / 1e9, "of_hbm_peak": sc_slot / (sms / 1e3) / 1e9 / HBM_PEAK_GBS},
            "roofline": _roof(bytes_, sms, algorithmic_bytes_per_launch=bytes_,
                              bytes="the whole slot (every lag is computed)" if every else "required (see required_bytes)")}
    return out, res


def run(api, torch, n_frames, steps, device, cpu=True, grp=None):
    """grp (ofdm_amd.dist.Group, optional): every rank decodes its OWN n_frames captures (weak scaling, no data-path
    collective); times are the max over ranks, rates the aggregate over all ranks."""
    W = 1 if grp is None else grp.world
    rank = 0 if grp is None else grp.rank
    ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True, device=device)
    x, payload = synth(api, torch, ctx, n_frames, seed=3 + rank)
    D = ctx.data_symbols(NBYTES)
    out = {"workload": "cfg3: 2080-sample 64QAM frames at stride 2176, delay 1..64, CFO +-0.95 pi/80, FIR CHANNEL, 30 dB noise "
                       "(channel_batch = src/channel.rs:33-74 on the GPU)", "frames": n_frames, "n_gpus": W,
           "frames_per_gpu": n_frames,
           "parity": "64-QAM and the Schmidl-Cox detector are north-star extensions the reference lacks (EXT-1, EXT-3): parity unpinned by "
                     "the reference, the oracle is the definition"}
    blocks, res = chain_block(api, torch, ctx, x, payload, n_frames, steps, grp, W,
                              traffic_keys=("k_sc80", "k_rxframe64"))
    out.update(blocks)
    full, bounded = res["full"], res["bounded"]
    nok, ber = _ber(torch, full, payload)  # this rank's frames (rank 0's in the report)
    out["frames_decoded"] = nok
    out["ber_decoded_frames_vs_tx_payload"] = ber
    both = (full["status"] == 0) & (bounded["status"] == 0)
    out["bounded_vs_full_search"] = {
        "frames_ok_in_both_but_different": int(((full["offset"] != bounded["offset"]) | (full["len"] != bounded["len"])
                                                | (full["bytes"][:, :NBYTES] != bounded["bytes"][:, :NBYTES]).any(dim=1))[both].sum()),
        "frames_with_different_status": int((full["status"] != bounded["status"]).sum()),
        "note": "n_lags also clips the peak window [d1, d1 + W]: see DESIGN.md section 3 (EXT-3) and "
                "tests/test_gpu_parity.py::test_bounded_search_clips_the_peak_window"}
    if cpu and rank == 0:
        from tools import cpu_baseline as cb
        threads = cb.host_threads()
        out["cpu_baseline"] = cpu_leg(x, payload, full, D, min(n_frames, 65536, 1024 * threads), threads)
        out["speedup_vs_cpu"] = out["full_chain_all_lags"]["msamples_per_s"] / out["cpu_baseline"]["value"]
    del full, bounded, res
    # --- slots without a packet: k_sc80 has to evaluate every lag and read every byte -- its rate against the slot roofline
    nn = min(n_frames, 262144)
    xn = torch.view_as_complex(torch.randn((nn, SPAN, 2), dtype=torch.float32, device=ctx.device) * 0.004)
    nms, _, _ = _timed(ctx, torch, lambda: ctx.sc_correlate(xn), steps, grp)
    out["schmidl_cox_noise_only_slots"] = {"kernel": ctx.last_dispatch() + " (no packet: every lag of every slot evaluated exactly, every byte read)",
                                           "frames": nn, "kernel_ms": nms, "roofline": _roof(nn * (SPAN * 8 + 16), nms, bytes="the whole slot")}
    del xn
    # --- TX side of the hot path: encode (modulate + encode_block + IFFT + CP + header + normalise) for the same payloads
    npay = min(n_frames, 262144)
    txo = ctx.encode_batch(payload[:npay])
    tms, _, _ = _timed(ctx, torch, lambda: ctx.encode_batch(payload[:npay], out=txo), steps, grp)
    tx_bytes = npay * (txo.shape[-1] * 8 + NBYTES)
    out["tx_encode"] = {"kernel": "k_txframe64<6, true>", "frames": npay, "ms": tms, "msamples_per_s": W * npay * txo.shape[-1] / tms / 1e3,
                        "roofline": _roof(tx_bytes, tms)}
    del txo, x, payload
    torch.cuda.empty_cache()
    # --- the placements an early exit cannot decide early: delay uniform over the whole slack of a 2560-sample slot
    #     (half of the crossings beyond the first launch's reach) and 10 % of the slots without a packet
    late_delay = LATE_SPAN - 2080 - 63
    xl, pl = synth(api, torch, ctx, n_frames, span=LATE_SPAN, seed=31 + rank, max_delay=late_delay, noise_only=LATE_NOISE_ONLY)
    lblocks, lres = chain_block(api, torch, ctx, xl, pl, n_frames, steps, grp, W, name_suffix="_late_packets", bounded=False, span=LATE_SPAN)
    nok_l, ber_l = _ber(torch, lres["full"], pl)
    late_cpu = None
    if cpu and rank == 0:  # the oracle on a sample of the late / empty slots: the placements round 3's f32 filter sent to the all-f64 kernel
        from tools import cpu_baseline as cb
        late_cpu = cpu_leg(xl, pl, lres["full"], D, min(n_frames, 16384), cb.host_threads(), span=LATE_SPAN, target_s=1.5)
    out["late_packets"] = {"cpu_check": late_cpu,"workload": f"the same frames in {LATE_SPAN}-sample slots, delay uniform over 1..{late_delay}, "
                                       f"{int(LATE_NOISE_ONLY * 100)} % of the slots noise only",
                           "frames_decoded": nok_l, "ber_decoded_frames_vs_tx_payload": ber_l, **lblocks}
    return out
