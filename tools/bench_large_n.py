"""Configs 4 and 5 (BASELINE.json configs[3], [4]) on ONE GPU -- the 8-GPU versions are frame-index replicas of these
(no data-path collective), so per-GPU numbers are what scales.
   cfg4: N=1024, 64-QAM + Hamming(7,4), full RX chain; a resident ring of 64 Ki frames (9.5 GB) through the library's TX and
         GPU channel model (FIR CHANNEL, delay, signed CFO; 40 dB in channel.rs's definition, which is ~28 dB against the DATA
         symbols' power: at N = 1024 the time-domain header blocks dominate the frame's pseudo-variance), re-processed until 10 M
         frames are counted (BASELINE: "10M-frame stream");
   cfg5: N=4096, 256-QAM, continuous symbols: TX (map + IFFT + CP) writes HBM, RX (CP strip + FFT + demap) reads it back.
Each block carries a roofline object, a bounded-sample CPU baseline (the oracle, all cores) and GPU-vs-CPU equality."""
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ofdm_amd import api

HBM_PEAK_GBS = 8000.0


def _timed(ctx, fn, steps, grp=None):
    from tools import rank_timing
    ms, per_rank, r = rank_timing.timed(ctx, torch, fn, steps, grp)
    return ms, r, per_rank


def cfg5(n_sym=65536, steps=5, cpu=True, grp=None, device=None):
    """grp (ofdm_amd.dist.Group, optional): every rank runs its OWN continuous stream of n_sym symbols (BASELINE configs[4]:
    "continuous stream, 8 x MI355X"); times are the max over ranks, rates the aggregate over all ranks."""
    W = 1 if grp is None else grp.world
    rank = 0 if grp is None else grp.rank
    device = torch.cuda.current_device() if device is None else device
    ctx = api.Context(n_fft=4096, modulation=api.QAM256, guard_bands=True, device=device)
    g = torch.Generator(device=ctx.device); g.manual_seed(5 + rank)
    bps = ctx.bytes_per_symbol
    nb = n_sym * bps
    pay = torch.randint(0, 256, (nb,), dtype=torch.uint8, device=ctx.device, generator=g)

    def tx_staged():
        pts = ctx.modulate(pay)
        bins = ctx.encode_block(pts.view(-1, ctx.data_carriers))
        return ctx.prefix_block(bins)

    tx_staged_ms, x, _ = _timed(ctx, tx_staged, 2, grp)
    xf = torch.empty((n_sym, ctx.S), dtype=torch.complex64, device=ctx.device)
    tx_ms, _, tx_per_rank = _timed(ctx, lambda: ctx.tx_symbols(pay, out=xf), steps, grp)   # the same three stages in one pass (k_tx4096)
    tx_disp = ctx.last_dispatch()
    same = float((xf.view(-1) - x.view(-1)).abs().max() / x.view(-1).abs().max())
    del x
    out = torch.empty((1, nb), dtype=torch.uint8, device=ctx.device)
    rx_ms, _, rx_per_rank = _timed(ctx, lambda: ctx.rx_demod(xf.view(1, -1), syms_per_frame=n_sym, out=out), steps, grp)
    rx_disp = ctx.last_dispatch()
    ok = bool((out.view(-1) == pay).all())

    def both():
        ctx.tx_symbols(pay, out=xf)
        return ctx.rx_demod(xf.view(1, -1), syms_per_frame=n_sym, out=out)

    both_ms, _, both_per_rank = _timed(ctx, both, steps, grp)
    ns = n_sym * ctx.S
    tx_bytes, rx_bytes = ns * 8 + nb, ns * 8 + nb
    res = {"workload": "cfg5: N=4096 256QAM guard bands, continuous symbols, TX IFFT then RX FFT", "symbols": n_sym,
           "n_gpus": W, "symbols_per_gpu": n_sym, "samples_per_gpu": ns,
           "samples": W * ns, "tx_ms": tx_ms, "rx_ms": rx_ms, "tx_then_rx_ms": both_ms,
           "tx_ms_per_rank": tx_per_rank, "rx_ms_per_rank": rx_per_rank, "tx_then_rx_ms_per_rank": both_per_rank,
           "tx_msamples_per_s": W * ns / tx_ms / 1e3, "rx_msamples_per_s": W * ns / rx_ms / 1e3,
           "tx_then_rx_msamples_per_s": W * ns / both_ms / 1e3, "tx_staged_ms": tx_staged_ms,
           "dispatch": {"tx": tx_disp, "rx": rx_disp},
           "tx_fused_vs_staged_max_rel_err": same, "rx_bytes_equal_tx_payload": ok,
           "roofline_tx": {"bound": "hbm", "kernel": "ofdm::k_tx4096<true>", "achieved": tx_bytes / (tx_ms / 1e3) / 1e9,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": tx_bytes / (tx_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                           "algorithmic_bytes_per_launch": tx_bytes},
           "roofline_rx": {"bound": "hbm", "kernel": "ofdm::k_demod4096<8, true>", "achieved": rx_bytes / (rx_ms / 1e3) / 1e9,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rx_bytes / (rx_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                           "algorithmic_bytes_per_launch": rx_bytes}}
    if cpu and rank == 0:
        from oracle import oracle as orc
        from tools import cpu_baseline as cb

        orc.lib(); orc.set_fft_cache(True)
        threads = cb.host_threads()
        per = 8
        n_s = threads * per
        payh = pay[: n_s * bps].cpu().numpy()
        nd = ctx.data_carriers

        def work(i):
            blk = bytes(payh[i * per * bps:(i + 1) * per * bps])
            pts = orc.modulate(blk, orc.QAM256)
            sym = np.concatenate([orc.prefix_block(orc.encode_block(pts[k * nd:(k + 1) * nd], 4096, True)[0]) for k in range(per)])
            rx = orc.rx_demod(sym.astype(np.complex64).astype(np.complex128), 4096, True, orc.QAM256)
            return sym, rx

        rec, outs = cb.timed(work, list(range(threads)), n_s * ctx.S / 1e6, target_s=2.5)
        gx = xf[:n_s].cpu().numpy()
        gb = out.view(-1)[: n_s * bps].cpu().numpy()
        tx_err = max(float(np.linalg.norm(gx[i * per:(i + 1) * per].ravel() - outs[i][0]) / np.linalg.norm(outs[i][0])) for i in range(threads))
        rx_same = all(bytes(gb[i * per * bps:(i + 1) * per * bps]) == outs[i][1] for i in range(threads))
        orc.set_fft_cache(False)
        res["cpu_baseline"] = {"value": rec["value"], "unit": "Msamples/s (TX + RX of the same symbols)", "cores": threads, "kind": "port",
                               "sample": f"first {n_s} symbols x {rec['passes_over_sample']} passes, oracle modulate + encode_block + "
                                         f"prefix_block, then rx_demod (f64, cached twiddles), {rec['seconds']:.1f} s wall",
                               "gpu_tx_vs_cpu_tx_max_rel_err": tx_err, "gpu_bytes_equal_cpu_bytes": bool(rx_same)}
        res["speedup_vs_cpu"] = res["tx_then_rx_msamples_per_s"] / rec["value"]
    return res


CFG4_NBYTES = 1304  # -> 2282 coded bytes + 16-byte header = 4 data symbols of 576 B: the 17 920-sample frame BASELINE.md suggests


def _cfg4_ring(ctx, g, n_frames, span, snr_db, rank, max_delay=64, noise_only=0.0):
    pay = torch.randint(0, 256, (n_frames, CFG4_NBYTES), dtype=torch.uint8, device=ctx.device, generator=g)
    x = torch.empty((n_frames, span), dtype=torch.complex64, device=ctx.device)
    chunk = 8192
    for lo in range(0, n_frames, chunk):
        hi = min(lo + chunk, n_frames)
        tx = ctx.encode_batch(pay[lo:hi].contiguous())
        d = torch.randint(1, max_delay + 1, (hi - lo,), device=ctx.device, generator=g, dtype=torch.int32)
        fd = (torch.rand((hi - lo,), device=ctx.device, generator=g, dtype=torch.float64) * 1.9 - 0.95) * math.pi / ctx.S
        ctx.channel_batch(tx, snr_db=snr_db, seed=4_000_003 + lo + 1_000_000_007 * rank + 17 * max_delay, delay=d, f_delta=fd, out=x[lo:hi])
        if noise_only > 0:  # slots without a packet: noise of the packets' own floor (see tools/bench_cfg3.synth)
            empty = torch.rand((hi - lo,), device=ctx.device, generator=g) < noise_only
            k = int(empty.sum())
            if k:
                sigma = float(x[lo:hi, 0].abs().mean()) * 0.755
                x[lo:hi][empty] = torch.view_as_complex(torch.randn((k, span, 2), dtype=torch.float32, device=ctx.device, generator=g) * sigma)
        del tx
    torch.cuda.synchronize()
    return x, pay


def cfg4(n_frames=65536, total_frames=10_000_000, cpu=True, snr_db=40.0, grp=None, device=None):
    """grp (ofdm_amd.dist.Group, optional): BASELINE configs[3] is "10M-frame stream, frame-sharded across 8 x MI355X" -- rank r of R
    owns frames [r F / R, (r + 1) F / R) of the stream (ofdm_amd.dist.shard_range) and counts them by passes over its OWN
    resident ring of n_frames frames; no data-path collective.  Times are the max over ranks, rates the aggregate."""
    from ofdm_amd.dist import shard_range
    from tools.bench_cfg3 import required_sync_bytes, _roof, _traffic
    W = 1 if grp is None else grp.world
    rank = 0 if grp is None else grp.rank
    device = torch.cuda.current_device() if device is None else device
    ctx = api.Context(n_fft=1024, modulation=api.QAM64, guard_bands=True, ecc=api.ECC_HAMMING74, device=device)
    g = torch.Generator(device=ctx.device); g.manual_seed(4 + rank)
    D = ctx.data_symbols(CFG4_NBYTES)
    flen = ctx.frame_samples(CFG4_NBYTES)
    span = flen + 256
    x, pay = _cfg4_ring(ctx, g, n_frames, span, snr_db, rank)
    lo_f, hi_f = shard_range(total_frames, rank, W)          # this rank's share of the stream
    my_frames = hi_f - lo_f
    passes = max(1, -(-my_frames // n_frames))
    res = {"workload": f"cfg4: N=1024 64QAM + Hamming(7,4), frames of {flen} samples in {span}-sample slots, FIR CHANNEL, delay 1..64, "
                       f"CFO +-0.95 pi/1280, {snr_db:g} dB (channel.rs definition), full RX chain",
           "parity": "N = 1024, 64-QAM, Hamming(7,4) and the Schmidl-Cox detector are north-star extensions (EXT-1..4): parity unpinned by the "
                     "reference, the oracle is the definition",
           "n_gpus": W, "stream_frames": total_frames, "stream_frames_this_rank": my_frames,
           "ring_frames": n_frames, "passes": passes, "frames_counted": W * passes * n_frames, "data_symbols": D}
    L, Wn = ctx.S, ctx.params.sync_window_reps * ctx.S

    def block(x, pay, span, passes, names):
        slot_bytes = n_frames * (span * 8 + CFG4_NBYTES)
        d_all, _, _ = ctx.sc_correlate(x)
        sync_req = required_sync_bytes(torch, d_all, span, Wn, L)
        found = int((d_all >= 0).sum())
        chain_req = sync_req + found * ((5 + D) * ctx.n_fft * 8 + CFG4_NBYTES)
        out = {"required_bytes": {"sync_per_frame": sync_req / n_frames, "chain_per_frame": chain_req / n_frames, "slot_per_frame": span * 8,
                                  "frames_with_a_detection": found, "definition": "as in cfg3.required_bytes"}}
        full = None
        for name, lags in names:
            ms, r, per_rank = _timed(ctx, lambda: ctx.decode_batch(x, max_symbols=D, n_lags=lags), passes, grp)
            ok = (r["status"] == 0) & (r["len"] >= CFG4_NBYTES)
            good = int(((r["bytes"][:, :CFG4_NBYTES] == pay).all(dim=1) & ok).sum())
            diff = torch.bitwise_xor(r["bytes"][:, :CFG4_NBYTES], pay)[ok]
            bits = sum(int(((diff >> sh) & 1).sum()) for sh in range(8))
            # k_sc_stream stops reading a slot when the peak window has closed -- over every lag AND in a bounded search; the bytes the
            # decision requires are what its roofline is measured against (a bounded search never needs more than its last lag's window)
            req = chain_req if lags == 0 else min(chain_req, n_frames * ((lags + Wn + L) * 8) + found * ((5 + D) * ctx.n_fft * 8 + CFG4_NBYTES))
            tr, src = _traffic(("k_sc_stream", "k_rxframe1024")) if lags == 0 and span == flen + 256 else (None, None)
            out[name] = {"ms_per_pass": ms, "ms_per_pass_per_rank": per_rank, "stream_seconds": ms * passes / 1e3,
                         "msamples_per_s": W * n_frames * span / ms / 1e3, "dispatch": ctx.last_dispatch(),
                         "frames_synchronised": int(ok.sum()), "frames_decoded_exactly": good,
                         "ber_after_hamming_vs_tx_payload": bits / max(1, int(ok.sum()) * CFG4_NBYTES * 8),
                         "capture_throughput": {"gb_per_s": slot_bytes / (ms / 1e3) / 1e9, "of_hbm_peak": slot_bytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                                                "note": "slot bytes / time; not a roofline: the streaming detector stops reading a slot early"},
                         "roofline": _roof(req, ms, algorithmic_bytes_per_launch=req, per="GPU (slowest rank)", bytes="required (see required_bytes)",
                                           traffic=None if tr is None else tr * n_frames, traffic_source=src,
                                           kernels="k_sc_stream<2> (L = 1280, stops when the peak window has closed) + k_rx_prepare + k_rxframe1024<6,true> (Hamming decode fused); see dispatch")}
            if lags == 0:
                full = r
        return out, full

    blk, full = block(x, pay, span, passes, (("full_chain_all_lags", 0), ("full_chain_bounded_2048_lags", 2048)))
    res.update(blk)
    if cpu and rank == 0:
        from oracle import oracle as orc
        from tools import cpu_baseline as cb

        orc.lib(); orc.set_fft_cache(True)
        threads = cb.host_threads()
        per = 4
        n_s = min(n_frames, threads * per)
        xs = x[:n_s].cpu().numpy()
        wide = [[xs[j].astype(np.complex128) for j in range(i, n_s, threads)] for i in range(threads)]

        def work(frames):
            return [orc.decode_sc(f, True, orc.QAM64, 1024, max_symbols=D) for f in frames]

        rec, outs = cb.timed(work, wide, n_s * span / 1e6, target_s=2.5)
        st = full["status"][:n_s].cpu().numpy(); off = full["offset"][:n_s].cpu().numpy()
        ln = full["len"][:n_s].cpu().numpy(); by = full["bytes"][:n_s].cpu().numpy()
        same = differ = sync_differ = 0
        for i, resl in enumerate(outs):
            for k, w in enumerate(resl):
                j = i + k * threads
                if st[j] != w["status"] or (w["status"] == 0 and off[j] != w["offset"]):
                    sync_differ += 1
                elif w["status"] != 0 or bytes(by[j][: ln[j]]) == orc.hamming74_decode(w["bytes"])[0]:
                    same += 1
                else:
                    differ += 1
        orc.set_fft_cache(False)
        res["cpu_baseline"] = {"value": rec["value"], "unit": "Msamples/s", "cores": threads, "kind": "port",
                               "sample": f"first {n_s} frames x {rec['passes_over_sample']} passes, oracle decode_sc over all lags + "
                                         f"Hamming decode (f64, cached twiddles), {rec['seconds']:.1f} s wall",
                               "frames_compared": n_s, "frames_identical_to_gpu": same, "frames_with_a_differing_decision": differ,
                               "frames_with_different_status_or_offset": sync_differ,
                               "gpu_bytes_equal_cpu_bytes": differ == 0 and sync_differ == 0}
        res["speedup_vs_cpu"] = res["full_chain_all_lags"]["msamples_per_s"] / rec["value"]
    del x, pay, full
    torch.cuda.empty_cache()
    # --- the placements the early exit cannot decide early: delay uniform over 4096 samples of slack, 10 % of the slots
    #     without a packet; a few passes are enough for a rate
    late_span = flen + 4096 + 256
    xl, pl = _cfg4_ring(ctx, g, n_frames, late_span, snr_db, rank, max_delay=4096, noise_only=0.10)
    lblk, lfull = block(xl, pl, late_span, min(passes, 8), (("full_chain_all_lags_late_packets", 0),))
    late_cpu = None
    if cpu and rank == 0:  # the oracle on the first 256 late / empty slots
        from oracle import oracle as orc
        orc.lib(); orc.set_fft_cache(True)
        xs = xl[:256].cpu().numpy()
        st = lfull["status"][:256].cpu().numpy(); off = lfull["offset"][:256].cpu().numpy()
        ln = lfull["len"][:256].cpu().numpy(); by = lfull["bytes"][:256].cpu().numpy()
        same = 0
        for j in range(256):
            w = orc.decode_sc(xs[j].astype(np.complex128), True, orc.QAM64, 1024, max_symbols=D)
            same += int(st[j] == w["status"] and (w["status"] != 0 or (off[j] == w["offset"] and bytes(by[j][: ln[j]]) == orc.hamming74_decode(w["bytes"])[0])))
        orc.set_fft_cache(False)
        late_cpu = {"frames_compared": 256, "frames_identical_to_gpu": same, "gpu_bytes_equal_cpu_bytes": same == 256}
    res["late_packets"] = {"cpu_check": late_cpu,"workload": f"the same frames in {late_span}-sample slots, delay uniform over 1..4096, 10 % of the slots noise only", **lblk}
    return res


if __name__ == "__main__":
    out = {}
    small = "--small" in sys.argv
    for name, fn in (("cfg5", (lambda: cfg5(8192, 3)) if small else cfg5), ("cfg4", (lambda: cfg4(8192, 65536)) if small else cfg4)):
        try:
            out[name] = fn()
        except Exception as e:
            import traceback
            out[name] = {"error": repr(e), "trace": traceback.format_exc()[-600:]}
    print(json.dumps(out))
