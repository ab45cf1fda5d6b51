#!/usr/bin/env python3
"""bench.py -- complex IQ Msamples/s through RX demod on N MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic, HBM-resident input:
  workload cfg2 (BASELINE configs[1]): 1 M frames x 16 OFDM symbols x 80 samples (64 carriers, 64-QAM, guard
  bands on), RX demod only = CP strip + FFT64 + pilot phase + hard demap -> 36 packed bytes per symbol.
One process per GPU; frames are independent, so ranks hold disjoint frame ranges (index split, weak scaling)
and the data path has no collective.  torch.distributed (RCCL) is only the barrier / max-time reduction.

Prints ONE compact JSON line (< 4 KB) on rank 0 -- the headline with its `roofline` and `cpu_baseline` objects plus ms / fraction
of the other BASELINE configurations (`configs`) -- and writes the full record of every block to --detail (bench_detail.json).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=1_000_000, help="frames per GPU (BASELINE config 2: 1M)")
    ap.add_argument("--snr-db", type=float, default=30.0)
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-cfg3", action="store_true", help="skip the config 3 / 4 / 5 reports (headline only)")
    ap.add_argument("--no-shapes", action="store_true", help="skip the per-length shapes block (N = 1 diagnostic)")
    ap.add_argument("--no-host", action="store_true", help="skip the host-buffer / long-capture blocks (N = 1 diagnostic)")
    ap.add_argument("--cfg3-frames", type=int, default=1_000_000, help="BASELINE config 3: 1M frames")
    ap.add_argument("--cfg4-ring", type=int, default=65536, help="config 4: frames in the resident ring")
    ap.add_argument("--cfg4-frames", type=int, default=10_000_000, help="config 4: frames counted (BASELINE: 10M-frame stream)")
    ap.add_argument("--cfg5-symbols", type=int, default=65536)
    ap.add_argument("--cpu-seconds", type=float, default=2.5, help="wall time of each CPU-baseline leg")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full record goes (every block of every configuration); stdout carries the compact line only")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / reduction plumbing only: no GPU work, value null (CPU tests)")
    return ap.parse_args()


def free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a):
    """`python bench.py --gpus N` started plainly (no torchrun): become the launcher.  The parent never touches the GPU
    (no HIP call, no torch.cuda call): it builds libofdm_hip.so once (so the ranks never race hipcc; the build is
    file-locked as well), starts N fresh child processes -- one rank per GPU, env as torchrun would set it -- relays
    rank 0's JSON line and exits non-zero if any rank fails.  Children are plain `python bench.py ...` processes: no
    re-exec of a process that has initialised the GPU."""
    import subprocess

    if not a.dry_run:
        from ofdm_amd import build as hip_build

        hip_build.build()
    port = free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, WORLD_SIZE=str(a.gpus), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading

    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)  # rank 0 prints the one JSON line
    reader.start()
    # A rank that dies early (no device, RCCL failure) would leave the others waiting in the rendezvous for minutes: as soon as
    # one rank has failed, the rest are ended (exactly the PIDs started above) and the launcher reports the failure.
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes):
            time.sleep(2.0)  # let the others notice on their own first
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    for p in procs:
        p.wait()
    reader.join(timeout=10.0)
    out0 = (buf[0] if buf else b"").decode()
    rc = 0
    for p in procs:
        rc = rc or p.returncode
    json_lines = [ln for ln in out0.splitlines() if ln.lstrip().startswith("{")]
    for ln in out0.splitlines():  # library chatter on rank 0's stdout (gloo / RCCL banners) goes to stderr: stdout carries ONE JSON line
        if ln not in json_lines and ln.strip():
            print(ln, file=sys.stderr)
    if json_lines:
        print(json_lines[-1], flush=True)
    if rc != 0 or not json_lines:
        print(f"bench.py: a rank failed (exit codes {[p.returncode for p in procs]})", file=sys.stderr)
        sys.exit(rc or 1)


def synth_cfg2(ctx, torch, n_frames, syms, snr_db, seed):
    """Config-2 input built on the GPU with the library's own TX stages (modulate -> encode_block -> IFFT+CP),
    AWGN from torch (plumbing), in chunks to bound the peak footprint.  Returns (samples [F, syms*S], payload [F, B])."""
    S, nd, bps = ctx.S, ctx.data_carriers, ctx.modulation
    bytes_per_frame = syms * nd * bps // 8
    g = torch.Generator(device=ctx.device)
    g.manual_seed(0x0FD3 + seed)
    x = torch.empty((n_frames, syms * S), dtype=torch.complex64, device=ctx.device)
    payload = torch.empty((n_frames, bytes_per_frame), dtype=torch.uint8, device=ctx.device)
    chunk = 65536
    sigma = None
    for lo in range(0, n_frames, chunk):
        hi = min(lo + chunk, n_frames)
        pay = torch.randint(0, 256, ((hi - lo) * bytes_per_frame,), dtype=torch.uint8, device=ctx.device, generator=g)
        payload[lo:hi] = pay.view(hi - lo, bytes_per_frame)
        pts = ctx.modulate(pay)
        bins = ctx.encode_block(pts.view(-1, nd))
        blk = ctx.prefix_block(bins).view(hi - lo, syms * S)
        if sigma is None:
            p = float((blk.real ** 2 + blk.imag ** 2).mean())
            sigma = (p / 10 ** (snr_db / 10) / 2) ** 0.5
        noise = torch.randn((hi - lo, syms * S, 2), dtype=torch.float32, device=ctx.device, generator=g) * sigma
        x[lo:hi] = blk + torch.view_as_complex(noise)
        del pts, bins, blk, noise
    torch.cuda.synchronize()
    return x, payload


def cpu_baseline_cfg2(x_host, syms, gpu_bytes, target_s):
    """The oracle (kind "port": the reference cannot be built here) timed on the host cores on a bounded sample, three ways
    (SURVEY.md 8d): ref-faithful single thread (twiddles regenerated on every transform, as the reference re-plans its FFT on
    every call, src/signals/mod.rs:41-58), cached-twiddle single thread, cached on all cores.  Returns the cpu_baseline
    object (value = all cores) and whether the GPU's bytes for the same frames equal the CPU's."""
    import numpy as np
    from oracle import oracle as orc
    from tools import cpu_baseline as cb

    orc.lib()
    F = x_host.shape[0]
    threads = cb.host_threads()
    one = [np.ascontiguousarray(x_host[: max(1, F // threads)].reshape(-1)).astype(np.complex128)]
    blocks = [np.ascontiguousarray(x_host[i::threads].reshape(-1)).astype(np.complex128) for i in range(threads)]

    def work(a):
        return orc.rx_demod(a, 64, True, orc.QAM64)

    legs = {}
    orc.set_fft_cache(False)
    legs["ref_faithful_1_thread"], _ = cb.timed(work, one, one[0].size / 1e6, target_s)
    orc.set_fft_cache(True)
    legs["cached_twiddles_1_thread"], _ = cb.timed(work, one, one[0].size / 1e6, target_s)
    legs["cached_twiddles_all_cores"], outs = cb.timed(work, blocks, F * syms * 80 / 1e6, target_s)
    orc.set_fft_cache(False)
    same = all(bytes(gpu_bytes[i::threads].reshape(-1)) == outs[i] for i in range(threads))
    top = legs["cached_twiddles_all_cores"]
    rec = {"value": top["value"], "unit": "Msamples/s", "cores": threads, "kind": "port",
           "sample": f"first {F} frames of the same batch ({F * syms * 80} samples) x {top['passes_over_sample']} passes, oracle rx_demod "
                     f"(f64, cached twiddles, {threads} threads), {top['seconds']:.1f} s wall",
           "host_cpus": os.cpu_count(),
           "variants": {k: {"value": v["value"], "unit": "Msamples/s", "threads": v["threads"], "wall_s": v["seconds"],
                            "passes_over_sample": v["passes_over_sample"]} for k, v in legs.items()},
           "gpu_bytes_equal_cpu_bytes": bool(same)}
    return rec


def dry_run(a):
    """The N-rank plumbing without the GPU: rendezvous, barrier, max / sum reductions, one JSON line from rank 0."""
    from ofdm_amd.dist import Group

    if os.environ.get("OFDM_BENCH_FAIL_RANK") == os.environ.get("RANK", "0"):  # test hook: this rank dies before the rendezvous
        sys.exit(7)
    # never RCCL here: the dry run uses no GPU, and N RCCL ranks on a box with fewer than N GPUs would wait for each other
    grp = Group(backend=os.environ.get("OFDM_DIST_BACKEND") or "gloo")
    if grp.world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but {grp.world} rank(s) came up", file=sys.stderr)
        sys.exit(3)
    grp.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        time.sleep(0.001 * (grp.rank + 1))
    grp.barrier()
    dt = time.perf_counter() - t0
    per_rank = grp.gather_floats(dt * 1e3 / a.steps)
    (dt,) = grp.reduce_max(dt)
    (ranks,) = grp.reduce_sum(1.0)
    res = {"metric": "complex IQ Msamples/s through RX demod", "value": None, "unit": "Msamples/s",
           "n_gpus": grp.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps,
           "dry_run": True, "ranks_seen": int(ranks), "world_size_seen": grp.world,
           "backend": grp.backend, "ms_per_step_per_rank": per_rank}
    # the config 3 / 4 / 5 blocks run on EVERY rank (each its own frames / ring / symbol stream) through the same barrier +
    # max-over-ranks reduction; here with sleeps in place of the kernels, so that the N-rank flow of those blocks is rehearsed
    # without a GPU (tests/test_dist_cpu.py)
    if not a.no_cfg3:
        from ofdm_amd.dist import shard_range
        for name in ("cfg3", "cfg5", "cfg4"):
            grp.barrier()
            t0 = time.perf_counter()
            time.sleep(0.002 * (grp.rank + 1))
            ms = (time.perf_counter() - t0) * 1e3
            grp.barrier()
            pr = grp.gather_floats(ms)
            blk = {"dry_run": True, "n_gpus": grp.world, "ms": max(pr), "ms_per_rank": pr}
            if name == "cfg4":
                lo, hi = shard_range(a.cfg4_frames, grp.rank, grp.world)
                (tot,) = grp.reduce_sum(float(hi - lo))
                blk["stream_frames"] = a.cfg4_frames
                blk["stream_frames_all_ranks"] = int(tot)
            res[name] = blk
    if grp.rank == 0:
        Detail(a.detail, 0).write(res)
        print(json.dumps(compact_line(res)))
    grp.close()


def _r(v, n=4):
    return round(v, n) if isinstance(v, float) else v


def _chain(blk, ms_key="ms"):
    """ms + fractions of one chain block (a `full_chain_*` object of tools/bench_cfg3.py / tools/bench_large_n.py)."""
    if not isinstance(blk, dict) or ms_key not in blk:
        return None
    return {"ms": _r(blk[ms_key]), "frac": _r((blk.get("capture_throughput") or {}).get("of_hbm_peak")),
            "frac_required": _r((blk.get("roofline") or {}).get("frac"))}


def compact_line(res):
    """The ONE stdout line: the headline with `roofline` and `cpu_baseline`, plus ms + fraction of every other BASELINE
    configuration.  Everything else (per-block rooflines, CPU legs, host-buffer and per-length blocks) lives in the detail
    file.  Kept under 4 KB: BENCH_r04.json's `parsed` was null because this line had grown to 23 KB."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "ber_vs_tx_payload", "frames_with_errors", "dry_run", "ranks_seen", "world_size_seen",
            "backend", "weak_scaling_efficiency_in_run", "speedup_vs_cpu", "detail")
    line = {k: _r(res[k], 6) for k in keep if k in res}
    if "roofline" in res:
        rk = ("bound", "kernel", "achieved", "peak", "unit", "frac", "kernel_ms", "algorithmic_bytes_per_launch", "traffic",
              "traffic_source", "frac_moved", "frac_of_box_ceiling", "out_buffer_population")
        line["roofline"] = {k: _r(res["roofline"][k], 6) for k in rk if k in res["roofline"]}
    if "cpu_baseline" in res:
        ck = ("value", "unit", "cores", "kind", "sample", "gpu_bytes_equal_cpu_bytes")
        line["cpu_baseline"] = {k: _r(res["cpu_baseline"][k], 3) for k in ck if k in res["cpu_baseline"]}
    cfgs = {}
    c3, c4, c5 = res.get("cfg3") or {}, res.get("cfg4") or {}, res.get("cfg5") or {}
    for name, blk in (("cfg3", _chain(c3.get("full_chain_all_lags"))),
                      ("cfg3_late", _chain((c3.get("late_packets") or {}).get("full_chain_all_lags_late_packets"))),
                      ("cfg4", _chain(c4.get("full_chain_all_lags"), "ms_per_pass")),
                      ("cfg4_late", _chain((c4.get("late_packets") or {}).get("full_chain_all_lags_late_packets"), "ms_per_pass"))):
        if blk:
            cfgs[name] = blk
    for name, key, ms in (("cfg5_tx", "roofline_tx", "tx_ms"), ("cfg5_rx", "roofline_rx", "rx_ms")):
        if key in c5:
            cfgs[name] = {"ms": _r(c5.get(ms)), "frac": _r(c5[key].get("frac"))}
    for name, blk in (("cfg3", c3), ("cfg4", c4), ("cfg5", c5)):
        if "error" in blk:
            cfgs[name + "_error"] = str(blk["error"])[:160]
        elif blk.get("dry_run"):
            cfgs[name] = {"dry_run": True, "ms": _r(blk.get("ms")), "n_gpus": blk.get("n_gpus")}
            if "stream_frames_all_ranks" in blk:
                cfgs[name]["stream_frames_all_ranks"] = blk["stream_frames_all_ranks"]
    if cfgs:
        line["configs"] = cfgs
    if isinstance(res.get("shapes"), list):
        line["shapes_min_frac"] = {k: min((s[k] for s in res["shapes"] if isinstance(s.get(k), float)), default=None)
                                   for k in ("tx_frac", "rx_frac", "encode_frac")}
    return line


class Detail:
    """The full record (every block, every leg) goes to a file, rewritten after each block so that a late crash loses nothing."""

    def __init__(self, path, rank):
        self.path = path if rank == 0 else None

    def write(self, res):
        if not self.path:
            return
        try:
            tmp = self.path + ".tmp"
            with open(tmp, "w") as f:
                json.dump(res, f, indent=1)
            os.replace(tmp, self.path)
        except OSError as e:  # a read-only checkout must not cost the headline
            print(f"bench.py: could not write {self.path}: {e}", file=sys.stderr)
            self.path = None


def main():
    a = parse()
    if a.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a)
    if a.dry_run:
        return dry_run(a)
    import numpy as np
    import torch

    from ofdm_amd.dist import Group

    grp = Group()  # torch.distributed over RCCL when launched by torchrun; a no-op single rank otherwise
    world, rank, local = grp.world, grp.rank, grp.local
    if "OFDM_FORCE_DEVICE" in os.environ:  # rehearsal of the N-rank flow on a one-GPU box (with OFDM_DIST_BACKEND=gloo)
        local = int(os.environ["OFDM_FORCE_DEVICE"])
    dist = grp.dist
    n_gpus = world
    if a.gpus != world:  # a silent N=1 number under an N-GPU label is worse than no number
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE {world}", file=sys.stderr)
        grp.close()
        sys.exit(3)
    if local >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} wants cuda:{local} but only {torch.cuda.device_count()} device(s) are visible", file=sys.stderr)
        sys.exit(4)
    torch.cuda.set_device(local)

    from ofdm_amd import api

    syms = 16
    ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True, device=local)
    F = a.frames
    x, payload = synth_cfg2(ctx, torch, F, syms, a.snr_db, seed=rank)
    # The kernel's time comes in two populations, 1.54-1.60 and 1.73-1.76 ms per 1 M frames (profiles/r05_outbuf_*): some
    # GPUs of the pool give the slow figure for EVERY output buffer; on the others it is a property of the buffer (the same value in every
    # round), drawn by buffers allocated before the 10 GB input in about half of the processes and by none of ~40 buffers allocated after it.
    # The box's read-only and write-only ceilings are the same in both populations: the difference is what 5 % of stores cost inside a read
    # stream.  The rows are allocated after the capture (as a receiver that is handed a capture allocates them); the bench takes the buffer
    # it gets, never a second one, and reports which population it drew (roofline.out_buffer_population).
    out = torch.empty((F, syms * ctx.bytes_per_symbol), dtype=torch.uint8, device=ctx.device)
    def step():
        ctx.rx_demod(x, syms_per_frame=syms, out=out)

    step()
    headline_dispatch = ctx.last_dispatch()  # which kernel serves the headline (ofdm_last_dispatch)

    def barrier():
        torch.cuda.synchronize()
        grp.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(a.steps):
        step()
    ev_ms = ctx.timer_stop_ms()  # HIP events on the launch stream, spans exactly the K launches
    barrier()
    dt = time.perf_counter() - t0
    per_rank_ms = grp.gather_floats(ev_ms / a.steps)  # every rank's own kernel time per step
    dt, ev_ms = grp.reduce_max(dt, ev_ms)  # the slowest rank defines the step time
    solo_ms = None
    if n_gpus > 1:  # the N = 1 reference of THIS run: rank 0 repeats the K steps alone while the others idle at a barrier
        if rank == 0:
            ctx.timer_start()
            for _ in range(a.steps):
                step()
            solo_ms = ctx.timer_stop_ms() / a.steps
        barrier()

    # parity at full size through a size-independent property: decoded bytes == transmitted payload (BER)
    nerr = int((out != payload).any(dim=1).sum())
    diff = torch.bitwise_xor(out, payload)
    bits = 0
    for sh in range(8):
        bits += int(((diff >> sh) & 1).sum())
    bits, nerr = (int(v) for v in grp.reduce_sum(bits, nerr))
    ber = bits / (n_gpus * F * payload.shape[1] * 8)

    samples_per_step = F * syms * ctx.S
    value = n_gpus * samples_per_step * a.steps / dt / 1e6
    kern_s = ev_ms / 1e3 / a.steps
    alg_bytes = F * (syms * ctx.S * 8 + syms * ctx.bytes_per_symbol)  # 8 B/sample read + packed bytes written
    # HBM bytes per launch from the committed PMC passes (profiles/r0N_pmc_traffic.json, newest round), scaled to F: collected in
    # separate --pmc runs of tools/pmc_traffic5.py (counters cannot ride along with a timed run), NOT measured in this run
    traffic = traffic_source = None
    for rnd in ("r05", "r04", "r03", "r02"):
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")))["k_demod64"]
            traffic = F * (pm["read_bytes_per_frame"] + pm["write_bytes_per_frame"])
            traffic_source = f"profiles/{rnd}_pmc_traffic.json ({pm.get('counters', 'FETCH_SIZE / WRITE_SIZE')}; separate --pmc passes of tools/pmc_traffic5.py on the same kernel and shape, not measured in this run)"
            break
        except Exception:
            pass
    # What THIS box can do, measured in the same process on the same buffers (boxes of this pool differ by up to 10 % on the
    # write side): the read-only probe in k_demod64's own access pattern, a pure fill, and a device copy.
    box = {}
    try:
        def _t(fn, n=5):
            fn(); torch.cuda.synchronize(); ctx.timer_start()
            for _ in range(n):
                fn()
            return ctx.timer_stop_ms() / n
        rd_ms = _t(lambda: ctx.hbm_read_probe(x, 0))
        scratch = torch.empty(1 << 28, dtype=torch.float32, device=ctx.device)   # 1 GiB
        fill_ms = _t(lambda: scratch.fill_(1.0))
        half = scratch.numel() // 2
        copy_ms = _t(lambda: scratch[:half].copy_(scratch[half:]))
        box = {"box_read_ceiling_gbs": x.numel() * 8 / rd_ms / 1e6,         # algorithmic bytes of the capture / read-only probe time
               "box_read_probe_ms": rd_ms,
               "box_write_ceiling_gbs": scratch.numel() * 4 / fill_ms / 1e6,
               "box_copy_gbs": 2 * half * 4 / copy_ms / 1e6}
        # the kernel's time if it did nothing but its loads at the probe's rate and its stores at the fill's rate, back to back
        ideal_ms = rd_ms + out.numel() / (box["box_write_ceiling_gbs"] * 1e6)
        box["frac_of_box_ceiling"] = ideal_ms / (kern_s * 1e3)
        del scratch
    except Exception as e:  # the headline must survive a failing probe
        box = {"box_probe_error": repr(e)}
    roof = {"bound": "hbm", "kernel": "ofdm::k_demod64<6, true, false, 16> (BPS, GUARD, HK, groups per store burst)" if headline_dispatch == "k_demod64<burst16>" else "ofdm::" + headline_dispatch, "achieved": alg_bytes / kern_s / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": alg_bytes / kern_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
            "kernel_ms": kern_s * 1e3, "algorithmic_bytes_per_launch": alg_bytes, "dispatch": headline_dispatch, **box}
    per_m = kern_s * 1e3 * (1_000_000 / F)   # ms per 1 M frames
    roof["out_buffer_population"] = "fast (<= 1.65 ms per 1 M frames)" if per_m <= 1.65 else "slow (> 1.65 ms per 1 M frames)"
    roof["out_buffer_allocated"] = "after the input capture"
    if traffic:  # the same launch priced on the bytes the counters saw move (the cyclic prefix of a symbol is never fetched)
        roof["frac_moved"] = traffic / kern_s / 1e9 / HBM_PEAK_GBS

    res = {
        "metric": "complex IQ Msamples/s through RX demod", "value": value, "unit": "Msamples/s", "n_gpus": n_gpus,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cfg2: 64-subcarrier 64QAM guard-band frames, RX demod only (CP strip+FFT64+pilot phase+demap)",
                   "frames_per_gpu": F, "symbols_per_frame": syms, "samples_per_frame": syms * ctx.S,
                   "snr_db": a.snr_db, "parallelism": f"frame-index split x{n_gpus}, no collective",
                   "parity": "FFT / CP strip / pilot phase follow src/receiver.rs:99-145; the 64-QAM demap is a north-star extension the reference "
                             "lacks (EXT-1): parity unpinned by the reference, the oracle is the definition"},
        "ber_vs_tx_payload": ber, "frames_with_errors": nerr, "roofline": roof,
        "world_size_seen": world, "backend": grp.backend, "kernel_ms_per_rank": per_rank_ms,
    }
    if solo_ms:  # informational (the driver computes scaling from its own N = 1 run): N ranks' aggregate over N x rank 0 alone
        res["in_run_single_rank_kernel_ms"] = solo_ms
        res["weak_scaling_efficiency_in_run"] = (n_gpus * samples_per_step / (dt / a.steps)) / (n_gpus * samples_per_step / (solo_ms / 1e3))

    if rank == 0 and n_gpus == 1 and not a.no_cpu:
        ncpu = min(a.cpu_frames or 131072, F)
        xh = x[:ncpu].cpu().numpy()
        res["cpu_baseline"] = cpu_baseline_cfg2(xh, syms, out[:ncpu].cpu().numpy(), a.cpu_seconds)
        res["speedup_vs_cpu"] = value / res["cpu_baseline"]["value"] if res["cpu_baseline"]["value"] > 0 else None
        del xh

    # The headline is complete here: it goes to stderr and to the detail file NOW, so that nothing the later blocks do can lose it
    # (the same object, with the other configurations' summaries added, is the one stdout line at the end).
    detail = Detail(a.detail, rank)
    res["detail"] = os.path.relpath(a.detail, ROOT) if os.path.abspath(a.detail).startswith(ROOT) else a.detail
    if rank == 0:
        print("bench.py headline: " + json.dumps(compact_line(res)), file=sys.stderr, flush=True)
        detail.write(res)

    # BASELINE configs 3, 4, 5 at their stated sizes on this GPU (rank 0 of a single-GPU run): each block carries its own
    # roofline, bounded-sample CPU baseline and GPU-vs-CPU equality.  The headline line above must survive their failure.
    # Every rank runs them (its own captures / ring / symbol stream: BASELINE configs[3] and [4] are defined on 8 GPUs), through
    # the same barrier + max-over-ranks timing as the headline; the CPU baselines run on rank 0 of a single-GPU run only.
    if not a.no_cfg3:
        del x, payload, out
        torch.cuda.empty_cache()
        cpu = (not a.no_cpu) and n_gpus == 1
        for name, fn in (("cfg3", lambda: __import__("tools.bench_cfg3", fromlist=["run"]).run(
                                     api, torch, a.cfg3_frames, max(3, a.steps // 4), local, cpu=cpu, grp=grp)),
                         ("cfg5", lambda: __import__("tools.bench_large_n", fromlist=["cfg5"]).cfg5(
                                     a.cfg5_symbols, 5, cpu=cpu, grp=grp, device=local)),
                         ("cfg4", lambda: __import__("tools.bench_large_n", fromlist=["cfg4"]).cfg4(
                                     a.cfg4_ring, a.cfg4_frames, cpu=cpu, grp=grp, device=local))):
            # A rank that fails inside a block would leave the others waiting at that block's barrier: every rank reports
            # whether it got through, and a failure anywhere skips the remaining blocks on all ranks.
            try:
                blk = fn()
                ok = 1.0
            except Exception as e:
                import traceback
                blk = {"error": repr(e), "trace": traceback.format_exc()[-800:]}
                ok = 0.0
            res[name] = blk
            detail.write(res)
            torch.cuda.empty_cache()
            if n_gpus > 1:
                (oks,) = grp.reduce_sum(ok)
                if oks < n_gpus:
                    res[name].setdefault("error", "another rank failed in this block")
                    break
    if rank == 0 and n_gpus == 1 and not a.no_cfg3 and not a.no_host:
        # host buffers in, host buffers out (the reference's own calling convention) and ONE long capture per decode: reported beside
        # the resident numbers, never as `value`
        try:
            res.update(__import__("tools.bench_host", fromlist=["run"]).run(api, torch, local))
        except Exception as e:
            res["h2d_inclusive"] = {"error": repr(e)}
        detail.write(res)
        torch.cuda.empty_cache()
    if rank == 0 and n_gpus == 1 and not a.no_cfg3 and not a.no_shapes:
        # every transform length the library accepts: symbol-stream TX / RX and frame-level encode against their one-pass
        # rooflines (which lengths run shape-specialised kernels)
        try:
            shapes = __import__("tools.bench_shapes", fromlist=["one"])
            # (N = 64 RX is the headline line itself: not repeated here, so that rocprofv3's average for k_demod64 is the
            # average of the timed configuration only)
            res["shapes"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in shapes.one(n, 6, 1 << 27, 3, 8, rx=n != 64).items()}
                             for n in (64, 128, 256, 512, 1024, 2048, 4096)]
        except Exception as e:
            res["shapes"] = {"error": repr(e)}
        torch.cuda.empty_cache()

    if rank == 0:
        detail.write(res)
        line = compact_line(res)
        for drop in ("shapes_min_frac", "configs"):  # never reached with today's blocks; the line must parse whatever they grow into
            if len(json.dumps(line)) >= 4096:
                line.pop(drop, None)
        print(json.dumps(line), flush=True)
    grp.close()


if __name__ == "__main__":
    main()
