#!/usr/bin/env python3
"""bench.py -- complex IQ Msamples/s through RX demod on N MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic, HBM-resident input:
  workload cfg2 (BASELINE configs[1]): 1 M frames x 16 OFDM symbols x 80 samples (64 carriers, 64-QAM, guard
  bands on), RX demod only = CP strip + FFT64 + pilot phase + hard demap -> 36 packed bytes per symbol.
One process per GPU; frames are independent, so ranks hold disjoint frame ranges (index split, weak scaling)
and the data path has no collective.  torch.distributed (RCCL) is only the barrier / max-time reduction.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and `cpu_baseline` objects.
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
    ap.add_argument("--no-cfg3", action="store_true", help="skip the config-3 (full RX chain / Schmidl-Cox) report")
    ap.add_argument("--cfg3-frames", type=int, default=262_144)
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
    out0 = procs[0].stdout.read().decode()  # rank 0 prints the one JSON line (after the closing barrier)
    rc = 0
    deadline = time.time() + 120
    for p in procs:
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()  # exactly the PID started above
            p.wait()
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


def cpu_baseline_cfg2(x_host, syms, n_threads):
    """The oracle (kind "port": the reference cannot be built here) timed on the host cores on a bounded sample."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc

    orc.lib()
    F = x_host.shape[0]
    xs = [np.ascontiguousarray(x_host[i::n_threads].reshape(-1)).astype(np.complex128) for i in range(n_threads)]

    def work(a):
        return orc.rx_demod(a, 64, True, orc.QAM64)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(n_threads) as ex:  # ctypes releases the GIL inside the C oracle
        outs = list(ex.map(work, xs))
    dt = time.perf_counter() - t0
    return F * syms * 80 / dt / 1e6, dt, outs


def dry_run(a):
    """The N-rank plumbing without the GPU: rendezvous, barrier, max / sum reductions, one JSON line from rank 0."""
    from ofdm_amd.dist import Group

    grp = Group()
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
    if grp.rank == 0:
        print(json.dumps({"metric": "complex IQ Msamples/s through RX demod", "value": None, "unit": "Msamples/s",
                          "n_gpus": grp.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps,
                          "dry_run": True, "ranks_seen": int(ranks), "world_size_seen": grp.world,
                          "backend": grp.backend, "ms_per_step_per_rank": per_rank}))
    grp.close()


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
    out = torch.empty((F, syms * ctx.bytes_per_symbol), dtype=torch.uint8, device=ctx.device)

    def step():
        ctx.rx_demod(x, syms_per_frame=syms, out=out)

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
    traffic = None  # HBM bytes per launch from the committed PMC passes (profiles/r01_pmc_traffic.json), scaled to F
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["k_demod64"]
        traffic = F * (pm["read_bytes_per_frame"] + pm["write_bytes_per_frame"])
    except Exception:
        pass
    roof = {"bound": "hbm", "kernel": "ofdm::k_demod64<6, true, false>", "achieved": alg_bytes / kern_s / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": alg_bytes / kern_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
            "kernel_ms": kern_s * 1e3, "algorithmic_bytes_per_launch": alg_bytes}

    res = {
        "metric": "complex IQ Msamples/s through RX demod", "value": value, "unit": "Msamples/s", "n_gpus": n_gpus,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cfg2: 64-subcarrier 64QAM guard-band frames, RX demod only (CP strip+FFT64+pilot phase+demap)",
                   "frames_per_gpu": F, "symbols_per_frame": syms, "samples_per_frame": syms * ctx.S,
                   "snr_db": a.snr_db, "parallelism": f"frame-index split x{n_gpus}, no collective"},
        "ber_vs_tx_payload": ber, "frames_with_errors": nerr, "roofline": roof,
        "world_size_seen": world, "backend": grp.backend, "kernel_ms_per_rank": per_rank_ms,
    }
    if solo_ms:  # informational (the driver computes scaling from its own N = 1 run): N ranks' aggregate over N x rank 0 alone
        res["in_run_single_rank_kernel_ms"] = solo_ms
        res["weak_scaling_efficiency_in_run"] = (n_gpus * samples_per_step / (dt / a.steps)) / (n_gpus * samples_per_step / (solo_ms / 1e3))

    if rank == 0 and n_gpus == 1 and not a.no_cpu:
        ncores = os.cpu_count() or 1
        nthreads = max(1, min(ncores, 64))
        ncpu = a.cpu_frames or 4000 * nthreads
        ncpu = min(ncpu, F)
        xh = x[:ncpu].cpu().numpy()
        msps, cdt, outs = cpu_baseline_cfg2(xh, syms, nthreads)
        # the same sample on the GPU must give the same bytes as the CPU path (identical decoded BER)
        gpu_bytes = out[:ncpu].cpu().numpy()
        same = all(bytes(gpu_bytes[i::nthreads].reshape(-1)) == outs[i] for i in range(nthreads))
        res["cpu_baseline"] = {"value": msps, "unit": "Msamples/s", "cores": nthreads, "kind": "port",
                               "sample": f"first {ncpu} frames of the same batch ({ncpu * syms * 80} samples), "
                                         f"oracle rx_demod (f64), {cdt:.1f} s wall", "gpu_bytes_equal_cpu_bytes": bool(same)}
        res["speedup_vs_cpu"] = value / msps if msps > 0 else None

    if rank == 0 and n_gpus == 1 and not a.no_cfg3:
        try:
            from tools import bench_cfg3

            del x, payload, out
            torch.cuda.empty_cache()
            res["cfg3"] = bench_cfg3.run(api, torch, a.cfg3_frames, max(3, a.steps // 4), local)
        except Exception as e:  # the headline number must not be lost to the secondary report
            res["cfg3"] = {"error": repr(e)}

    if rank == 0 and n_gpus == 1 and not a.no_cfg3:
        try:  # configs 4 and 5 on this GPU (N = 1024 chain, N = 4096 TX + RX): a few seconds, informational
            from tools import bench_large_n

            torch.cuda.empty_cache()
            res["cfg5"] = bench_large_n.cfg5(steps=3)
            res["cfg4"] = bench_large_n.cfg4(steps=3)
        except Exception as e:
            res["cfg45_error"] = repr(e)

    if rank == 0:
        print(json.dumps(res))
    grp.close()


if __name__ == "__main__":
    main()
