"""CPU-baseline legs of bench.py: the oracle (kind "port" -- the reference is nightly Rust with un-vendored dependencies and
cannot be built here, oracle/Makefile) timed on the host cores on a BOUNDED sample of the same workload.

Three variants, as SURVEY.md 8(d) asks:
  (i)   ref-faithful, one thread: twiddles from libm on every transform, like the reference's FftPlanner-per-call
        (src/signals/mod.rs:41-58);
  (ii)  ref-optimised, one thread: cached twiddle tables;
  (iii) (ii) on all host cores (one block of the sample per thread; ctypes releases the GIL inside the C oracle).
Each leg runs for about `target_s` seconds of wall time: the bounded sample is passed over `reps` times (reps sized from a
one-pass pilot), so the memory footprint stays that of the sample.  Only bench.py and its tools import this."""
import math
import os
import time
from concurrent.futures import ThreadPoolExecutor


def host_threads(cap=None):
    """All host cores (hardware_concurrency); callers bound it only by the number of work units."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))  # the cores this process may actually run on (cgroup / taskset)
    except (AttributeError, OSError):
        pass
    return max(1, n if cap is None else min(n, cap))


def _pass(fn, blocks, reps):
    def work(b):
        out = None
        for _ in range(reps):
            out = fn(b)
        return out

    t0 = time.perf_counter()
    if len(blocks) == 1:
        outs = [work(blocks[0])]
    else:
        with ThreadPoolExecutor(len(blocks)) as ex:
            outs = list(ex.map(work, blocks))
    return time.perf_counter() - t0, outs


def timed(fn, blocks, units_per_pass, target_s=2.5, max_reps=4096):
    """fn(block) -> result for ONE block; blocks = one per thread.  Returns (record, results of the last pass)."""
    dt1, outs = _pass(fn, blocks, 1)
    reps = int(min(max_reps, max(1, math.ceil(target_s / max(dt1, 1e-6)))))
    if reps > 1:
        dt, outs = _pass(fn, blocks, reps)
    else:
        dt = dt1
    return {"value": units_per_pass * reps / dt, "seconds": dt, "passes_over_sample": reps, "threads": len(blocks)}, outs
