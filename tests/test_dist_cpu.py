"""Multi-rank path on CPU: world_size-2 gloo.  Frames are index-split across ranks with no data-path collective
(SURVEY.md 8e); the only communication is the timing barrier / max reduction and an off-path gather.  The per-shard
worker here is the CPU oracle (tests may use it); the property checked is that the sharded result equals the
single-process result byte for byte."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from ofdm_amd.dist import shard_range

    for n in (0, 1, 7, 8, 1000, 1_000_003):
        for world in (1, 2, 3, 8):
            cuts = [shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _split_search(orc, cap, world, L, reps):
    """The `world`-way halo split of one capture with the oracle's bounded detector as every rank's worker."""
    from ofdm_amd.dist import halo_ranges, merge_first_detection, search_own_range

    n, W = cap.size, reps * L
    dets, covered = [], 0
    for r in range(world):
        lag_lo, lag_hi, s_lo, s_hi, n_lags = halo_ranges(n, r, world, L, W)
        assert lag_lo == covered and s_lo == lag_lo and s_hi <= n
        assert r == 0 or lag_lo % 2 == 0                                         # inner cuts are even: 16-byte aligned slices
        covered = lag_hi
        assert s_hi - s_lo <= (lag_hi - lag_lo) + 2 * W + L - 1                  # own lags + the halo, never more

        def search(k, lo=s_lo, hi=s_hi):
            d, _, m, fd = orc.sc_sync(cap[lo:hi], L=L, window_reps=reps, n_lags=k, threshold=0.5)
            return d, fd, m

        dets.append((lag_lo, lag_hi) + search_own_range(search, lag_hi - lag_lo, n_lags))
    assert covered == max(n - W - L + 1, 0)
    return merge_first_detection(dets)


def test_halo_split_of_a_long_capture_equals_the_whole_search(orc):
    """SURVEY.md 8(e), continuous-stream variant (the reference's caller: examples/jetson_rx.rs:16,48-49,86, one decode! per
    long buffer): the Schmidl-Cox search of ONE capture split over `world` ranks by lag range with a read-only halo of
    2 W + L - 1 samples and no exchange.  Per shard the worker is the oracle's sc_sync through dist.search_own_range (own lags
    first, the whole window only for a crossing among them); the merged detection must be exactly the whole capture's -- for a
    two-frame capture, a frame that straddles a shard boundary, a crossing in the last lags of a shard (peak in the next one's
    range), and a noise-only capture."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ofdm_amd.dist import halo_ranges
    from util import through_channel

    L, reps = 80, 3
    W = reps * L
    rng = np.random.default_rng(2026)
    tx = [orc.encode(bytes(rng.integers(0, 256, 300, dtype=np.uint8)), True, orc.QAM64, 64) for _ in range(2)]
    flen = tx[0].size
    n = 9000
    for case, starts in (("two frames", (700, 5200)), ("second half only", (4600,)), ("straddles the cut", (4300 - 120,)),
                         ("late", (n - flen - 30,)), ("noise only", ())):
        cap = np.zeros(n, np.complex128)
        for t, st in zip(tx, starts):
            cap += through_channel(orc, rng, t, n, st, 0.004, snr_db=None, data_start=800).astype(np.complex128)
        cap += 0.004 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        cap = cap.astype(np.complex64).astype(np.complex128)
        want_d, _, want_m, want_fd = orc.sc_sync(cap, L=L, window_reps=reps, n_lags=0, threshold=0.5)
        assert (want_d >= 0) == bool(starts), case
        for world in (1, 2, 3, 4, 7):
            got_d, got_fd, got_m = _split_search(orc, cap, world, L, reps)
            assert got_d == want_d, (case, world, got_d, want_d)
            if want_d >= 0:
                assert abs(got_fd - want_fd) <= 1e-12 and abs(got_m - want_m) <= 1e-12, (case, world)
    assert halo_ranges(100, 0, 2, 80, 240) == (0, 0, 0, 100, 0)                   # no lag fits: nothing to search
    with pytest.raises(ValueError):
        halo_ranges(1000, 2, 2, 80, 240)


def test_halo_split_with_the_crossing_swept_across_a_cut(orc):
    """The case round 3's merge got wrong: a rank whose FIRST crossing falls into its W-lag overrun sees that
    crossing's peak window cut off at the end of its samples, so its answer must not win.  The frame start is swept in steps of
    20 samples so that the crossing moves from W + L lags before a cut to W lags behind it, for two- and four-way splits."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ofdm_amd.dist import lag_ranges
    from util import through_channel

    L, reps = 80, 3
    W = reps * L
    rng = np.random.default_rng(77)
    tx = orc.encode(bytes(rng.integers(0, 256, 300, dtype=np.uint8)), True, orc.QAM64, 64)
    n = 9000
    noise = 0.004 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    overrun = 0
    for world in (2, 4):
        cut = lag_ranges(n, world, L, W)[world // 2 - 1][1]                      # the cut nearest the middle of the capture
        for st in range(cut - W - 2 * L, cut + W, 20):                          # the crossing follows the frame start by ~L + 10 lags
            cap = through_channel(orc, rng, tx, n, st, 0.004, snr_db=None, data_start=800).astype(np.complex128) + noise
            cap = cap.astype(np.complex64).astype(np.complex128)
            want_d, _, want_m, want_fd = orc.sc_sync(cap, L=L, window_reps=reps, n_lags=0, threshold=0.5)
            assert want_d >= 0
            got_d, got_fd, got_m = _split_search(orc, cap, world, L, reps)
            assert got_d == want_d, (world, st, cut, got_d, want_d)
            assert abs(got_fd - want_fd) <= 1e-12 and abs(got_m - want_m) <= 1e-12
            # how often the old rule (lowest rank with ANY detection over own + W lags) would have been wrong here
            lo, hi = lag_ranges(n, world, L, W)[world // 2 - 1]
            d_old, _, _, _ = orc.sc_sync(cap[lo:min(n, hi + 2 * W + L - 1)], L=L, window_reps=reps, n_lags=hi - lo + W, threshold=0.5)
            overrun += d_old >= 0 and lo + d_old != want_d
    assert overrun > 0                                                            # the sweep does reach the case


class _FakeCtx:
    """Stands in for api.Context in the timing helper: the 'kernel' is a sleep, the 'event timer' the wall clock."""

    def timer_start(self):
        import time
        self.t0 = time.perf_counter()

    def timer_stop_ms(self):
        import time
        return (time.perf_counter() - self.t0) * 1e3


class _FakeTorch:
    class cuda:
        @staticmethod
        def synchronize():
            pass


def _timing_worker(rank, world, port, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import time
    from ofdm_amd.dist import Group
    from tools import rank_timing

    g = Group(backend="gloo")
    ms, per_rank, r = rank_timing.timed(_FakeCtx(), _FakeTorch, lambda: time.sleep(0.01 * (rank + 1)) or rank, 3, g)
    if rank == 0:
        q.put((ms, per_rank, r))
    g.close()


def test_config_block_timing_is_max_over_ranks():
    """tools/rank_timing.timed -- the timed region of bench.py's config 3 / 4 / 5 blocks under --gpus N: every rank times its
    own K launches between barriers, the job's step time is the slowest rank's, every rank's own time is reported."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 400)
    procs = [ctx.Process(target=_timing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ms, per_rank, r = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len(per_rank) == 2 and ms == max(per_rank) and per_rank[1] > per_rank[0] and per_rank[0] >= 9.0 and r == 0


def _failing_timing_worker(rank, world, port, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from ofdm_amd.dist import Group
    from tools import rank_timing

    g = Group(backend="gloo")
    calls = [0]

    def fn():
        calls[0] += 1
        if rank == 1 and calls[0] == 2:   # the warm-up passes, the first timed step dies on rank 1 only
            raise ValueError("boom on rank 1")
        return rank

    try:
        rank_timing.timed(_FakeCtx(), _FakeTorch, fn, 3, g)
        q.put((rank, "no error"))
    except ValueError as e:
        q.put((rank, "own: " + str(e)))
    except rank_timing.RankFailed as e:
        q.put((rank, "peer: " + str(e)[:40]))
    (n,) = g.reduce_sum(1.0)              # the ranks are still in step: the next collective matches
    q.put((rank, f"after: {int(n)}"))
    g.close()


def test_config_block_timing_survives_a_failing_rank():
    """a rank that raises inside a timed block must not leave its peers in a mismatched collective.  timed() catches the
    error, still takes part in the region's barrier / gather (NaN for its time) and raises afterwards; the healthy rank raises
    RankFailed; both reach the block's closing reduction."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 400)
    procs = [ctx.Process(target=_failing_timing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(4))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][0] == 0 and got[0][1].startswith("after: 2") or got[1][1].startswith("after: 2")
    msgs = {(r, m.split(":")[0]) for r, m in got}
    assert msgs == {(0, "peer"), (0, "after"), (1, "own"), (1, "after")}, got


def _worker(rank, world, port, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as orc
    from ofdm_amd.dist import Group, shard_range
    from util import make_symbols, wide

    g = Group(backend="gloo")
    assert (g.world, g.rank) == (world, rank)
    rng = np.random.default_rng(123)                      # every rank regenerates the same batch ...
    x, data = make_symbols(orc, rng, 24 * 4, 64, True, orc.QAM64, snr_db=32.0)
    frames = x.reshape(24, 4 * 80)
    lo, hi = shard_range(24, rank, world)                 # ... and decodes only its own frame range
    g.barrier()
    mine = orc.rx_demod(wide(frames[lo:hi].reshape(-1)), 64, True, orc.QAM64)
    tmax, = g.reduce_max(float(rank + 1))                 # the bench's max-over-ranks timing reduction
    total, = g.reduce_sum(float(hi - lo))
    parts = g.gather_bytes(mine)
    if rank == 0:
        q.put((b"".join(parts), data, tmax, total))
    g.close()


def test_two_rank_index_split_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    joined, data, tmax, total = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert joined == data          # concatenated shards == the whole batch decoded in one piece
    assert tmax == 2.0 and total == 24.0


def _run_bench(*args, env=None):
    import json
    import subprocess

    e = dict(os.environ)
    e.pop("WORLD_SIZE", None), e.pop("RANK", None), e.pop("LOCAL_RANK", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, capture_output=True, text=True,
                       timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return r.returncode, [json.loads(ln) for ln in lines if ln.lstrip().startswith("{")], lines, r.stderr


def test_bench_gpus_n_starts_n_ranks():
    """`python bench.py --gpus 2` launched plainly (no torchrun) must start two ranks itself and report n_gpus: 2
    (VERDICT r1 item 1).  --dry-run keeps the GPU out of it: launcher, rendezvous (gloo), barrier and reductions only."""
    import json
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        det = os.path.join(td, "detail.json")
        rc, recs, lines, err = _run_bench("--gpus", "2", "--dry-run", "--steps", "2", "--detail", det, env={"OFDM_DIST_BACKEND": "gloo"})
        assert rc == 0, err
        full = json.load(open(det))       # the full record: per-rank times of every block
    assert len(lines) == 1 and len(recs) == 1, lines   # ONE JSON line on stdout
    rec = recs[0]
    assert len(lines[0]) < 4096           # ... that the driver can parse (a 23 KB line came back `parsed: null`)
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["world_size_seen"] == 2
    assert len(full["ms_per_step_per_rank"]) == 2 and rec["ms_per_step"] >= max(full["ms_per_step_per_rank"]) - 1e-5
    # configs 3, 4 and 5 run on every rank too (BASELINE configs[3] / [4] are 8-GPU workloads): their blocks carry n_gpus: 2,
    # per-rank times, and config 4's stream is frame-sharded over the ranks with nothing lost
    for name in ("cfg3", "cfg4", "cfg5"):
        assert full[name]["n_gpus"] == 2 and len(full[name]["ms_per_rank"]) == 2 and full[name]["ms"] == max(full[name]["ms_per_rank"])
        assert rec["configs"][name]["n_gpus"] == 2 and abs(rec["configs"][name]["ms"] - full[name]["ms"]) < 1e-3
    assert rec["configs"]["cfg4"]["stream_frames_all_ranks"] == full["cfg4"]["stream_frames"] == 10_000_000


def test_bench_line_stays_compact_with_every_block_filled_in():
    """The compact line built from a full-size record (every block of round 4's 23 KB line) stays under 4 KB, parses, and carries
    the objects the contract names: roofline, cpu_baseline and ms + fraction per configuration."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    line = json.dumps(bench.compact_line(full))
    assert len(line) < 4096, len(line)
    rec = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_launch"):
        assert k in rec["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in rec["cpu_baseline"], k
    assert set(rec["configs"]) >= {"cfg3", "cfg3_late", "cfg4", "cfg4_late", "cfg5_tx", "cfg5_rx"}
    assert all("ms" in v and "frac" in v for v in rec["configs"].values())
    assert "model" not in rec["config"] and rec["config"]["workload"].startswith("cfg2")


def test_bench_gpus_1_is_single_rank():
    rc, recs, lines, err = _run_bench("--gpus", "1", "--dry-run", "--steps", "2")
    assert rc == 0, err
    assert recs[0]["n_gpus"] == 1 and recs[0]["ranks_seen"] == 1


def test_bench_refuses_a_mismatched_world():
    """Under a launcher that brought up fewer ranks than --gpus asks for, bench.py exits non-zero instead of printing
    an N=1 number under an N-GPU label."""
    rc, recs, lines, err = _run_bench("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and not recs


def test_bench_ends_all_ranks_when_one_dies():
    """A rank that dies before the rendezvous must not leave the launcher waiting for the others' rendezvous timeout: the
    launcher ends the remaining ranks and exits non-zero within seconds."""
    import time
    t0 = time.time()
    rc, recs, lines, err = _run_bench("--gpus", "2", "--dry-run", env={"OFDM_BENCH_FAIL_RANK": "1"})
    assert rc != 0 and not recs
    assert time.time() - t0 < 60.0


def test_build_lock_serialises_concurrent_builders():
    """Two processes calling ofdm_amd.build.build() at once must not overlap inside the locked region."""
    import subprocess

    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from ofdm_amd import build as b\n"
            "with b._BuildLock():\n"
            "    t0 = time.time(); time.sleep(0.6); t1 = time.time()\n"
            "print(t0, t1)\n") % ROOT
    ps = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True) for _ in range(2)]
    spans = sorted(tuple(float(v) for v in p.communicate(timeout=60)[0].split()) for p in ps)
    assert all(p.returncode == 0 for p in ps)
    assert spans[1][0] >= spans[0][1] - 1e-3, spans   # the second holder entered after the first left
