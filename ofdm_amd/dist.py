"""Frame-index data parallelism: one process per GPU, no collective on the data path.

Frames are independent (decode is a pure function of its samples, src/receiver.rs:9-96), so rank r of R simply
owns a contiguous frame range; results are concatenated by the host.  torch.distributed (RCCL on GPUs, gloo on
CPU in the tests) is used only for the timing barrier and the max-over-ranks reduction that bench.py reports.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split: rank r gets [r*F/R, (r+1)*F/R); sizes differ by at most one frame."""
    if world <= 0 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad shard request")
    return (n_frames * rank) // world, (n_frames * (rank + 1)) // world


def lag_ranges(n_samples: int, world: int, L: int, W: int):
    """The lag ranges [lag_lo, lag_hi) of one capture for `world` ranks: a contiguous split of its n_samples - W - L + 1 lags whose
    inner cuts are EVEN (a rank's slice then starts on a 16-byte boundary of the fc32 capture, which the LDS-DMA kernels want)."""
    if world <= 0 or L <= 0 or W <= 0:
        raise ValueError("bad halo request")
    valid = max(n_samples - W - L + 1, 0)
    cuts = [0] + [min(valid, ((valid * r) // world) & ~1) for r in range(1, world)] + [valid]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def halo_ranges(n_samples: int, rank: int, world: int, L: int, W: int) -> Tuple[int, int, int, int, int]:
    """Continuous-capture variant of the split (the reference's caller is examples/jetson_rx.rs:16,48-49,86: one
    decode! per 2 M-sample buffer): the Schmidl-Cox search of ONE long capture over `world` ranks with no exchange.

    The capture has `n_samples - W - L + 1` lags.  Rank r owns the contiguous lag range [lag_lo, lag_hi) and reads the samples
    [lag_lo, sample_hi): its own lags, plus W more lags because the detector is threshold-then-peak -- a first crossing at the
    last own lag is followed by a peak window of W lags -- plus the W + L - 1 samples the last of those
    windows spans.  The overlap of 2 W + L - 1 samples is a read-only halo (overhead, not algorithmic bytes).

    Returns (lag_lo, lag_hi, sample_lo, sample_hi, n_lags): the rank works on samples [sample_lo, sample_hi), whose n_lags lags
    are its own_lags = lag_hi - lag_lo lags followed by the W lags of overrun.  A rank's answer counts only if its FIRST CROSSING
    lies among its own lags (a crossing in the overrun belongs to the next rank, which sees that crossing's whole peak window; the
    window of the overrun is cut off at n_lags): search_own_range() does exactly that with any bounded detector, and
    ofdm_sc_correlate_long(lag_lo, lag_hi) is that search on the GPU.  merge_first_detection() then takes the lowest rank's answer,
    which is what one search over the whole capture returns."""
    if world <= 0 or not (0 <= rank < world) or L <= 0 or W <= 0:
        raise ValueError("bad halo request")
    valid = n_samples - W - L + 1
    if valid <= 0:
        return 0, 0, 0, max(n_samples, 0), 0
    lag_lo, lag_hi = lag_ranges(n_samples, world, L, W)[rank]
    n_lags = min(lag_hi - lag_lo + W, valid - lag_lo)  # the peak window may run W lags past the own range, never past the capture
    if lag_hi == lag_lo:
        n_lags = 0
    sample_hi = min(n_samples, lag_lo + n_lags + W + L - 1)
    return lag_lo, lag_hi, lag_lo, sample_hi, n_lags


def search_own_range(search, own_lags: int, n_lags: int):
    """A rank's part of the split search with any bounded threshold-then-peak detector `search(n_lags) -> (d_hat, f_delta, metric)`
    over the rank's samples (d_hat < 0: none): first over the OWN lags only -- that decides whether the first crossing is the
    rank's -- and, if it is, again over own + W lags so that the peak window is whole.  Returns (d_hat, f_delta, metric) relative
    to the rank's first lag, d_hat = -1 when the rank has no crossing of its own."""
    if own_lags <= 0 or n_lags <= 0:
        return -1, 0.0, 0.0
    d, fd, m = search(min(own_lags, n_lags))
    if d is None or d < 0:
        return -1, 0.0, 0.0
    if n_lags > own_lags:
        d, fd, m = search(n_lags)
    return int(d), float(fd), float(m)


def merge_first_detection(detections):
    """detections: per rank, in rank order, (lag_lo, lag_hi, d_hat, f_delta, metric) with d_hat relative to lag_lo (-1 = none),
    each the result of search_own_range() / ofdm_sc_correlate_long(): a detection whose first crossing lies in [lag_lo, lag_hi).
    The capture's detection is the one of the LOWEST rank that has any (threshold-then-peak: the first crossing wins, and that
    rank evaluated its whole peak window).  A d_hat whose crossing may lie in the rank's overrun -- a plain bounded search over
    own + W lags -- must NOT be passed here: its window is cut off at the end of the rank's samples.  Returns (d_hat, f_delta,
    metric) with d_hat a lag of the whole capture, or (-1, 0.0, 0.0)."""
    for lag_lo, lag_hi, d_hat, f_delta, metric in detections:
        if d_hat is not None and d_hat >= 0:
            return lag_lo + int(d_hat), float(f_delta), float(metric)
    return -1, 0.0, 0.0


def env_world() -> Tuple[int, int, int]:
    """(world_size, rank, local_rank) from the torchrun environment (1, 0, 0 when launched plainly)."""
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


class Group:
    """Thin wrapper over torch.distributed for the bench / tests: barrier, max and sum of a few scalars, gather."""

    def __init__(self, backend: Optional[str] = None, device=None):
        import torch

        self.torch = torch
        self.world, self.rank, self.local = env_world()
        self.dist = None
        self.device = device
        self.backend = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if backend is None:  # RCCL on GPUs; OFDM_DIST_BACKEND=gloo rehearses the multi-rank flow on one GPU / on CPU
                backend = os.environ.get("OFDM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            kw = {}
            if backend == "nccl":
                torch.cuda.set_device(self.local)
                kw["device_id"] = torch.device("cuda", self.local)
                self.device = torch.device("cuda", self.local)
            if not dist.is_initialized():
                import datetime

                # a bounded wait: a rank that dies between two collectives must end the job with an error, not hang it
                kw["timeout"] = datetime.timedelta(seconds=int(os.environ.get("OFDM_DIST_TIMEOUT_S", "300")))
                dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist
            self.backend = backend
        if self.device is None:
            self.device = torch.device("cpu")

    def barrier(self):
        if self.torch.cuda.is_available() and self.device.type == "cuda":
            self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()

    def reduce_max(self, *vals: float):
        if self.dist is None:
            return list(vals)
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t]

    def reduce_sum(self, *vals: float):
        if self.dist is None:
            return list(vals)
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t]

    def gather_floats(self, val: float):
        """Every rank's value in rank order, on every rank (per-rank step times in the bench report)."""
        if self.dist is None:
            return [float(val)]
        t = self.torch.zeros(self.world, dtype=self.torch.float64, device=self.device)
        t[self.rank] = float(val)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t]

    def gather_bytes(self, payload: bytes):
        """All ranks' byte strings in rank order (rank 0 only; others get None).  Off the timed path."""
        if self.dist is None:
            return [payload]
        out = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(payload, out, dst=0)
        return out

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
