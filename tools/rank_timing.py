"""Timed regions shared by bench.py's config blocks: HIP events on the context's stream around exactly K launches, bracketed by a
barrier + device synchronisation on both sides when several ranks run (one process per GPU, every rank its own data: frames
are independent, so there is no data-path collective -- ofdm_amd/dist.py).  The step time of the job is the MAX over ranks.

Failure-aware: a rank whose `fn` raises still takes part in every collective of the region (with a NaN in place of its time) and
raises afterwards, and so does every other rank ("another rank failed") -- the ranks never sit in mismatched collectives."""


class RankFailed(RuntimeError):
    pass


def timed(ctx, torch, fn, steps, grp=None):
    """-> (ms per step, max over ranks; every rank's own ms per step; the last call's result)."""
    multi = grp is not None and grp.world > 1
    err, r, ms = None, None, float("nan")
    try:
        r = fn()  # warm-up: workspaces are grown here, never inside the timed loop
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001 -- reported after the collectives below
        err = e
    if multi:
        grp.barrier()
    if err is None:
        try:
            ctx.timer_start()
            for _ in range(steps):
                r = fn()
            ms = ctx.timer_stop_ms() / steps
        except Exception as e:  # noqa: BLE001
            err = e
    if not multi:
        if err is not None:
            raise err
        return ms, [ms], r
    grp.barrier()
    per_rank = grp.gather_floats(ms)
    if err is not None:
        raise err
    if any(v != v for v in per_rank):
        raise RankFailed(f"another rank failed in this block (per-rank ms: {per_rank})")
    return max(per_rank), per_rank, r


def world_of(grp):
    return 1 if grp is None else grp.world
