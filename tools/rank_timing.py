"""Timed regions shared by bench.py's config blocks: HIP events on the context's stream around exactly K launches, bracketed by a
barrier + device synchronisation on both sides when several ranks run (one process per GPU, every rank its own data: frames
are independent, so there is no data-path collective -- ofdm_amd/dist.py).  The step time of the job is the MAX over ranks."""


def timed(ctx, torch, fn, steps, grp=None):
    """-> (ms per step, max over ranks; every rank's own ms per step; the last call's result)."""
    r = fn()  # warm-up: workspaces are grown here, never inside the timed loop
    torch.cuda.synchronize()
    if grp is not None:
        grp.barrier()
    ctx.timer_start()
    for _ in range(steps):
        r = fn()
    ms = ctx.timer_stop_ms() / steps
    if grp is None or grp.world == 1:
        return ms, [ms], r
    grp.barrier()
    per_rank = grp.gather_floats(ms)
    return max(per_rank), per_rank, r


def world_of(grp):
    return 1 if grp is None else grp.world
