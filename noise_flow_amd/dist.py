"""Data-parallel evaluation over the patch index range (one process per GPU).

The reference has NO multi-device path for Noise Flow (``--num_gpus`` is parsed
and never read, ``sidd/ArgParser.py:108-109``); patches are independent in eval
mode, so the only exchange this path needs is the final reduction of the epoch
statistics that ``train_noise_flow.py:402-407`` averages on the host:

    (Σ_b nll_b, Σ_b sd_b, count)  --one all-reduce(sum) of 3 fp64 scalars-->  means

``backend="nccl"`` is RCCL over xGMI on ROCm; the message is 24 bytes, i.e. pure
latency — it is issued ONCE per evaluation, never per minibatch.  Sampling needs
no collective at all (outputs stay sharded).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

from .patches import shard_range  # noqa: F401  (re-exported)


def allreduce_sums(sums, group=None):
    """In-place SUM all-reduce of the ``float64[3]`` accumulator (device tensor for
    RCCL, CPU tensor for gloo).  No-op when torch.distributed is not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums


def run_shard(eval_chunk: Callable[[int, int, object], object], start: int, stop: int, chunk: int, sums):
    """One rank's share of one evaluation: patches [start, stop) in chunks, then the evaluator's
    ``finish`` hook (the HIP path accumulates in slots and folds once).  No collective."""
    k = start
    while k < stop:
        n = min(chunk, stop - k)
        eval_chunk(k, n, sums)
        k += n
    finish = getattr(eval_chunk, "finish", None)
    if finish is not None:
        finish(sums)
    return sums


def evaluate_sharded(eval_chunk: Callable[[int, int, object], object], n_total: int, chunk: int,
                     rank: int, world: int, sums, group=None) -> Tuple[float, float, int]:
    """Evaluate patches [0, n_total) split in contiguous blocks over ``world`` ranks.

    ``eval_chunk(first_patch, count, sums)`` must ADD ``(Σ nll, Σ sd, count)`` of
    patches ``first_patch … first_patch+count-1`` into ``sums`` (for the HIP path:
    ``NoiseFlow.nll_sums`` on ``synth_patches(seed, first_patch, count)``).
    Returns the global ``(mean_nll, mean_sd_z, n)`` — identical for every ``world``.
    """
    start, stop = shard_range(n_total, rank, world)
    run_shard(eval_chunk, start, stop, chunk, sums)
    allreduce_sums(sums, group)
    s = sums.detach().cpu().numpy()
    n = int(round(float(s[2])))
    if n != n_total:
        raise RuntimeError("sharded evaluation covered %d patches, expected %d" % (n, n_total))
    return float(s[0] / s[2]), float(s[1] / s[2]), n


def timed_sharded_evaluations(eval_chunk: Callable[[int, int, object], object], n_total: int, chunk: int, rank: int,
                              world: int, steps: int, warmup: int, new_sums: Callable[[], object],
                              sync: Optional[Callable[[], None]] = None, group=None,
                              on_step: Optional[Callable[[int, str], None]] = None) -> dict:
    """The benchmark form of :func:`evaluate_sharded` (``bench.py --gpus N``, BASELINE configs[3]).

    Runs ``warmup`` untimed and then ``steps`` timed COMPLETE evaluations of the patch range
    ``[0, n_total)``: every rank evaluates its ``shard_range`` block and each evaluation ends with
    its ONE all-reduce of ``(Σ nll, Σ sd, count)``.  Nothing is read back and no barrier is issued
    inside the timed region (the results stay on the device until the clock has stopped), so
    consecutive evaluations pipeline on the device; the region is bracketed by ``sync`` +
    ``barrier`` + ``sync`` before and ``sync`` after, as the driver's contract asks, and the caller
    takes the max of ``elapsed`` over ranks.  ``on_step(i, "begin"|"end")`` (timed steps only) lets
    the caller record per-evaluation device events around the rank's own work, before the
    collective.  Returns ``{"elapsed": seconds, "results": [(mean_nll, mean_sd, n)] * steps}`` —
    every result is checked to have covered exactly ``n_total`` patches."""
    import time
    import torch.distributed as dist
    start, stop = shard_range(n_total, rank, world)
    have_pg = dist.is_available() and dist.is_initialized()
    sync = sync or (lambda: None)

    def one(i, timed):
        sums = new_sums()
        if timed and on_step is not None:
            on_step(i, "begin")
        run_shard(eval_chunk, start, stop, chunk, sums)
        if timed and on_step is not None:
            on_step(i, "end")
        allreduce_sums(sums, group)
        return sums

    for i in range(warmup):
        one(i, False)
    sync()
    if have_pg:
        dist.barrier(group=group)
    sync()
    t0 = time.perf_counter()
    outs = [one(i, True) for i in range(steps)]
    sync()
    elapsed = time.perf_counter() - t0
    results = []
    for sums in outs:
        s = sums.detach().cpu().numpy()
        n = int(round(float(s[2])))
        if n != n_total:
            raise RuntimeError("sharded evaluation covered %d patches, expected %d" % (n, n_total))
        results.append((float(s[0] / s[2]), float(s[1] / s[2]), n))
    return {"elapsed": elapsed, "results": results, "shard": (start, stop)}


def flow_eval_chunk(model, seed: int, cond=( [0.0], [0.0], [100.0], [2.0]), height: int = 32, width: int = 32,
                    n_streams: int = 2):
    """The HIP-path ``eval_chunk`` for :func:`evaluate_sharded` on synthetic patches.

    Consecutive chunks alternate between ``n_streams`` HIP streams: a chunk of 1 024 patches fills the
    GPU exactly once (1 024 resident workgroups), so on ONE stream every launch pays its own ramp-up
    and drain; on two, the next launch's workgroups take the slots the previous one frees and the
    evaluation runs at the steady-state rate (measured: 53.4 → 46.0 µs per 1 024 patches).  All
    chunks add into one slotted accumulator (atomics), folded once by ``finish``."""
    import torch
    from .patches import synth_patches
    nlf0, nlf1, iso, cam = cond
    dev = model._dev.device
    wide = model.new_sums()   # slotted accumulator of this evaluation (no same-line atomics)
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, int(n_streams)))]
    state = {"i": 0, "dirty": False}

    def run(first, count, sums):
        s = streams[state["i"] % len(streams)]
        state["i"] += 1
        if not state["dirty"]:   # the accumulator was created / zeroed on the caller's stream
            for t in streams:
                t.wait_stream(torch.cuda.current_stream(dev))
            state["dirty"] = True
        with torch.cuda.stream(s):
            x, y = synth_patches(seed, first, count, height, width, device=dev.index)
            model.nll_sums(x, y, nlf0, nlf1, iso, cam, wide)
        return sums

    def finish(sums):
        """Join the streams and fold the slotted accumulator into the plain triple (once, before the
        all-reduce)."""
        cur = torch.cuda.current_stream(dev)
        for t in streams:
            cur.wait_stream(t)
        model.fold_sums(wide, out=sums)
        wide.zero_()
        state["dirty"] = False
        return sums

    run.finish = finish
    return run


class ResidentShard:
    """One rank's block of the synthetic patch range kept in HBM for repeated evaluation.

    2^20 patches of 32x32x4 are 34 GB of ``x`` + ``y`` — a fraction of the 288 GB of one MI355X — so the
    benchmark of BASELINE configs[3] holds the whole shard resident (the timed region then starts with
    its inputs in HBM, as for configs[1]) instead of re-synthesising chunks.  Patches are generated by
    ``nf_synth_patches`` from ``(seed, global patch index)``: any sharding holds bit-identical data."""

    def __init__(self, model, seed: int, n_total: int, rank: int, world: int, height: int = 32, width: int = 32,
                 fill_chunk: int = 1 << 15):
        import torch
        from .patches import synth_patches
        self.model = model
        self.start, self.stop = shard_range(n_total, rank, world)
        n = self.stop - self.start
        dev = model._dev.device
        self.x = torch.empty((n, height, width, 4), dtype=torch.float32, device=dev)
        self.y = torch.empty_like(self.x)
        k = 0
        while k < n:
            c = min(fill_chunk, n - k)
            synth_patches(seed, self.start + k, c, height, width, device=dev.index, out=(self.x[k:k + c], self.y[k:k + c]))
            k += c

    @property
    def nbytes(self) -> int:
        return 2 * self.x.numel() * 4

    def eval_chunk(self, cond=([0.0], [0.0], [100.0], [2.0])):
        """``eval_chunk`` for :func:`evaluate_sharded` / :func:`timed_sharded_evaluations` over the resident
        block, on the CURRENT stream (with the inputs resident a chunk can be the whole shard — one launch
        of the persistent kernel — so there is no ramp-up / drain between chunks to hide on a second stream)."""
        model, nlf0, nlf1, iso, cam = self.model, *cond
        wide = model.new_sums()

        def run(first, count, sums):
            k = first - self.start
            if k < 0 or k + count > self.x.shape[0]:
                raise IndexError("patches [%d, %d) are outside this rank's resident block" % (first, first + count))
            model.nll_sums(self.x[k:k + count], self.y[k:k + count], nlf0, nlf1, iso, cam, wide)
            return sums

        def finish(sums):
            model.fold_sums(wide, out=sums)
            wide.zero_()
            return sums

        run.finish = finish
        return run
