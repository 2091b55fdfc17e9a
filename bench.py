#!/usr/bin/env python
"""Benchmark of the Noise Flow hot path on MI355X — driver contract.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`, or plain
    `python bench.py --gpus N`, which then launches its own N ranks that way on a free local port)

N = 1 (the headline, BASELINE configs[1] — the configuration the metric is quoted on): a "step" is one
pass of the fused likelihood-direction kernel (per-patch NLL written to HBM + log|det J| + batch sums)
over ONE batch of 1024 synthetic 32x32x4 raw patches with the shipped Noise Flow checkpoint, inputs
resident in HBM.  With --steps < 200 the K-step block is timed R = 11 times back to back and the MEDIAN
block is reported (a 20-step block is ~1.1 ms of device time; one block is a noisy sample).

N > 1 (BASELINE configs[3]): a "step" is one COMPLETE evaluation of the 2^20-patch range, sharded in
contiguous blocks of the patch index over the ranks (strong scaling: the total work is fixed).  Every
rank keeps its block resident in HBM (34 GB / N), evaluates it with the persistent fused kernel and the
evaluation ends with ONE RCCL all-reduce of (sum nll, sum sd_z, count) — 24 bytes, the only collective;
no barrier and no read-back inside the timed region.  NF_BENCH_FORCE_DIST=1 runs this leg on one rank.  The same line
carries BASELINE configs[4] (2^18 patches of 64x64x4, fp16 coupling CNN / fp32 log-det, sharded the same way) as the nested
section `fp16_cnn_64x64_sharded` with its own roofline; `--config c5` makes it the headline instead.  `cpu_baseline` is timed
on rank 0 after the timed regions.

Patches are a pure function of (seed, global patch index), so any sharding evaluates the same data.
Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      the dominant kernel against the dense fp32 matrix/vector peak (the binding roofline:
                algorithmic flops / wall-clock step time), with the HBM view nested
  cpu_baseline  CPU ports of the reference arithmetic (oracle/): fused plain-C/OpenMP (the
                reported value) and the op-per-layer torch-CPU restatement of the TF1 graph,
                timed on this box's host cores on bounded samples (N = 1 only)
  sampling      the sampling direction at BASELINE configs[2] (batch 4096)
  nll_check     GPU mean NLL vs the fp64 CPU oracle on a 64-patch subset
  fp16_cnn_64x64  BASELINE configs[4] shape (64x64x4, fp16 coupling CNN / fp32 log-det)
  wide_cnn      the paper-scale coupling CNN (width 32 / 16) on the f32 matrix cores, own roofline
  sharded_1m    configs[3] on this one GPU (2^20 resident patches, no process group)
  training      one training step (fwd batch-BN + bwd + EMA + Adam) at the reference's minibatch of 138: the shipped
                architecture (width 4) and, nested as `width32`, the paper-scale coupling width on the matrix cores;
                `width512`: the reference's default width (hand-written fp32 matrix-core GEMMs, csrc/nf_train_mm.h)
  two_streams   the headline workload with consecutive steps alternating between two HIP streams
  large_patches 256x256x4 images (beyond the 64x64 a workgroup holds): overlapping tiles, DESIGN 4.8
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ARCH_LABEL = "sdn5|unc|unc|unc|unc|gain4|unc|unc|unc|unc"
ALGO_BYTES_PER_PATCH = 2 * 32 * 32 * 4 * 4          # read x and y once (fp32): 32768 B
ALGO_FLOP_PER_PATCH = 5.1e6
# ... of which the kernels EXECUTE less: §8d counts l_last at 36 x 5 = 180 MAC per pixel and coupling (the edge-indicator channel as a
# fifth input), the kernels fold that channel into a 16-entry border table and run 36 x 4 = 144 — both fractions are reported
EXEC_FLOP_PER_PATCH = ALGO_FLOP_PER_PATCH - 8 * 1024 * 36 * 2
HBM_PEAK_GBS = 8000.0                                # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3                             # fp32 vector peak = fp32-input matrix peak
FP16_SHAPE_PEAK_TFLOPS = 248.0 / (216.0 / 512 + 16.0 / 128 + 16.0 / 32) * 2 * 1024 * 2.4e9 / 1e12   # see _fp16_cnn
FP16_MFMA_PEAK_TFLOPS = 2500.0                       # dense fp16 matrix peak (MI355X_MICROARCH.md)


def _usable_cores(threads: int) -> int:
    """Threads actually backed by CPU time: min(threads, cgroup CPU quota, affinity mask)."""
    n = threads
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(round(q / int(g.read())))))
            break
        except Exception:
            continue
    return int(n)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.11 s of timed work after 13 ms of warm-up — a 20-step warm-up (1.3 ms) ends before the
    # GPU has left its idle clocks and reads ~5 % slow (measured: 62.7 us/step vs 59.5 us steady state)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=1024, help="N = 1: patches per step (configs[1]: 1024)")
    ap.add_argument("--total-patches", type=int, default=0,
                    help="N > 1: size of the patch range one step evaluates, sharded over the ranks (0 = the workload's own: "
                         "configs[3] 2^20 patches of 32x32x4, configs[4] 2^18 patches of 64x64x4 — 34 GB either way)")
    ap.add_argument("--config", choices=("auto", "c4", "c5"), default="auto",
                    help="N > 1 headline: c4 = BASELINE configs[3] (32x32x4, fp32; the default) with configs[4] nested as "
                         "`fp16_cnn_64x64_sharded`; c5 = BASELINE configs[4] (64x64x4, fp16 coupling CNN) as the headline")
    ap.add_argument("--shard-chunk", type=int, default=0,
                    help="N > 1: patches per kernel launch inside a rank's block (0 = the whole block in one launch)")
    ap.add_argument("--sample-batch", type=int, default=4096, help="sampling-direction batch (configs[2]: 4096)")
    ap.add_argument("--pool", type=int, default=16, help="distinct resident batches cycled through")
    ap.add_argument("--ramp-ms", type=float, default=250.0,
                    help="untimed clock ramp before the W warm-up steps: the same step repeated for this long, so that "
                         "a short W does not leave the GPU at its idle clocks (0 disables)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--seed", type=int, default=0)
    return ap.parse_args()


def _kernel_source_sha() -> str:
    """Content hash of the kernel sources a PMC traffic measurement belongs to (the GPU box has no .git)."""
    h = hashlib.sha256()
    for name in ("nf_kernels.hip", "nf_device.h", "nf_dev_util.h"):
        with open(os.path.join(ROOT, "noise_flow_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _traffic(entry=None):
    """HBM bytes per launch from the committed PMC passes (profiles/traffic.json) — only when that file was
    measured on THIS kernel source; otherwise null (a replayed number must not outlive its kernel).
    `entry` names one of the other measured launches (`configs` in the file: "sampling", "fp16_cnn_64x64")."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(tpath) as f:
            t = json.load(f)
        if entry is not None:
            t = t.get("configs", {}).get(entry)
            if t is None:
                return None, "profiles/traffic.json has no entry %r" % entry
        if t.get("kernel_source_sha") != _kernel_source_sha():
            return None, "profiles/traffic.json was measured on a different nf_kernels.hip (sha %s): not reported" % t.get(
                "kernel_source_sha")
        return t.get("hbm_bytes_per_launch"), "profiles/traffic.json (PMC FETCH_SIZE/WRITE_SIZE passes on this kernel source)"
    except Exception as e:
        return None, "no traffic record: %s" % e


def rank_launch_command(n: int, port: int, argv) -> list:
    """The command `python bench.py --gpus N` turns itself into when it was not started under torch.distributed.run
    (the launch line the driver's contract names, with a free port)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def _launch_ranks(n: int) -> int:
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    return subprocess.call(rank_launch_command(n, port, sys.argv[1:]), env=env)


def main():
    global _SECTION_RAMP_MS
    args = parse_args()
    _SECTION_RAMP_MS = min(float(args.ramp_ms), 80.0)
    # stdout carries exactly ONE line, the JSON, written last: native libraries (RCCL prints a version
    # banner through C stdio, flushed at exit AFTER Python's output) would otherwise follow it.  Everything
    # else that targets fd 1 goes to stderr; the JSON is written straight to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher — one process per GPU under torch.distributed.run on a
        # free local port; the ranks inherit this stdout, so rank 0's JSON line is this command's one line
        os.dup2(json_fd, 1)
        os.close(json_fd)
        sys.exit(_launch_ranks(args.gpus))
    args.gpus = world

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    # NF_BENCH_ONE_GPU=1 (test aid): every rank uses GPU 0 — with NF_BENCH_BACKEND=gloo this runs the N > 1 leg end to
    # end on a single-GPU box (RCCL refuses two ranks on one device)
    if os.environ.get("NF_BENCH_ONE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # NF_BENCH_FORCE_DIST=1 runs the N > 1 leg (configs[3] + RCCL) on a single rank too (validation aid)
    use_dist = world > 1 or bool(os.environ.get("NF_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("NF_BENCH_BACKEND", "nccl")                                   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    from noise_flow_amd import NoiseFlow, default_hps
    from noise_flow_amd.ckpt import load_checkpoint

    variables = load_checkpoint(os.path.join(ROOT, "models", "NoiseFlow", "ckpt", "model.ckpt.best"))
    model = NoiseFlow([32, 32, 4], False, default_hps(), variables=variables, device=local_rank)
    ctx = dict(args=args, rank=rank, local_rank=local_rank, world=world, dev=dev, model=model, variables=variables)
    out = sharded_leg(ctx) if use_dist else single_gpu_leg(ctx)
    line = json.dumps(out) if rank == 0 else None
    if use_dist:
        dist.barrier()                      # rank 0's extra sections are done before any rank tears down
        dist.destroy_process_group()
    # flush what native libraries still hold in C stdio (RCCL's banner) so that the JSON is the last thing
    # written even when the caller merges stdout and stderr
    sys.stdout.flush()
    sys.stderr.flush()
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        os.write(json_fd, (line + "\n").encode())
    os.close(json_fd)


_SECTION_RAMP_MS = 80.0     # per extra section (set from --ramp-ms in main: min(ramp_ms, 80))


def _clock_ramp(step, ramp_ms, dev):
    """Untimed: leave the idle clocks (reported as "clock_ramp_ms")."""
    import torch
    if ramp_ms <= 0:
        return
    t_r = time.perf_counter()
    while (time.perf_counter() - t_r) * 1e3 < ramp_ms:
        for i in range(64):
            step(i)
        torch.cuda.synchronize(dev)


# ---------------------------------------------------------------------------------------------------------
# N > 1: BASELINE configs[3] (and configs[4]) — a patch range sharded by patch index, ONE all-reduce per evaluation
# ---------------------------------------------------------------------------------------------------------
SHARDED_WORKLOADS = {
    # BASELINE configs[3]: forward NLL over 2^20 synthetic 32x32x4 patches, exact fp32
    "c4": dict(label="configs[3]", height=32, width=32, cnn_dtype="fp32", total=1 << 20, dtype="f32",
               kernel="nf_flow_kernel<4,256,4,false,true,true,0,false>", flop_per_patch=ALGO_FLOP_PER_PATCH,
               peak=VALU_PEAK_TFLOPS, peak_name="dense fp32 matrix/vector peak"),
    # BASELINE configs[4]: 64x64x4 patches, fp16 coupling CNN / fp32 log-det; 2^18 patches = the same 34 GB of input
    "c5": dict(label="configs[4]", height=64, width=64, cnn_dtype="fp16", total=1 << 18, dtype="f16 CNN / f32 log-det",
               kernel="nf_flow_kernel<4,1024,4,false,true,true,1,false>", flop_per_patch=4 * ALGO_FLOP_PER_PATCH,
               peak=FP16_MFMA_PEAK_TFLOPS, peak_name="dense fp16 matrix peak"),
}


def sharded_workload(name: str, total_patches: int = 0) -> dict:
    """The sharded workload `name` ("c4" | "c5") with its patch range (0 = the workload's own size)."""
    w = dict(SHARDED_WORKLOADS[name], name=name)
    if total_patches > 0:
        w["total"] = int(total_patches)
    w["bytes_per_patch"] = 2 * w["height"] * w["width"] * 4 * 4
    return w


def _run_sharded(ctx, spec, K, Wm):
    """K timed complete evaluations of spec's patch range on this rank's resident block (+ the one all-reduce each);
    returns this rank's measurements and, gathered, every rank's."""
    import torch
    import torch.distributed as dist
    from noise_flow_amd import NoiseFlow, default_hps
    from noise_flow_amd.dist import ResidentShard, timed_sharded_evaluations
    args, rank, world, dev = ctx["args"], ctx["rank"], ctx["world"], ctx["dev"]
    # Everything that can fail on ONE rank alone (model creation, the rank's share of the resident block: 34 GB / N) happens
    # before the first collective of this workload, and the ranks agree on the outcome: a rank that raised while the others
    # entered the all-reduce of the first evaluation would leave them waiting forever — and with them the headline line.
    model = shard = None
    err = None
    try:
        if spec["name"] == "c4":
            model = ctx["model"]
        else:
            model = NoiseFlow([spec["height"], spec["width"], 4], False, default_hps(), variables=ctx["variables"],
                              device=ctx["local_rank"], cnn_dtype=spec["cnn_dtype"])
        shard = ResidentShard(model, args.seed, spec["total"], rank, world, spec["height"], spec["width"],
                              fill_chunk=max(1, (1 << 25) // (spec["height"] * spec["width"])))
    except Exception as e:
        err = e
    ok = torch.tensor([0.0 if err is not None else 1.0], dtype=torch.float64, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok.item()) < 1.0:
        del shard
        torch.cuda.empty_cache()
        raise RuntimeError("set-up of the sharded workload %r failed on %s" % (
            spec["name"], "this rank: %s: %s" % (type(err).__name__, err) if err is not None else "another rank"))
    n_total = spec["total"]
    n_local = shard.stop - shard.start
    chunk = args.shard_chunk if args.shard_chunk > 0 else max(1, n_local)
    eval_chunk = shard.eval_chunk()
    stream = torch.cuda.current_stream(dev)
    new_sums = lambda: torch.zeros(3, dtype=torch.float64, device=dev)   # noqa: E731
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]

    def on_step(i, what):            # the rank's own device work of evaluation i, before the collective
        ev[i][0 if what == "begin" else 1].record(stream)

    if args.ramp_ms > 0:             # untimed clock ramp on a slice of the resident block
        nb = min(max(1, (1 << 20) // (spec["height"] * spec["width"])), n_local)
        scratch = model.new_sums()
        _clock_ramp(lambda i: model.nll_sums(shard.x[:nb], shard.y[:nb], [0.0], [0.0], [100.0], [2.0], scratch),
                    args.ramp_ms, dev)
    res = timed_sharded_evaluations(eval_chunk, n_total, chunk, rank, world, K, Wm, new_sums,
                                    sync=lambda: torch.cuda.synchronize(dev), on_step=on_step)
    kernel_ms_local = sum(a.elapsed_time(b) for a, b in ev) / K

    # the all-reduce on its own (untimed diagnostic): 24-byte message, pure latency
    probe = torch.zeros(3, dtype=torch.float64, device=dev)
    for _ in range(5):
        dist.all_reduce(probe)
    torch.cuda.synchronize(dev)
    t_a = time.perf_counter()
    n_probe = 50
    for _ in range(n_probe):
        dist.all_reduce(probe)
    torch.cuda.synchronize(dev)
    allreduce_us = (time.perf_counter() - t_a) / n_probe * 1e6

    stats = torch.tensor([res["elapsed"], kernel_ms_local, allreduce_us], dtype=torch.float64, device=dev)
    gathered = [torch.zeros_like(stats) for _ in range(world)]
    dist.all_gather(gathered, stats)
    per_rank = [g.cpu().tolist() for g in gathered]
    nbytes = shard.nbytes
    del shard, eval_chunk
    torch.cuda.empty_cache()
    return dict(res=res, per_rank=per_rank, chunk=chunk, nbytes=nbytes)


def _sharded_record(spec, run, world, K, Wm, backend):
    """The JSON fields of one sharded workload (rank 0)."""
    res, per_rank, n_total = run["res"], run["per_rank"], spec["total"]
    elapsed = max(p[0] for p in per_rank)
    means = [r[0] for r in res["results"]]
    ms_step = 1e3 * elapsed / K
    kernel_ms = max(p[1] for p in per_rank)
    gbs = spec["bytes_per_patch"] * (n_total / world) / (kernel_ms * 1e-3) / 1e9       # per GPU
    tfl = spec["flop_per_patch"] * (n_total / world) / (kernel_ms * 1e-3) / 1e12
    patch = "%dx%dx4" % (spec["height"], spec["width"])
    return {
        "value": n_total * K / elapsed, "unit": "patches/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": ms_step, "scaling": "strong", "dtype": spec["dtype"],
        "config": {"workload": "%s: forward NLL over %d synthetic %s patches (full arch %s, shipped checkpoint, ISO 100 / cam S6%s), "
                               "patch range sharded in contiguous blocks over %d GPU(s); one step = one complete evaluation "
                               "ending in ONE all-reduce of (sum nll, sum sd_z, count)"
                               % (spec["label"], n_total, patch, ARCH_LABEL,
                                  ", fp16 coupling CNN with fp32 log-det accumulation" if spec["cnn_dtype"] == "fp16" else "", world),
                   "total_patches": n_total, "global_batch": n_total, "patches_per_gpu": n_total // world, "patch": patch,
                   "launch_chunk": run["chunk"], "inputs": "resident in HBM (%.1f GB per GPU)" % (run["nbytes"] / 1e9),
                   "backend": backend,
                   "parallelism": "dp%d: patch-index sharding, %s" % (
                       world, "one RCCL all-reduce of 3 fp64 scalars per evaluation" if world > 1 else
                       "single rank (NF_BENCH_FORCE_DIST: the collective is issued on a 1-rank group)")},
        "mean_nll": means[-1], "sd_z": res["results"][-1][1],
        "mean_nll_identical_across_steps": bool(max(means) - min(means) <= 1e-9 * abs(means[0])),
        "per_rank": {"ranks": world, "elapsed_s": [p[0] for p in per_rank], "kernel_ms_per_step": [p[1] for p in per_rank],
                     "allreduce_us": [p[2] for p in per_rank]},
        "collective": {"allreduce_us": max(p[2] for p in per_rank), "per_step": 1,
                       "share_of_step": max(p[2] for p in per_rank) * 1e-3 / ms_step,
                       "non_kernel_share_of_step": max(0.0, 1.0 - kernel_ms / ms_step),
                       "note": "allreduce_us = one all-reduce + device sync, timed alone after the run; "
                               "non_kernel_share = 1 - (slowest rank's kernel time / step time)"},
        "roofline": {"bound": "mfma", "achieved": tfl, "peak": spec["peak"], "unit": "TFLOP/s",
                     "frac": tfl / spec["peak"], "traffic": None, "scope": "per GPU, slowest rank's kernel time",
                     "peak_is": spec["peak_name"], "kernel": spec["kernel"], "kernel_ms": kernel_ms,
                     "algorithmic_flop_per_launch": spec["flop_per_patch"] * n_total / world,
                     "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": spec["bytes_per_patch"] * n_total / world}},
    }


def sharded_leg(ctx):
    import torch.distributed as dist
    args, rank, world = ctx["args"], ctx["rank"], ctx["world"]
    K, Wm = args.steps, args.warmup
    head = "c5" if args.config == "c5" else "c4"
    spec = sharded_workload(head, args.total_patches)
    run = _run_sharded(ctx, spec, K, Wm)
    nested = None
    if head == "c4":                                   # BASELINE configs[4] rides along as a nested section
        spec5 = sharded_workload("c5", max(0, args.total_patches // 4))
        k5 = max(1, min(K, 10))
        try:
            nested = (spec5, _run_sharded(ctx, spec5, k5, min(Wm, 2)), k5)
        except Exception as e:                         # the headline must not depend on the nested section
            nested = e
    if rank != 0:
        return None
    out = {"metric": "patches/sec (32x32x4) fwd-NLL and inverse-sample; mean NLL vs CPU ref"}
    out.update(_sharded_record(spec, run, world, K, Wm, dist.get_backend()))
    out.update({"higher_is_better": True, "vs_baseline": None, "data": "synthetic", "clock_ramp_ms": args.ramp_ms})
    if head == "c4":
        traffic, traffic_src = _traffic()
        if traffic is not None:                        # PMC bytes of a 1024-patch launch of the same kernel, per patch x patches per launch
            out["roofline"]["traffic"] = traffic / 1024.0 * (spec["total"] / world)
            out["roofline"]["traffic_source"] = traffic_src + ", scaled from the measured 1024-patch launch to this launch's patch count"
    def fp16_traffic(rec, sp):     # the 64x64 fp16 kernel's PMC bytes, measured on a 1024-patch launch, per patch x patches of this launch
        tr, src = _traffic("fp16_cnn_64x64")
        if tr is not None:
            rec["roofline"]["traffic"] = tr / 1024.0 * (sp["total"] / world)
            rec["roofline"]["traffic_source"] = src + ", scaled from the measured 1024-patch launch to this launch's patch count"
    if head != "c4":
        fp16_traffic(out, spec)
    if isinstance(nested, tuple):
        sec = _sharded_record(nested[0], nested[1], world, nested[2], min(Wm, 2), dist.get_backend())
        sec["roofline"]["hbm"]["achievable_frac"] = sec["roofline"]["hbm"]["achieved"] / 6290.0     # guide: 6.29 TB/s achievable
        fp16_traffic(sec, nested[0])
        out["fp16_cnn_64x64_sharded"] = sec
    elif nested is not None:
        out["fp16_cnn_64x64_sharded"] = {"error": "%s: %s" % (type(nested).__name__, nested)}
    # CPU baseline: rank 0 only, AFTER the timed regions (the other ranks wait in main()'s closing barrier)
    out["cpu_baseline"] = None
    if not args.no_cpu_baseline:
        from noise_flow_amd.patches import synth_patches
        xs, ys = synth_patches(args.seed, 0, 1024, device=ctx["local_rank"])       # patches [0, 1024) of configs[3]'s range
        out["cpu_baseline"] = _cpu_baseline(args, ctx["variables"], xs.cpu().numpy(), ys.cpu().numpy(), 1024)
        out["cpu_baseline"]["workload_note"] = "32x32x4 fp32 forward NLL (the per-patch work of configs[1] / configs[3])"
    return out


# ---------------------------------------------------------------------------------------------------------
# N = 1: BASELINE configs[1] headline + the other single-GPU sections
# ---------------------------------------------------------------------------------------------------------
def single_gpu_leg(ctx):
    import numpy as np
    import torch
    from noise_flow_amd import NoiseFlow, default_hps, _lib
    from noise_flow_amd.patches import synth_patches
    args, local_rank, dev, model, variables = ctx["args"], ctx["local_rank"], ctx["dev"], ctx["model"], ctx["variables"]
    lib = _lib.load()
    B, K, Wm = args.batch, args.steps, args.warmup
    R = 1 if K >= 200 else 11               # short blocks are repeated and the median block reported
    cond = _lib.nf_cond(100.0, 2.0, 0.000479, 0.000002)      # ISO 100, S6 (train_noise_flow.py:143-147)

    # ---- resident synthetic data: batch j holds patches [j*B, +B) ----
    pool = max(1, min(args.pool, K * R + Wm))
    batches = [synth_patches(args.seed, j * B, B, device=local_rank) for j in range(pool)]
    nll_buf = torch.empty(B, dtype=torch.float32, device=dev)      # the metric includes the per-patch NLL
    sums = torch.zeros(3, dtype=torch.float64, device=dev)
    # per-workgroup partial sums go to the C ABI's slotted accumulator (NF_SUMS_WIDE: 64 slots on separate
    # cache lines); ONE nf_sums_reduce after the last step folds it into (sum nll, sum sd, count)
    wide = torch.zeros(_lib.NF_SUMS_SLOTS * _lib.NF_SUMS_STRIDE, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)
    sptr = int(stream.cuda_stream)
    hptr = model._flow.ptr

    def nll_step(i, flags=_lib.NF_ACCUMULATE | _lib.NF_SUMS_WIDE):
        x, y = batches[i % pool]
        rc = lib.nf_nll(hptr, x.data_ptr(), y.data_ptr(), B, C.byref(cond), nll_buf.data_ptr(), None, None, None,
                        wide.data_ptr(), flags, sptr)
        if rc != 0:
            _lib.check(rc)

    _clock_ramp(nll_step, args.ramp_ms, dev)
    for i in range(Wm):
        nll_step(i)
    sums.zero_()
    wide.zero_()
    torch.cuda.synchronize(dev)

    blocks, kblocks = [], []
    step_no = 0
    for r in range(R):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(K):
            nll_step(step_no)
            step_no += 1
        ev1.record(stream)
        torch.cuda.synchronize(dev)
        blocks.append(time.perf_counter() - t0)
        kblocks.append(ev0.elapsed_time(ev1) / K)
    _lib.check(lib.nf_sums_reduce(wide.data_ptr(), sums.data_ptr(), 0, sptr))
    torch.cuda.synchronize(dev)
    elapsed = float(np.median(blocks))
    kernel_ms = float(np.median(kblocks))   # average launch duration over the (median) timed block (HIP events)
    s = sums.cpu().numpy()
    assert int(round(s[2])) == B * K * R, (s, B * K * R)
    value = B * K / elapsed
    ms_step = 1e3 * elapsed / K

    extras = {}
    if not args.no_extras:
        for name, fn in (("two_streams", _two_streams), ("sampling", _sampling), ("fp16_cnn_64x64", _fp16_cnn),
                         ("wide_cnn", _wide_cnn), ("sharded_1m", _sharded_1m), ("training", _training),
                         ("large_patches", _large_patches)):
            try:
                extras[name] = fn(ctx, batches, cond, wide)
            except Exception as e:           # the headline metric must not depend on an optional section
                extras[name] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- parity + CPU baseline (oracle/ is the checker, never the product) ----
    from oracle.nf_oracle import NoiseFlowOracle
    x0, y0 = batches[0]
    nb = 64
    nll_gpu, _ = model._loss(x0[:nb], y0[:nb], [0.0], [0.0], [100.0], [2.0])
    ref = NoiseFlowOracle(ARCH_LABEL, variables).nll(x0[:nb].cpu().numpy(), y0[:nb].cpu().numpy(), 100.0, 2.0)[0]
    g = nll_gpu.double().cpu().numpy()
    nll_check = {"patches": nb, "gpu_mean_nll": float(g.mean()), "cpu_fp64_mean_nll": float(ref.mean()),
                 "rel_err_mean": float(abs(g.mean() - ref.mean()) / abs(ref.mean())),
                 "max_rel_err_per_patch": float(np.max(np.abs(g - ref) / np.abs(ref))), "tolerance": 1e-5}
    cpu_baseline = None
    if not args.no_cpu_baseline:
        cpu_baseline = _cpu_baseline(args, variables, x0.cpu().numpy(), y0.cpu().numpy(), B)

    traffic, traffic_src = _traffic()
    # roofline from the WALL-CLOCK step time (what the driver's clock can confirm); the HIP-event figure is nested
    gbs = ALGO_BYTES_PER_PATCH * B / (ms_step * 1e-3) / 1e9
    tfl = ALGO_FLOP_PER_PATCH * B / (ms_step * 1e-3) / 1e12
    out = {
        "metric": "patches/sec (32x32x4) fwd-NLL and inverse-sample; mean NLL vs CPU ref",
        "value": value, "unit": "patches/s", "n_gpus": 1, "steps": K, "warmup": Wm,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: full NoiseFlow arch (%s) forward NLL (per-patch NLL + batch sums), batch %d "
                               "synthetic 32x32x4 patches, shipped checkpoint, ISO 100 / cam S6" % (ARCH_LABEL, B),
                   "batch_per_gpu": B, "global_batch": B, "patch": "32x32x4",
                   "parallelism": "single GPU, no collective (N > 1 runs configs[3]: patch-index sharding + one all-reduce)"},
        "timing": {"repetitions": R, "statistic": "median of R back-to-back K-step blocks" if R > 1 else "one K-step block",
                   "block_ms": [1e3 * b for b in blocks], "block_ms_min": 1e3 * min(blocks), "block_ms_max": 1e3 * max(blocks)},
        "mean_nll": float(s[0] / s[2]), "sd_z": float(s[1] / s[2]),
        # The fused kernel is compute-bound (~156 flop/B): its FMAs are v_mfma_f32_4x4x1 on the fp32
        # matrix/vector datapath (dense fp32 peak 157.3 TFLOP/s) -> that is the binding roofline.
        # The HBM view the task also asks for is nested under "hbm"; "traffic" = HBM bytes per
        # launch from the FETCH_SIZE / WRITE_SIZE PMC passes (profiles/traffic.json).
        "roofline": {"bound": "mfma", "achieved": tfl, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": tfl / VALU_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "time_base": "wall-clock ms_per_step",
                     "dtype": "f32 (v_mfma_f32_4x4x1_16b_f32, exact fp32; shares the fp32 datapath with VALU)",
                     "kernel": "nf_flow_kernel<4,256,4,false,true,true,0,false>", "kernel_ms": kernel_ms,
                     "frac_from_kernel_ms": ALGO_FLOP_PER_PATCH * B / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                     "algorithmic_flop_per_launch": ALGO_FLOP_PER_PATCH * B,
                     "executed_flop_per_launch": EXEC_FLOP_PER_PATCH * B,
                     "frac_executed": EXEC_FLOP_PER_PATCH * B / (ms_step * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                     "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PATCH * B,
                             "note": "structurally capped near 13 %: 32 KiB and 5.1 MFLOP per patch"}},
        "cpu_baseline": cpu_baseline, "nll_check": nll_check, "clock_ramp_ms": args.ramp_ms,
    }
    out.update(extras)
    return out


def _two_streams(ctx, batches, cond, wide):
    """The same K steps alternating between TWO streams: a 1024-patch launch fills the GPU exactly once, so
    back-to-back launches on one stream each pay their own ramp-up and drain; on two streams the next launch takes
    the workgroup slots the previous one frees (what noise_flow_amd.dist.flow_eval_chunk does)."""
    import torch
    from noise_flow_amd import _lib
    args, dev, model = ctx["args"], ctx["dev"], ctx["model"]
    lib, B, K, pool = _lib.load(), args.batch, args.steps, len(batches)
    side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    wide2 = torch.zeros_like(wide)
    nll2 = [torch.empty(B, dtype=torch.float32, device=dev) for _ in side]
    torch.cuda.synchronize(dev)

    def step(i):
        x, y = batches[i % pool]
        rc = lib.nf_nll(model._flow.ptr, x.data_ptr(), y.data_ptr(), B, C.byref(cond), nll2[i & 1].data_ptr(), None, None, None,
                        wide2.data_ptr(), _lib.NF_ACCUMULATE | _lib.NF_SUMS_WIDE, int(side[i & 1].cuda_stream))
        if rc != 0:
            _lib.check(rc)
    for i in range(max(args.warmup, 500)):
        step(i)
    torch.cuda.synchronize(dev)
    wide2.zero_()
    torch.cuda.synchronize(dev)
    K2 = max(K, 1000)
    t2 = time.perf_counter()
    for i in range(K2):
        step(i)
    torch.cuda.synchronize(dev)
    el2 = time.perf_counter() - t2
    s2 = torch.zeros(3, dtype=torch.float64, device=dev)
    _lib.check(lib.nf_sums_reduce(wide2.data_ptr(), s2.data_ptr(), 0, int(torch.cuda.current_stream(dev).cuda_stream)))
    s2 = s2.cpu().numpy()
    assert int(round(s2[2])) == B * K2
    return {"value": B * K2 / el2, "unit": "patches/s", "ms_per_step": el2 / K2 * 1e3, "steps": K2, "streams": 2,
            "mean_nll": float(s2[0] / s2[2]),
            "note": "same workload and kernel, consecutive steps alternate between two HIP streams so that launches overlap "
                    "their ramp-up / drain; the headline value stays single-stream so that roofline.kernel_ms agrees with "
                    "rocprof's per-kernel duration"}


def _sampling(ctx, batches, cond, wide):
    """Sampling direction (configs[2]): B = 4096, fixed cam / ISO, in-kernel Philox eps."""
    import torch
    from noise_flow_amd import _lib
    from noise_flow_amd.patches import synth_patches
    args, dev, model = ctx["args"], ctx["dev"], ctx["model"]
    lib, SB = _lib.load(), args.sample_batch
    stream = torch.cuda.current_stream(dev)
    sptr = int(stream.cuda_stream)
    _, ys = synth_patches(args.seed, 1 << 40, SB, device=dev.index, want_x=False)
    xs = torch.empty_like(ys)
    ks = max(10, min(args.steps, 100))

    def sample_step(i):
        rc = lib.nf_sample(model._flow.ptr, ys.data_ptr(), None, args.seed, i * SB, 1.0, SB, C.byref(cond), xs.data_ptr(), sptr)
        if rc != 0:
            _lib.check(rc)

    for i in range(5):
        sample_step(i)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = time.perf_counter()
    e0.record(stream)
    for i in range(ks):
        sample_step(i)
    e1.record(stream)
    torch.cuda.synchronize(dev)
    ts = time.perf_counter() - ts
    kms = e0.elapsed_time(e1) / ks
    tfl = ALGO_FLOP_PER_PATCH * SB / (kms * 1e-3) / 1e12
    sbytes = 2 * 32 * 32 * 4 * 4 * SB              # reads y, writes x (eps is drawn in-kernel)
    traffic, traffic_src = _traffic("sampling")
    return {"value": SB * ks / ts, "unit": "patches/s", "batch": SB, "steps": ks,
            "ms_per_step": 1e3 * ts / ks, "kernel_ms": kms, "temp": 1.0,
            "eps": "in-kernel Philox4x32-10", "workload": "configs[2]: inverse sampling, clean patch + fixed cam/ISO",
            # same arithmetic as the NLL direction (the Philox / Box-Muller draw is not counted): fp32 matrix = vector peak
            "roofline": {"bound": "mfma", "achieved": tfl, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / VALU_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src, "time_base": "HIP events over the timed launches",
                         "kernel": "nf_flow_kernel<4,256,4,true,true,true,0,false>", "kernel_ms": kms,
                         "algorithmic_flop_per_launch": ALGO_FLOP_PER_PATCH * SB,
                         "executed_flop_per_launch": EXEC_FLOP_PER_PATCH * SB,
                         "frac_executed": EXEC_FLOP_PER_PATCH * SB / (kms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                         "hbm": {"achieved": sbytes / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": sbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": sbytes}}}


def _time_nll(model, x, y, cond, n, dev):
    """Average HIP-event duration (ms) of n NLL launches with slotted sums on the current stream."""
    import torch
    from noise_flow_amd import _lib
    lib = _lib.load()
    stream = torch.cuda.current_stream(dev)
    sptr = int(stream.cuda_stream)
    acc = torch.zeros(_lib.NF_SUMS_SLOTS * _lib.NF_SUMS_STRIDE, dtype=torch.float64, device=dev)
    nll = torch.empty(x.shape[0], dtype=torch.float32, device=dev)

    def step():
        rc = lib.nf_nll(model._flow.ptr, x.data_ptr(), y.data_ptr(), int(x.shape[0]), C.byref(cond), nll.data_ptr(), None, None,
                        None, acc.data_ptr(), _lib.NF_ACCUMULATE | _lib.NF_SUMS_WIDE, sptr)
        if rc != 0:
            _lib.check(rc)
    for _ in range(5):
        step()
    torch.cuda.synchronize(dev)
    # untimed clock ramp of this section (each section builds its model on the host first: the GPU idles and its clocks drop; a
    # section timed from there read up to 10 % slower than the kernel-stats csv of the same launches, profiles/README.md)
    t_r = time.perf_counter()
    while (time.perf_counter() - t_r) * 1e3 < _SECTION_RAMP_MS:
        for _ in range(4):
            step()
        torch.cuda.synchronize(dev)
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f0.record(stream)
    for _ in range(n):
        step()
    f1.record(stream)
    torch.cuda.synchronize(dev)
    return f0.elapsed_time(f1) / n, nll


def _fp16_cnn(ctx, batches, cond, wide):
    """BASELINE configs[4] shape: 64x64x4 patches, fp16 coupling CNN / fp32 log-det."""
    from noise_flow_amd import NoiseFlow, default_hps
    from noise_flow_amd.patches import synth_patches
    args, dev = ctx["args"], ctx["dev"]
    m16 = NoiseFlow([64, 64, 4], False, default_hps(), variables=ctx["variables"], device=dev.index, cnn_dtype="fp16")
    B16 = 1024
    x16, y16 = synth_patches(args.seed, 1 << 41, B16, 64, 64, device=dev.index)
    k16 = max(10, min(args.steps, 50))
    ms16, _ = _time_nll(m16, x16, y16, cond, k16, dev)
    bytes16 = 2 * 64 * 64 * 4 * 4 * B16
    flop16 = 4 * ALGO_FLOP_PER_PATCH * B16
    tfl16 = flop16 / (ms16 * 1e-3) / 1e12
    traffic, traffic_src = _traffic("fp16_cnn_64x64")
    old = os.environ.get("NF_H16") == "4x4"
    return {"workload": "configs[4] shape: forward NLL, 64x64x4 patches, fp16 coupling CNN "
                        "(%s, fp32 accumulate + fp32 log-det), fp32 I/O, 1 GPU"
                        % ("v_mfma_f32_4x4x4_16b_f16" if old else "v_mfma_f32_16x16x32_f16 for the 3x3 convs"),
            "batch": B16, "steps": k16, "kernel_ms": ms16, "value": B16 / (ms16 * 1e-3), "unit": "patches/s",
            "pixels_per_s": B16 * 4096 / (ms16 * 1e-3), "algorithmic_tflops": tfl16,
            "roofline": {"bound": "mfma", "achieved": tfl16, "peak": FP16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl16 / FP16_MFMA_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src, "time_base": "HIP events over the timed launches",
                         "kernel": "nf_flow_kernel<4,1024,4,false,true,true,%d,false>" % (1 if old else 2), "kernel_ms": ms16,
                         "algorithmic_flop_per_launch": flop16,
                         # the peak of the instructions this kernel actually issues: per pixel and coupling 216 MAC (l_1, l_last) on
                         # v_mfma_f32_16x16x32_f16 (512 MAC/clk/SIMD = the dense fp16 rate), 16 (l_2) on v_mfma_f32_4x4x4_16b_f16 (128),
                         # 16 (the fp32 1x1 mix) on v_mfma_f32_4x4x1_16b_f32 (32): 248 MAC in 1.047 clk
                         "shape_peak": {"peak": FP16_SHAPE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl16 / FP16_SHAPE_PEAK_TFLOPS,
                                        "note": "MAC-weighted rate of the three matrix instructions of the kernel at 1024 SIMDs x 2.4 GHz; "
                                                "the fp32 mix alone is half of that time"},
                         "hbm": {"achieved": bytes16 / (ms16 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": bytes16 / (ms16 * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes16}},
            "hbm": {"achieved": bytes16 / (ms16 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": bytes16 / (ms16 * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes16}}


def _large_patches(ctx, batches, cond, wide):
    """Patches beyond 64x64: 256x256x4 images as overlapping 64-pixel tiles, shipped model, forward NLL."""
    import ctypes
    import numpy as np
    from noise_flow_amd import NoiseFlow, default_hps, _lib, params
    from noise_flow_amd.patches import synth_patches
    args, dev = ctx["args"], ctx["dev"]
    side, Bl = 256, 64
    hps = default_hps()
    m = NoiseFlow([side, side, 4], False, hps, variables=ctx["variables"], device=dev.index)
    xl, yl = synth_patches(args.seed, 1 << 42, Bl, side, side, device=dev.index)
    kl = max(10, min(args.steps, 50))
    ms, nll = _time_nll(m, xl, yl, cond, kl, dev)
    _, descs, flat = params.pack(hps.arch, ctx["variables"], hps.width)
    seg = (ctypes.c_int32 * 80)()
    nseg = int(m._flow.lib.nf_tile_segments(ctypes.byref(_lib.nf_config(side, side, 4, len(descs), -1, 0)), descs,
                                            flat.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), flat.size, 0, seg, 16))
    tiles = [int(seg[5 * i + 3] * seg[5 * i + 4]) for i in range(max(nseg, 0))]
    px = Bl * side * side
    return {"workload": "forward NLL, %dx%dx4 images, shipped model, fp32: overlapping 64x64 tiles, %d segment(s)" % (side, side, nseg),
            "batch": Bl, "steps": kl, "kernel_ms": ms, "value": Bl / (ms * 1e-3), "unit": "patches/s",
            "pixels_per_s": px / (ms * 1e-3), "segments": nseg, "tiles_per_image_per_segment": tiles,
            "halo": [int(seg[5 * i + 2]) for i in range(max(nseg, 0))],
            "finite": bool(np.isfinite(nll.cpu().numpy()).all()),
            "whole_patch_rate_note": "a 64x64 patch held whole runs at ~1.9e10 pixels/s; the rest is halo recomputation",
            "hbm": {"achieved": 2 * px * 16 / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "algorithmic_bytes_per_launch": 2 * px * 16}}


def wide_flop_per_pixel(width: int, n_couplings: int = 8) -> float:
    """Algorithmic flops per pixel of the shipped layer sequence with coupling-CNN width w
    (layers.py:463-497): per coupling 1x1 mix 16 + l_1 18w + l_2 w^2 + l_last 9(w+1)4 MAC, + ~56 flop of
    bias/BN/ReLU/tanh/exp/affine; + ~40 flop for sdn/gain/prior per pixel (generalised in w)."""
    mac = 16 + 18 * width + width * width + 36 * (width + 1)
    return n_couplings * (2 * mac + 56) + 40


def _wide_cnn(ctx, batches, cond, wide):
    """Paper-scale coupling CNN (job_noise_flow.sh:19: width 32; also 16) and Glow's default width 512 (sidd/ArgParser.py:43; also
    128) on the GEMM kernel: same layer sequence, fresh wide CNN weights
    (no wide checkpoint ships), forward NLL at B = 1024, 32x32x4 — f32 matrix cores (v_mfma_f32_32x32x2_f32 at width 32,
    v_mfma_f32_16x16x4_f32 at width 16) and the fp16 CNN mode (v_mfma_f32_32x32x16_f16)."""
    import numpy as np
    from noise_flow_amd import NoiseFlow, default_hps, params as _params
    args, dev = ctx["args"], ctx["dev"]
    x, y = batches[0]
    out = {}
    for w, dt in ((32, "fp32"), (16, "fp32"), (32, "fp16"), (512, "fp32"), (128, "fp32"), (512, "fp16")):
        hps = default_hps(width=w)
        var = _params.init_variables(hps.arch, w, 4, 1234)
        rng = np.random.RandomState(w)
        for k in list(var):                 # fresh init has a zero last layer: perturb so that every term is live
            if k.endswith("l_last/W") or k.endswith("l_last/b"):
                var[k] = (0.02 * rng.randn(*var[k].shape) * min(1.0, (32.0 / w) ** 0.5)).astype(np.float32)
        m = NoiseFlow([32, 32, 4], False, hps, variables=var, device=dev.index, cnn_dtype=dt)
        nb = x.shape[0] if w <= 32 else 512          # width 512 is 4.7 GFLOP per patch: 512 patches = 2 rounds of the 256 CUs
        kw = max(5, min(args.steps, 20 if w <= 32 else 5))
        ms, nll = _time_nll(m, x[:nb], y[:nb], cond, kw, dev)
        flop = wide_flop_per_pixel(w) * 1024 * nb
        tfl = flop / (ms * 1e-3) / 1e12
        peak = VALU_PEAK_TFLOPS if dt == "fp32" else FP16_MFMA_PEAK_TFLOPS
        path = m._flow.lib.nf_kernel_path(m._flow.ptr, 0)
        out["w%d%s" % (w, "" if dt == "fp32" else "_fp16")] = {
            "width": w, "cnn_dtype": dt, "batch": int(nb), "steps": kw, "kernel_ms": ms,
            "value": nb / (ms * 1e-3), "unit": "patches/s", "finite": bool(np.isfinite(nll.cpu().numpy()).all()),
            "kernel_path": {0: "scalar-weight VALU kernel", 3: "nf_wide32_kernel (v_mfma_f32_32x32x2_f32)",
                            4: "nf_wide16_kernel (v_mfma_f32_16x16x4_f32)",
                            5: "nf_wide32_kernel (v_mfma_f32_32x32x16_f16)",
                            6: ("nf_gemmb_kernel (v_mfma_f32_32x32x2_f32, pixel tile per wavefront, weights resident in LDS)"
                                if w <= 128 and os.environ.get("NF_GEMM", "") != "a" else
                                "nf_gemm_kernel (v_mfma_f32_32x32x2_f32, LDS-staged GEMM, weights streamed from L2)"),
                            7: ("nf_gemm16b_kernel (v_mfma_f32_32x32x16_f16, pixel tile per wavefront, weights resident in LDS)"
                                if w <= 128 and os.environ.get("NF_GEMM16", "") != "a" else
                                "nf_gemm16_kernel (v_mfma_f32_32x32x16_f16, LDS-staged GEMM, weights streamed from L2)"),
                            }.get(path, str(path)),
            "roofline": {"bound": "mfma", "achieved": tfl, "peak": peak, "unit": "TFLOP/s", "frac": tfl / peak,
                         "algorithmic_flop_per_launch": flop, "mac_per_pixel_per_coupling": 16 + 18 * w + w * w + 36 * (w + 1),
                         "dtype": "f32 in / f32 accumulate (exact fp32)" if dt == "fp32" else
                                  "f16 in / f32 accumulate for the three CNN convs; everything else fp32"}}
        del m
    out["workload"] = ("forward NLL, synthetic 32x32x4 patches (1024 at widths <= 32, 512 beyond), arch %s with coupling-CNN width 32 / 16 "
                       "and, on the LDS-staged GEMM kernel, 512 (sidd/ArgParser.py:43 default) / 128 (fresh wide CNN weights: no wide "
                       "checkpoint ships); w32_fp16 = NF_CFG_FP16_CNN" % ARCH_LABEL)
    return out


def _sharded_1m(ctx, batches, cond, wide):
    """configs[3] on ONE GPU without a process group: 2^20 resident patches, K' complete evaluations."""
    import torch
    from noise_flow_amd.dist import ResidentShard, timed_sharded_evaluations
    args, dev, model = ctx["args"], ctx["dev"], ctx["model"]
    n_total = args.total_patches or (1 << 20)
    shard = ResidentShard(model, args.seed, n_total, 0, 1)
    ks = 3
    res = timed_sharded_evaluations(shard.eval_chunk(), n_total, n_total, 0, 1, ks, 1,
                                    lambda: torch.zeros(3, dtype=torch.float64, device=dev),
                                    sync=lambda: torch.cuda.synchronize(dev))
    el = res["elapsed"]
    out = {"workload": "configs[3] on one GPU: forward NLL over %d resident patches per evaluation, one launch" % n_total,
           "steps": ks, "ms_per_step": 1e3 * el / ks, "value": n_total * ks / el, "unit": "patches/s",
           "mean_nll": res["results"][-1][0], "resident_gb": shard.nbytes / 1e9}
    del shard
    torch.cuda.empty_cache()
    return out


def _training(ctx, batches, cond, wide):
    """Training step at the reference's minibatch of 138 patches."""
    import torch
    from noise_flow_amd import default_hps
    from noise_flow_amd.patches import synth_patches
    from noise_flow_amd.train import Trainer
    args, dev = ctx["args"], ctx["dev"]
    stream = torch.cuda.current_stream(dev)
    TB_ = 138                                            # job_noise_flow.sh: --n_batch_train 138
    trn = Trainer([32, 32, 4], default_hps(), variables=ctx["variables"], device=dev.index, max_batch=TB_)
    xt_, yt_ = synth_patches(args.seed, 1 << 42, TB_, device=dev.index)
    kt = max(10, min(args.steps, 50))
    for _ in range(5):
        trn.step(xt_, yt_, [0.0], [0.0], [100.0], [2.0], lr=1e-4, sync=False)
    torch.cuda.synchronize(dev)
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0.record(stream)
    for _ in range(kt):
        trn.step(xt_, yt_, [0.0], [0.0], [100.0], [2.0], lr=1e-4, sync=False)
    g1.record(stream)
    torch.cuda.synchronize(dev)
    mst = g0.elapsed_time(g1) / kt
    trn.close()
    out = {"workload": "one training step = forward with batch-statistics BN + backward + BN EMA + Adam "
                       "(train_noise_flow.py:64-66,187-198), shipped architecture, 138 patches 32x32x4",
           "batch": TB_, "steps": kt, "ms_per_step": mst, "value": TB_ / (mst * 1e-3), "unit": "patches/s",
           "bound": "kernel latency: a chain of stream-ordered launches per step (profiles/)"}
    # the same step at the paper-scale coupling width (job_noise_flow.sh:19: "for Noise Flow it is 32"), fresh initialisation: the
    # patch-resident stages of csrc/nf_train_pr.h (the CNN recomputed from z between the batch-statistics barriers, every dense
    # stage on v_mfma_f32_32x32x2_f32, no [pixel][32] tensor in HBM)
    trw = Trainer([32, 32, 4], default_hps(width=32), device=dev.index, max_batch=TB_)
    for _ in range(3):
        trw.step(xt_, yt_, [0.0], [0.0], [100.0], [2.0], lr=1e-4, sync=False)
    torch.cuda.synchronize(dev)
    g0.record(stream)
    for _ in range(kt):
        trw.step(xt_, yt_, [0.0], [0.0], [100.0], [2.0], lr=1e-4, sync=False)
    g1.record(stream)
    torch.cuda.synchronize(dev)
    msw = g0.elapsed_time(g1) / kt
    trw.close()
    # matrix instructions a step issues per 32-pixel row tile and coupling (9 / 25 / 43 / 75 / 75 / 93 for the six stages),
    # each 32 x 32 x 2 MACs: the recomputation included — executed work, not the algorithmic 3 x forward
    mfma_flop = 320 * 2.0 * 32 * 32 * 2 * 32 * 8
    out["width32"] = {"workload": "the same step, coupling width 32 (fresh initialisation), 138 patches 32x32x4", "ms_per_step": msw,
                      "value": TB_ / (msw * 1e-3), "unit": "patches/s",
                      "executed_matrix_tflops": mfma_flop * TB_ / (msw * 1e-3) / 1e12,
                      "note": "patch-resident stages (csrc/nf_train_pr.h): 4 launches per coupling on the critical path (+ d l_last/W on "
                              "the side stream, in the idle CUs), one workgroup per patch on 138 of the CUs; fixed cost per launch "
                              "and the idle CUs bound it (DESIGN 4.7)"}
    try:   # ... and with the GPU full: 1 024 patches (one workgroup per CU walks 4 patches)
        xb_, yb_ = synth_patches(args.seed, 1 << 43, 1024, device=dev.index)
        trb = Trainer([32, 32, 4], default_hps(width=32), device=dev.index, max_batch=1024)
        for _ in range(2):
            trb.step(xb_, yb_, [0.0], [0.0], [100.0], [2.0], lr=1e-4, sync=False)
        torch.cuda.synchronize(dev)
        g0.record(stream)
        for _ in range(10):
            trb.step(xb_, yb_, [0.0], [0.0], [100.0], [2.0], lr=1e-4, sync=False)
        g1.record(stream)
        torch.cuda.synchronize(dev)
        msb = g0.elapsed_time(g1) / 10
        trb.close()
        del xb_, yb_
        out["width32"]["b1024"] = {"batch": 1024, "ms_per_step": msb, "value": 1024 / (msb * 1e-3), "unit": "patches/s",
                                    "executed_matrix_tflops": mfma_flop * 1024 / (msb * 1e-3) / 1e12,
                                    "executed_frac_of_f32_matrix_peak": mfma_flop * 1024 / (msb * 1e-3) / 1e12 / VALU_PEAK_TFLOPS}
    except Exception as ex:
        out["width32"]["b1024"] = {"error": str(ex)[:300]}
    # ... and at the width the reference's flags default to (sidd/ArgParser.py:43: 512), and at 64: the dense products of a step are the
    # hand-written matrix-core GEMMs of csrc/nf_train_mm.h (BN + ReLU fused into the operand staging, batch sums into the epilogues;
    # the 128-column products and the pixel-K filter gradients as fp32-ACCURATE products on v_mfma_f32_32x32x16_bf16: three-way
    # operand split, six partial products, "bf16 x 6") between kernels of run-time width (csrc/nf_train_gemm.h)
    for wg_, kg in ((512, 5), (64, 20)):
        key = "width%d" % wg_
        try:
            trg = Trainer([32, 32, 4], default_hps(width=wg_), device=dev.index, max_batch=TB_)
            for _ in range(2):
                trg.step(xt_, yt_, [0.0], [0.0], [100.0], [2.0], lr=1e-4, sync=False)
            torch.cuda.synchronize(dev)
            g0.record(stream)
            for _ in range(kg):
                trg.step(xt_, yt_, [0.0], [0.0], [100.0], [2.0], lr=1e-4, sync=False)
            g1.record(stream)
            torch.cuda.synchronize(dev)
            msg = g0.elapsed_time(g1) / kg
            trg.close()
            flop = 3.0 * 2.0 * (18 * wg_ + wg_ * wg_ + wg_ * 36) * 8 * TB_ * 1024      # forward + two transposed products per filter
            out[key] = {"workload": "the same step, coupling width %d (%sfresh initialisation), 138 patches 32x32x4"
                                    % (wg_, "the reference's default flag; " if wg_ == 512 else ""),
                        "steps": kg, "ms_per_step": msg, "value": TB_ / (msg * 1e-3), "unit": "patches/s",
                        "dense_tflops": flop / (msg * 1e-3) / 1e12,
                        "dense_frac_of_f32_matrix_peak": flop / (msg * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                        "note": "this repo's own matrix-core GEMMs (csrc/nf_train_mm.h): fp32-accurate products, the large ones on the "
                                "bf16 pipe as six partial products of three-way split operands — the fraction is against the f32 "
                                "matrix peak (157.3 TFLOP/s), which such products may exceed; no library GEMM"}
        except Exception as ex:    # reported, the other sections stand
            out[key] = {"error": str(ex)[:300]}
    return out


def _cpu_baseline(args, variables, xc, yc, B):
    import torch
    # (1) fused plain-C / OpenMP port of the reference arithmetic (oracle/nf_oracle.c): what a good
    #     CPU implementation does on all host cores -> the reported cpu_baseline
    from oracle.nf_oracle_c import COracle
    cc = COracle(ARCH_LABEL, variables)
    host_cores = _usable_cores(os.cpu_count() or 1)           # affinity mask and cgroup CPU quota
    cc.set_threads(host_cores)
    cc.nll(xc[:256], yc[:256], 100.0, 2.0)                      # warm-up (thread pool, page faults)
    n_batches, tc, budget = 0, 0.0, 0.6 * args.cpu_seconds
    t_start = time.perf_counter()
    while tc < budget and n_batches < 4096:                     # time-bounded sample
        cc.nll(xc, yc, 100.0, 2.0)
        n_batches += 1
        tc = time.perf_counter() - t_start
    cpu_baseline = {"value": n_batches * B / tc, "unit": "patches/s", "cores": _usable_cores(cc.threads()),
                    "kind": "port",
                    "sample": "forward NLL of %d batches x %d patches (same workload), fused plain-C fp32 port "
                              "of the reference arithmetic with OpenMP over patches (%d threads; TF1 "
                              "unavailable), %.1f s" % (n_batches, B, cc.threads(), tc)}
    # (2) op-per-layer torch-CPU port: how the TF1 graph actually executes (every op materialised)
    from oracle.nf_cpu_torch import TorchCpuFlow
    cpu = TorchCpuFlow(ARCH_LABEL, variables)
    torch.set_num_threads(host_cores)
    cpu.nll(xc[:128], yc[:128], 100.0, 2.0)
    n_batches, tc, budget = 0, 0.0, 0.4 * args.cpu_seconds
    t_start = time.perf_counter()
    while tc < budget and n_batches < 64:
        cpu.nll(xc, yc, 100.0, 2.0)
        n_batches += 1
        tc = time.perf_counter() - t_start
    cpu_baseline["op_per_layer_torch"] = {
        "value": n_batches * B / tc, "unit": "patches/s", "cores": _usable_cores(int(torch.get_num_threads())),
        "kind": "port",
        "sample": "forward NLL of %d batches x %d patches, torch-CPU fp32 op-per-layer restatement of the "
                  "TF1 graph, %.1f s" % (n_batches, B, tc)}
    return cpu_baseline


if __name__ == "__main__":
    main()
