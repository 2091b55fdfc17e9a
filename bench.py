#!/usr/bin/env python
"""Benchmark of the Noise Flow hot path on MI355X — driver contract.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the fused likelihood-direction kernel (per-patch NLL +
log|det J| + batch sums) over ONE batch of 1024 synthetic 32x32x4 raw patches per
GPU — BASELINE.json configs[1], the configuration the metric is quoted on — with
the shipped Noise Flow checkpoint.  Inputs are resident in HBM before the timed
region; patches are a pure function of (seed, global patch index), so any
sharding evaluates the same data.  For N > 1 the patch range is sharded across
ranks (weak scaling: 1024 patches per GPU per step) and the evaluation ends with
ONE RCCL all-reduce of (sum nll, sum sd_z, count) inside the timed region.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      the dominant kernel against the dense fp32 matrix/vector peak (the binding
                roofline: algorithmic flops / HIP-event time), with the HBM view nested
  cpu_baseline  CPU ports of the reference arithmetic (oracle/): fused plain-C/OpenMP (the
                reported value) and the op-per-layer torch-CPU restatement of the TF1 graph,
                timed on this box's host cores on bounded samples (N = 1 only)
  sampling      the sampling direction at BASELINE configs[2] (batch 4096)
  nll_check     GPU mean NLL vs the fp64 CPU oracle on a 64-patch subset
  fp16_cnn_64x64  BASELINE configs[4] shape (64x64x4, fp16 coupling CNN / fp32 log-det)
  training      one training step (fwd batch-BN + bwd + EMA + Adam) at the reference's minibatch of 138
  two_streams   the headline workload with consecutive steps alternating between two HIP streams
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ARCH_LABEL = "sdn5|unc|unc|unc|unc|gain4|unc|unc|unc|unc"
ALGO_BYTES_PER_PATCH = 2 * 32 * 32 * 4 * 4          # read x and y once (fp32): 32768 B
ALGO_FLOP_PER_PATCH = 5.1e6
HBM_PEAK_GBS = 8000.0                                # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3                             # fp32 vector peak


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
    # defaults: 0.13 s of timed work after 13 ms of warm-up — a 20-step warm-up (1.3 ms) ends before the
    # GPU has left its idle clocks and reads ~5 % slow (measured: 62.7 us/step vs 59.5 us steady state)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=1024, help="patches per GPU per step (configs[1]: 1024)")
    ap.add_argument("--sample-batch", type=int, default=4096, help="sampling-direction batch (configs[2]: 4096)")
    ap.add_argument("--pool", type=int, default=16, help="distinct resident batches cycled through")
    ap.add_argument("--ramp-ms", type=float, default=250.0,
                    help="untimed clock ramp before the W warm-up steps: the same step repeated for this long, so that "
                         "a short W does not leave the GPU at its idle clocks (0 disables)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    return ap.parse_args()


def main():
    args = parse_args()
    # stdout carries exactly ONE line, the JSON, written last: native libraries (RCCL prints a version
    # banner through C stdio, flushed at exit AFTER Python's output) would otherwise follow it.  Everything
    # else that targets fd 1 goes to stderr; the JSON is written straight to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # NF_BENCH_FORCE_DIST=1 runs the RCCL code path on a single rank too (validation aid)
    use_dist = world > 1 or bool(os.environ.get("NF_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)   # "nccl" is RCCL on ROCm

    from noise_flow_amd import NoiseFlow, default_hps, _lib
    from noise_flow_amd.ckpt import load_checkpoint
    from noise_flow_amd.patches import synth_patches
    from noise_flow_amd.dist import allreduce_sums

    variables = load_checkpoint(os.path.join(ROOT, "models", "NoiseFlow", "ckpt", "model.ckpt.best"))
    model = NoiseFlow([32, 32, 4], False, default_hps(), variables=variables, device=local_rank)
    lib = _lib.load()
    B, K, Wm = args.batch, args.steps, args.warmup
    cond = _lib.nf_cond(100.0, 2.0, 0.000479, 0.000002)      # ISO 100, S6 (train_noise_flow.py:143-147)

    # ---- resident synthetic data: batch j of rank r holds patches [(j*world + r)*B, +B) ----
    pool = max(1, min(args.pool, K + Wm))
    batches = [synth_patches(args.seed, (j * world + rank) * B, B, device=local_rank) for j in range(pool)]
    sums = torch.zeros(3, dtype=torch.float64, device=dev)
    # per-workgroup partial sums go to the C ABI's slotted accumulator (NF_SUMS_WIDE: 64 slots on separate
    # cache lines); ONE nf_sums_reduce after the last step folds it into (sum nll, sum sd, count)
    wide = torch.zeros(_lib.NF_SUMS_SLOTS * _lib.NF_SUMS_STRIDE, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)
    sptr = int(stream.cuda_stream)
    hptr = model._flow.ptr

    def nll_step(i, flags=_lib.NF_ACCUMULATE | _lib.NF_SUMS_WIDE):
        x, y = batches[i % pool]
        rc = lib.nf_nll(hptr, x.data_ptr(), y.data_ptr(), B, C.byref(cond), None, None, None, None,
                        wide.data_ptr(), flags, sptr)
        if rc != 0:
            _lib.check(rc)

    def barrier():
        if use_dist:
            dist.barrier()

    if args.ramp_ms > 0:                    # untimed: leave the idle clocks (reported as "clock_ramp_ms")
        t_r = time.perf_counter()
        while (time.perf_counter() - t_r) * 1e3 < args.ramp_ms:
            for i in range(64):
                nll_step(i)
            torch.cuda.synchronize(dev)
    for i in range(Wm):
        nll_step(i)
    if use_dist:
        allreduce_sums(sums.clone())        # warm the RCCL communicator outside the timed region
    sums.zero_()
    wide.zero_()
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(K):
        nll_step(i)
    ev1.record(stream)
    _lib.check(lib.nf_sums_reduce(wide.data_ptr(), sums.data_ptr(), 0, sptr))   # inside the timed wall clock
    if use_dist:
        allreduce_sums(sums)                # ONE RCCL all-reduce of 3 fp64 scalars finishes the evaluation
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / K   # average launch duration over the timed region (HIP events)

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    s = sums.cpu().numpy()
    total_patches = world * B * K
    assert int(round(s[2])) == total_patches, (s, total_patches)
    value = total_patches / elapsed

    # ---- the same K steps alternating between TWO streams (rank 0): a 1024-patch launch fills the GPU exactly
    # once, so back-to-back launches on one stream each pay their own ramp-up and drain; on two streams the next
    # launch takes the workgroup slots the previous one frees (what noise_flow_amd.dist.flow_eval_chunk does) ----
    two_streams = None
    if rank == 0:
        try:
            side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            wide2 = torch.zeros_like(wide)
            torch.cuda.synchronize(dev)

            def nll_step2(i):
                x, y = batches[i % pool]
                rc = lib.nf_nll(hptr, x.data_ptr(), y.data_ptr(), B, C.byref(cond), None, None, None, None,
                                wide2.data_ptr(), _lib.NF_ACCUMULATE | _lib.NF_SUMS_WIDE, int(side[i & 1].cuda_stream))
                if rc != 0:
                    _lib.check(rc)
            for i in range(max(Wm, 50)):
                nll_step2(i)
            torch.cuda.synchronize(dev)
            wide2.zero_()
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            for i in range(K):
                nll_step2(i)
            torch.cuda.synchronize(dev)
            el2 = time.perf_counter() - t2
            s2 = torch.zeros(3, dtype=torch.float64, device=dev)
            _lib.check(lib.nf_sums_reduce(wide2.data_ptr(), s2.data_ptr(), 0, sptr))
            s2 = s2.cpu().numpy()
            assert int(round(s2[2])) == B * K
            two_streams = {"value": B * K / el2, "unit": "patches/s", "ms_per_step": el2 / K * 1e3, "steps": K, "streams": 2,
                           "mean_nll": float(s2[0] / s2[2]),
                           "note": "same workload and kernel, consecutive steps alternate between two HIP streams so that "
                                   "launches overlap their ramp-up / drain; the headline value stays single-stream so that "
                                   "roofline.kernel_ms agrees with rocprof's per-kernel duration"}
        except Exception as e:
            two_streams = {"error": str(e)}

    # ---- sampling direction (configs[2]): B = 4096, fixed cam / ISO, in-kernel Philox eps ----
    sampling = None
    nll_check = None
    cpu_baseline = None
    if rank == 0:
        SB = args.sample_batch
        _, ys = synth_patches(args.seed, 1 << 40, SB, device=local_rank, want_x=False)
        xs = torch.empty_like(ys)
        ks = max(10, min(K, 100))

        def sample_step(i):
            rc = lib.nf_sample(hptr, ys.data_ptr(), None, args.seed, i * SB, 1.0, SB, C.byref(cond), xs.data_ptr(), sptr)
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
        sampling = {"value": SB * ks / ts, "unit": "patches/s", "batch": SB, "steps": ks,
                    "ms_per_step": 1e3 * ts / ks, "kernel_ms": e0.elapsed_time(e1) / ks, "temp": 1.0,
                    "eps": "in-kernel Philox4x32-10", "workload": "configs[2]: inverse sampling, clean patch + fixed cam/ISO"}

    # ---- BASELINE configs[4] shape: 64x64x4 patches, fp16 coupling CNN / fp32 log-det (rank 0) ----
    fp16_cnn = None
    if rank == 0:
        try:
            m16 = NoiseFlow([64, 64, 4], False, default_hps(), variables=variables, device=local_rank, cnn_dtype="fp16")
            B16 = 1024
            x16, y16 = synth_patches(args.seed, 1 << 41, B16, 64, 64, device=local_rank)
            s16 = torch.zeros(_lib.NF_SUMS_SLOTS * _lib.NF_SUMS_STRIDE, dtype=torch.float64, device=dev)
            k16 = max(10, min(K, 50))

            def step16():
                rc = lib.nf_nll(m16._flow.ptr, x16.data_ptr(), y16.data_ptr(), B16, C.byref(cond), None, None, None,
                                None, s16.data_ptr(), _lib.NF_ACCUMULATE | _lib.NF_SUMS_WIDE, sptr)
                if rc != 0:
                    _lib.check(rc)
            for _ in range(5):
                step16()
            torch.cuda.synchronize(dev)
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record(stream)
            for _ in range(k16):
                step16()
            f1.record(stream)
            torch.cuda.synchronize(dev)
            ms16 = f0.elapsed_time(f1) / k16
            bytes16 = 2 * 64 * 64 * 4 * 4 * B16
            fp16_cnn = {"workload": "configs[4] shape: forward NLL, 64x64x4 patches, fp16 coupling CNN "
                                    "(v_mfma_f32_4x4x4_16b_f16, fp32 accumulate + fp32 log-det), fp32 I/O, 1 GPU",
                        "batch": B16, "steps": k16, "kernel_ms": ms16, "value": B16 / (ms16 * 1e-3), "unit": "patches/s",
                        "pixels_per_s": B16 * 4096 / (ms16 * 1e-3),
                        "hbm": {"achieved": bytes16 / (ms16 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": bytes16 / (ms16 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "algorithmic_bytes_per_launch": bytes16}}
            del m16, x16, y16
        except Exception as e:   # the headline metric must not depend on the optional mode
            fp16_cnn = {"error": str(e)}

    # ---- training step at the reference's minibatch of 138 patches (rank 0) ----
    training = None
    if rank == 0:
        try:
            from noise_flow_amd.train import Trainer
            TB_ = 138                                            # job_noise_flow.sh: --n_batch_train 138
            trn = Trainer([32, 32, 4], default_hps(), variables=variables, device=local_rank, max_batch=TB_)
            xt_, yt_ = synth_patches(args.seed, 1 << 42, TB_, device=local_rank)
            kt = max(10, min(K, 50))
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
            training = {"workload": "one training step = forward with batch-statistics BN + backward + BN EMA + Adam "
                                    "(train_noise_flow.py:64-66,187-198), shipped architecture, 138 patches 32x32x4",
                        "batch": TB_, "steps": kt, "ms_per_step": mst, "value": TB_ / (mst * 1e-3), "unit": "patches/s",
                        "bound": "kernel latency: ~120 stream-ordered launches per step (profiles/r01_train_kernel_stats.csv)"}
            trn.close()
            del trn, xt_, yt_
        except Exception as e:   # the headline metric must not depend on the optional section
            training = {"error": str(e)}

    # ---- parity + CPU baseline: rank 0, N = 1 only (oracle/ is the checker, never the product) ----
    if rank == 0 and world == 1:
        from oracle.nf_oracle import NoiseFlowOracle
        x0, y0 = batches[0]
        nb = 64
        nll_gpu, _ = model._loss(x0[:nb], y0[:nb], [0.0], [0.0], [100.0], [2.0])
        ref = NoiseFlowOracle(ARCH_LABEL, variables).nll(x0[:nb].cpu().numpy(), y0[:nb].cpu().numpy(), 100.0, 2.0)[0]
        g = nll_gpu.double().cpu().numpy()
        nll_check = {"patches": nb, "gpu_mean_nll": float(g.mean()), "cpu_fp64_mean_nll": float(ref.mean()),
                     "rel_err_mean": float(abs(g.mean() - ref.mean()) / abs(ref.mean())),
                     "max_rel_err_per_patch": float(np.max(np.abs(g - ref) / np.abs(ref))), "tolerance": 1e-5}
        if not args.no_cpu_baseline:
            xc, yc = x0.cpu().numpy(), y0.cpu().numpy()
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

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        gbs = ALGO_BYTES_PER_PATCH * B / (kernel_ms * 1e-3) / 1e9
        tfl = ALGO_FLOP_PER_PATCH * B / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "patches/sec (32x32x4) fwd-NLL and inverse-sample; mean NLL vs CPU ref",
            "value": value, "unit": "patches/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: full NoiseFlow arch (%s) forward NLL, batch %d synthetic 32x32x4 "
                                   "patches per GPU, shipped checkpoint, ISO 100 / cam S6" % (ARCH_LABEL, B),
                       "batch_per_gpu": B, "global_batch": B * world, "patch": "32x32x4",
                       "parallelism": "dp%d: patch-index sharding, one RCCL all-reduce of 3 fp64 scalars" % world},
            "mean_nll": float(s[0] / s[2]), "sd_z": float(s[1] / s[2]),
            # The fused kernel is compute-bound (~156 flop/B): its FMAs are v_mfma_f32_4x4x1 on the fp32
            # matrix/vector datapath (dense fp32 peak 157.3 TFLOP/s) -> that is the binding roofline.
            # The HBM view the task also asks for is nested under "hbm"; "traffic" = HBM bytes per
            # launch from the FETCH_SIZE / WRITE_SIZE PMC passes (profiles/traffic.json).
            "roofline": {"bound": "mfma", "achieved": tfl, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tfl / VALU_PEAK_TFLOPS, "traffic": traffic,
                         "dtype": "f32 (v_mfma_f32_4x4x1_16b_f32, exact fp32; shares the fp32 datapath with VALU)",
                         "kernel": "nf_flow_kernel<4,256,4,false,true,true,0>", "kernel_ms": kernel_ms,
                         "algorithmic_flop_per_launch": ALGO_FLOP_PER_PATCH * B,
                         "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PATCH * B,
                                 "note": "structurally capped near 13 %: 32 KiB and 5.1 MFLOP per patch"}},
            "cpu_baseline": cpu_baseline, "sampling": sampling, "nll_check": nll_check, "fp16_cnn_64x64": fp16_cnn,
            "training": training, "two_streams": two_streams, "clock_ramp_ms": args.ramp_ms,
        }
        line = json.dumps(out)
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


if __name__ == "__main__":
    main()
