"""The driver's bench contract: `python bench.py` prints exactly ONE JSON line on stdout with the
agreed keys (run here with a handful of steps and without the CPU-baseline leg)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "2", "--ramp-ms", "0",
                          "--no-cpu-baseline"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().split("\n") if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "patches/s" and d["scaling"] == "weak" and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["traffic"] is None or r["traffic"] > 0
    # the parity leg ran (rank 0, N = 1): GPU mean NLL within tolerance of the fp64 oracle
    assert d["nll_check"]["max_rel_err_per_patch"] <= d["nll_check"]["tolerance"]
    assert d["sampling"]["value"] > 0 and d["training"]["value"] > 0 and d["two_streams"]["value"] > 0
    # every measured BASELINE config carries the same roofline block as the headline (configs[2] sampling, configs[4] fp16 CNN)
    for sec in ("sampling", "fp16_cnn_64x64"):
        rr = d[sec]["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "kernel_ms", "traffic", "hbm"):
            assert k in rr, (sec, k)
        assert abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-9 and 0 < rr["hbm"]["frac"] < 1
    assert d["fp16_cnn_64x64"]["roofline"]["peak"] == 2500.0 and d["fp16_cnn_64x64"]["roofline"]["shape_peak"]["frac"] > 0
    # both flop counts in the line: SURVEY 8(d)'s algorithmic 5.1 MFLOP per patch and what the kernels execute (144-MAC l_last)
    for rr in (r, d["sampling"]["roofline"]):
        assert 0 < rr["executed_flop_per_launch"] < rr["algorithmic_flop_per_launch"] and 0 < rr["frac_executed"] < rr["frac"] * 1.05
    # the trainer at the widths without stage kernels: this repo's own GEMMs (no library GEMM in the note, a measured fraction)
    for key in ("width512", "width64"):
        tw = d["training"][key]
        assert "error" not in tw and tw["ms_per_step"] > 0 and 0 < tw["dense_frac_of_f32_matrix_peak"] < 1 and "no library GEMM" in tw["note"]
    w512 = d["wide_cnn"]["w512"]     # Glow's default width (sidd/ArgParser.py:43) on the LDS-staged GEMM kernel
    assert w512["finite"] and w512["value"] > 0 and "nf_gemm_kernel" in w512["kernel_path"] and w512["roofline"]["frac"] > 0.3
    lp = d["large_patches"]          # 256x256 images as overlapping tiles
    assert lp["finite"] and lp["segments"] >= 1 and lp["pixels_per_s"] > 1e9, lp


def test_bench_multi_rank_leg_under_torchrun_on_one_gpu():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank) — on this one-GPU box with
    both ranks on GPU 0 and gloo carrying the all-reduce (RCCL refuses two ranks on one device): the N > 1 leg (BASELINE
    configs[3]: sharded resident patch range, one all-reduce per evaluation) runs end to end and prints the contract line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, NF_BENCH_BACKEND="gloo", NF_BENCH_ONE_GPU="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--total-patches", "65536", "--ramp-ms", "0"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().split("\n") if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "strong" and d["unit"] == "patches/s"
    assert d["config"]["total_patches"] == 65536 and d["config"]["patches_per_gpu"] == 32768
    assert abs(d["value"] - 65536 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert len(d["per_rank"]["kernel_ms_per_step"]) == 2 and d["mean_nll_identical_across_steps"] is True
    assert d["roofline"]["frac"] > 0 and d["collective"]["per_step"] == 1
    # the sharded mean equals the single-process evaluation of the same patch range (any sharding holds the same data)
    import torch
    from conftest import SHIPPED_CKPT
    from noise_flow_amd import NoiseFlow, default_hps
    from noise_flow_amd.ckpt import load_checkpoint
    from noise_flow_amd.dist import ResidentShard, evaluate_sharded
    m = NoiseFlow([32, 32, 4], False, default_hps(), variables=load_checkpoint(SHIPPED_CKPT))
    shard = ResidentShard(m, 0, 65536, 0, 1)
    mean, sd, n = evaluate_sharded(shard.eval_chunk(), 65536, 65536, 0, 1, torch.zeros(3, dtype=torch.float64, device="cuda"))
    assert n == 65536 and abs(mean - d["mean_nll"]) <= 1e-9 * abs(mean) and abs(sd - d["sd_z"]) <= 1e-9 * sd


def test_bench_multi_rank_leg_without_torchrun_self_spawns():
    """Plain `python bench.py --gpus 2` (no torch.distributed.run, no WORLD_SIZE): bench.py launches its own two ranks.  On this
    one-GPU box both ranks share GPU 0 and gloo carries the all-reduce.  The line carries configs[3] as the headline, configs[4]
    (64x64, fp16 CNN, sharded the same way) as a nested section with its own roofline, and rank 0's CPU baseline."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(NF_BENCH_BACKEND="gloo", NF_BENCH_ONE_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--total-patches", "32768", "--ramp-ms", "0", "--cpu-seconds", "2"], cwd=ROOT, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().split("\n") if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["total_patches"] == 32768
    assert d["per_rank"]["ranks"] == 2 and len(d["per_rank"]["kernel_ms_per_step"]) == 2
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    c5 = d["fp16_cnn_64x64_sharded"]
    assert "error" not in c5, c5
    assert c5["config"]["patch"] == "64x64x4" and c5["config"]["total_patches"] == 8192 and c5["config"]["patches_per_gpu"] == 4096
    assert c5["roofline"]["peak"] == 2500.0 and c5["roofline"]["frac"] > 0 and c5["roofline"]["hbm"]["frac"] > 0
    assert c5["mean_nll_identical_across_steps"] is True and c5["value"] > 0
    # the fp16-CNN mean NLL of the sharded 64x64 range is the model's own answer on the same patches
    import torch
    from conftest import SHIPPED_CKPT
    from noise_flow_amd import NoiseFlow, default_hps
    from noise_flow_amd.ckpt import load_checkpoint
    from noise_flow_amd.dist import ResidentShard, evaluate_sharded
    m = NoiseFlow([64, 64, 4], False, default_hps(), variables=load_checkpoint(SHIPPED_CKPT), cnn_dtype="fp16")
    shard = ResidentShard(m, 0, 8192, 0, 1, 64, 64)
    mean, sd, n = evaluate_sharded(shard.eval_chunk(), 8192, 8192, 0, 1, torch.zeros(3, dtype=torch.float64, device="cuda"))
    assert n == 8192 and abs(mean - c5["mean_nll"]) <= 1e-9 * abs(mean)


def test_bench_c5_headline_on_one_rank():
    """`--config c5` makes BASELINE configs[4] the headline of the sharded leg (here: one rank, NF_BENCH_FORCE_DIST)."""
    env = dict(os.environ, NF_BENCH_FORCE_DIST="1", NF_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--config", "c5",
                          "--total-patches", "4096", "--ramp-ms", "0", "--no-cpu-baseline"], cwd=ROOT, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    d = json.loads([l for l in out.stdout.decode().split("\n") if l.strip().startswith("{")][0])
    assert d["config"]["patch"] == "64x64x4" and d["config"]["total_patches"] == 4096 and d["dtype"].startswith("f16")
    assert d["roofline"]["peak"] == 2500.0 and "fp16_cnn_64x64_sharded" not in d
