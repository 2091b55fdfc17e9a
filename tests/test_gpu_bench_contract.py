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
