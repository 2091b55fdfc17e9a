"""Profiling target: forward-NLL launches at a coupling-CNN width, exactly the model / batch of bench.py's `wide_cnn` section
(fresh wide weights, perturbed last layer; 1024 patches at widths <= 32, 512 beyond), after an untimed clock ramp.
    rocprofv3 --kernel-trace --stats ... -- python tools/prof_wide.py <width> [fp32|fp16] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from noise_flow_amd import NoiseFlow, default_hps, params as _params
from noise_flow_amd.patches import synth_patches

w = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dt = sys.argv[2] if len(sys.argv) > 2 else "fp32"
n = int(sys.argv[3]) if len(sys.argv) > 3 else (50 if w <= 32 else 10)
hps = default_hps(width=w)
var = _params.init_variables(hps.arch, w, 4, 1234)
rng = np.random.RandomState(w)
for k in list(var):
    if k.endswith("l_last/W") or k.endswith("l_last/b"):
        var[k] = (0.02 * rng.randn(*var[k].shape) * min(1.0, (32.0 / w) ** 0.5)).astype(np.float32)
m = NoiseFlow([32, 32, 4], False, hps, variables=var, cnn_dtype=dt)
nb = 1024 if w <= 32 else 512
x, y = synth_patches(0, 1 << 42, nb)
ramp_ms = float(os.environ.get("NF_PROF_RAMP_MS", "250"))
t0, i = time.perf_counter(), 0
while (time.perf_counter() - t0) * 1e3 < ramp_ms or i < 3:
    m.nll_sums(x, y, [0], [0], [100], [2])
    i += 1
    if i % 8 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
for _ in range(n):
    m.nll_sums(x, y, [0], [0], [100], [2])
torch.cuda.synchronize()
print("width %d %s: %d ramp + %d launches of %d patches" % (w, dt, i, n, nb))
