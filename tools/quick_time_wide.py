"""Tuning aid: time the forward NLL / sampling at a coupling-CNN width.  python tools/quick_time_wide.py [width] [B] [H] [iters] [fp32|fp16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from noise_flow_amd import _lib as _nf_lib
if os.environ.get("NF_TOOL_LIB"):   # A/B a differently-built library (this tool only)
    _nf_lib.LIB_PATH = os.path.abspath(os.environ["NF_TOOL_LIB"])
from noise_flow_amd import NoiseFlow, default_hps, params as _params
from noise_flow_amd.patches import synth_patches
w = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H = int(sys.argv[3]) if len(sys.argv) > 3 else 32
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dtype = sys.argv[5] if len(sys.argv) > 5 else "fp32"
hps = default_hps(width=w)
var = _params.init_variables(hps.arch, w, 4, 1234)
rng = np.random.RandomState(w)
for k in list(var):
    if k.endswith("l_last/W") or k.endswith("l_last/b"):
        var[k] = (0.02 * rng.randn(*var[k].shape)).astype(np.float32)
m = NoiseFlow([H, H, 4], False, hps, variables=var, cnn_dtype=dtype)
x, y = synth_patches(0, 0, B, H, H)
eps = torch.randn_like(x)
print("kernel path", m._flow.lib.nf_kernel_path(m._flow.ptr, 0))
for name, fn in (("nll", lambda: m.nll_sums(x, y, [0], [0], [100], [2])),
                 ("sample_eps", lambda: m.sample(y, 1.0, y, [0], [0], [100], [2], eps=eps))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    mac = 16 + 18 * w + w * w + 36 * (w + 1)
    fl = (8 * (2 * mac + 56) + 40) * H * H * B
    print("w=%d B=%d %dx%d %s: %.3f ms  %.3e patches/s  %.1f TFLOP/s (%.1f %% of 157.3)" % (w, B, H, H, name, dt * 1e3, B / dt, fl / dt / 1e12, fl / dt / 1.573e12))
