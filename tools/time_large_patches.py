"""Timing aid: forward NLL / sampling rate in pixels per second at any patch shape (patches beyond 64x64 run as overlapping
tiles, csrc/nf_device.h).  python tools/time_large_patches.py H W B [iters] [width] [fp32|fp16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_flow_amd import _lib as _nf_lib
if os.environ.get("NF_TOOL_LIB"):   # A/B a differently-built library (this tool only)
    _nf_lib.LIB_PATH = os.path.abspath(os.environ["NF_TOOL_LIB"])
from noise_flow_amd import NoiseFlow, default_hps, params as _params
from noise_flow_amd.patches import synth_patches
H, W, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
width = int(sys.argv[5]) if len(sys.argv) > 5 else 4
dtype = sys.argv[6] if len(sys.argv) > 6 else "fp32"
hps = default_hps(width=width)
m = NoiseFlow([H, W, 4], False, hps, variables=_params.init_variables(hps.arch, width, 4, 1234), cnn_dtype=dtype)
x, y = synth_patches(0, 0, B, H, W)
eps = torch.randn_like(x)
for name, fn in (("nll", lambda: m.nll_sums(x, y, [0], [0], [100], [2])),
                 ("sample", lambda: m.sample(y, 1.0, y, [0], [0], [100], [2], eps=eps))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print("w%d %s %dx%d B=%d %s: %.3f ms  %.3e patches/s  %.3e pixels/s" % (width, dtype, H, W, B, name, dt * 1e3, B / dt, B * H * W / dt))
