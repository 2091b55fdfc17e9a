"""Tuning aid: fp32 vs fp16-CNN mode timing. Usage: python tools/quick_time_fp16.py [H] [B] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_flow_amd import _lib as _nf_lib
if os.environ.get("NF_TOOL_LIB"):   # A/B a differently-built library (this tool only)
    _nf_lib.LIB_PATH = os.environ["NF_TOOL_LIB"]
from noise_flow_amd import NoiseFlow, default_hps
from noise_flow_amd.ckpt import load_checkpoint
from noise_flow_amd.patches import synth_patches
H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
v = load_checkpoint(os.path.join(root, "models/NoiseFlow/ckpt/model.ckpt.best"))
x, y = synth_patches(0, 0, B, H, H)
for mode in os.environ.get("MODES", "fp32,fp16").split(","):
    m = NoiseFlow([H, H, 4], False, default_hps(), variables=v, cnn_dtype=mode)
    for name, fn in (("nll", lambda: m.nll_sums(x, y, [0], [0], [100], [2])), ("sample", lambda: m.sample(y, 1.0, y, [0], [0], [100], [2]))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        print("%dx%d B=%d cnn=%s %s: %.3f ms  %.3e patches/s  (%.3e pixels/s)" % (H, H, B, mode, name, dt * 1e3, B / dt, B * H * H / dt))
