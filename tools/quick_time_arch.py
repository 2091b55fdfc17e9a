"""Tuning aid: time an arbitrary arch.  python tools/quick_time_arch.py "sdn5|gain4" [B] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_flow_amd import NoiseFlow, default_hps
from noise_flow_amd.ckpt import load_checkpoint
from noise_flow_amd.patches import synth_patches
arch = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
v = load_checkpoint(os.path.join(root, "models/NoiseFlow/ckpt/model.ckpt.best"))
m = NoiseFlow([32, 32, 4], False, default_hps(arch=arch), variables=v)
x, y = synth_patches(0, 0, B)
out = torch.empty_like(x)
for name, fn, nbytes in (("nll", lambda: m.nll_sums(x, y, [0], [0], [100], [2]), 32768),
                         ("sample(eps given)", lambda: m.sample(y, 1.0, y, [0], [0], [100], [2], eps=x), 49152)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print("%s B=%d %s: %.3f ms  %.3e patches/s  %.0f GB/s algorithmic" % (arch, B, name, dt * 1e3, B / dt, B * nbytes / dt / 1e9))
