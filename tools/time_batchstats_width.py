"""Batch-statistics (is_training=True) evaluation and sampling at other coupling widths, fresh initialisation."""
import sys, time
import torch
sys.path.insert(0, ".")
from noise_flow_amd import NoiseFlow, default_hps, patches

width = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for B in (138, 1024):
    x, y = patches.synth_patches(0, 0, B)
    for training in (False, True):
        m = NoiseFlow([32, 32, 4], training, default_hps(width=width))
        for what, fn in (("loss", lambda: m.loss(x, y, [0], [0], [800], [2])), ("sample", lambda: m.sample(y, 1.0, y, [0], [0], [800], [2]))):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            n = 5
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / n
            print("width %d B=%5d is_training=%-5s %-6s %.3f ms/call  %.3e patches/s" % (width, B, training, what, dt * 1e3, B / dt))
