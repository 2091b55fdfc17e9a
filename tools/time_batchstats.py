"""Wall time of one batch-statistics call (2 statistics passes per coupling + the fused pass)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from noise_flow_amd import NoiseFlow, default_hps, patches
from noise_flow_amd.ckpt import load_checkpoint

v = load_checkpoint("models/NoiseFlow/ckpt/model.ckpt.best")
for B in (138, 1024, 4096):
    x, y = patches.synth_patches(0, 0, B)
    for training in (False, True):
        m = NoiseFlow([32, 32, 4], training, default_hps(), variables=v)
        for _ in range(3):
            m.loss(x, y, [0], [0], [800], [2])
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 10
        for _ in range(n):
            m.loss(x, y, [0], [0], [800], [2])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        print("B=%5d is_training=%-5s  %.3f ms/call  %.3e patches/s" % (B, training, dt * 1e3, B / dt))
