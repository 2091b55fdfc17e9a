"""is_training=True evaluation (batch statistics) against the evaluation under running statistics, at coupling widths 32 / 4 / 64:  python tools/time_batchstats_width.py"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from noise_flow_amd import NoiseFlow, default_hps, patches, params as _params
for w in (32, 4, 64):
    hps = default_hps(width=w)
    var = _params.init_variables(hps.arch, w, 4, 1234)
    for B in (138, 1024):
        x, y = patches.synth_patches(0, 0, B)
        for training in (False, True):
            m = NoiseFlow([32, 32, 4], training, hps, variables=var)
            for _ in range(3):
                m.loss(x, y, [0], [0], [800], [2])
            torch.cuda.synchronize()
            t = time.perf_counter()
            n = 10
            for _ in range(n):
                m.loss(x, y, [0], [0], [800], [2])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / n
            print("w=%d B=%5d is_training=%-5s  %.3f ms/call  %.3e patches/s" % (w, B, training, dt * 1e3, B / dt))
