"""Host-fed (PCIe-inclusive) rates of the Python surface: numpy in -> numpy out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from noise_flow_amd import NoiseFlow, NoiseFlowWrapper, default_hps
from noise_flow_amd.ckpt import load_checkpoint
v = load_checkpoint("models/NoiseFlow/ckpt/model.ckpt.best")
m = NoiseFlow([32, 32, 4], False, default_hps(), variables=v)
rng = np.random.RandomState(0)
for B in (64, 1024, 4096):
    y = rng.rand(B, 32, 32, 4).astype(np.float32)
    x = (rng.randn(B, 32, 32, 4) * 0.02).astype(np.float32)
    y64, x64 = y.astype(np.float64), x.astype(np.float64)      # the reference's minibatch dtype (quirk Q9)
    for name, fn in (("loss(float32 numpy)", lambda: m.loss(x, y, [0], [0], [100], [2])),
                     ("loss(float64 numpy)", lambda: m.loss(x64, y64, [0], [0], [100], [2])),
                     ("sample(float32 numpy)", lambda: m.sample(y, 0.6, y, [0], [0], [100], [2]))):
        for _ in range(3):
            fn()
        n = 20
        t = time.perf_counter()
        for _ in range(n):
            fn()
        dt = (time.perf_counter() - t) / n
        print("B=%5d %-24s %.3f ms/call  %.3e patches/s" % (B, name, dt * 1e3, B / dt))
