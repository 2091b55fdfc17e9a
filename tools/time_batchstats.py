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

# the C call alone (caller-owned buffers, no Python-side EMA): what nf_nll_batchstats itself costs
import ctypes as C
from noise_flow_amd import _lib
lib = _lib.load()
for B in (138, 1024):
    x, y = patches.synth_patches(0, 0, B)
    m = NoiseFlow([32, 32, 4], True, default_hps(), variables=v)
    nll = torch.empty(B, device="cuda")
    mom = np.zeros((8, 4, 4), np.float32)
    cond = _lib.nf_cond(800.0, 2.0, 0.0, 0.0)
    st = torch.cuda.current_stream().cuda_stream
    def call():
        assert lib.nf_nll_batchstats(m._flow.ptr, x.data_ptr(), y.data_ptr(), B, C.byref(cond), nll.data_ptr(), None, None, None, None, 0,
                                     mom.ctypes.data, st) == 0
    for _ in range(3):
        call()
    t = time.perf_counter()
    for _ in range(20):
        call()
    dt = (time.perf_counter() - t) / 20
    print("B=%5d nf_nll_batchstats (C ABI, incl. its final stream sync): %.3f ms/call" % (B, dt * 1e3))
