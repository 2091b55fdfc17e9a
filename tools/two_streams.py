"""Tuning aid: do consecutive 1024-patch launches overlap their tails when they alternate between
two HIP streams?  (each launch fills the GPU exactly once: 1024 workgroups = the resident capacity)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_flow_amd import _lib as L, NoiseFlow, default_hps
from noise_flow_amd.ckpt import load_checkpoint
from noise_flow_amd.patches import synth_patches
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
v = load_checkpoint("models/NoiseFlow/ckpt/model.ckpt.best")
m = NoiseFlow([32, 32, 4], False, default_hps(), variables=v)
lib = L.load()
pool = [synth_patches(0, j * B, B) for j in range(16)]
cond = L.nf_cond(100.0, 2.0, 0.0, 0.0)
for ns in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    wides = [torch.zeros(L.NF_SUMS_SLOTS * L.NF_SUMS_STRIDE, dtype=torch.float64, device="cuda") for _ in range(ns)]
    def step(i):
        x, y = pool[i % 16]
        s = i % ns
        assert lib.nf_nll(m._flow.ptr, x.data_ptr(), y.data_ptr(), B, C.byref(cond), None, None, None, None,
                          wides[s].data_ptr(), L.NF_ACCUMULATE | L.NF_SUMS_WIDE, streams[s].cuda_stream) == 0
    for i in range(3000):
        step(i)
    torch.cuda.synchronize()
    import time
    for rnd in range(2):
        t0 = time.perf_counter()
        for i in range(K):
            step(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print("streams=%d B=%d round %d: %.4f ms/launch  %.3e patches/s" % (ns, B, rnd, dt * 1e3, B / dt))
