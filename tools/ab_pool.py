"""Tuning aid: bench-like timing (rotating pool of HBM-resident batches, slotted sums) for an
optionally different build of the library.  Usage: [NF_TOOL_LIB=...] python tools/ab_pool.py [B] [steps]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_flow_amd import _lib as L
if os.environ.get("NF_TOOL_LIB"):
    L.LIB_PATH = os.environ["NF_TOOL_LIB"]
from noise_flow_amd import NoiseFlow, default_hps
from noise_flow_amd.ckpt import load_checkpoint
from noise_flow_amd.patches import synth_patches
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
v = load_checkpoint("models/NoiseFlow/ckpt/model.ckpt.best")
m = NoiseFlow([32, 32, 4], False, default_hps(), variables=v)
lib = L.load()
pool = [synth_patches(0, j * B, B) for j in range(16)]
wide = torch.zeros(L.NF_SUMS_SLOTS * L.NF_SUMS_STRIDE, dtype=torch.float64, device="cuda")
cond = L.nf_cond(100.0, 2.0, 0.0, 0.0)
st = torch.cuda.current_stream().cuda_stream
def step(i):
    x, y = pool[i % 16]
    assert lib.nf_nll(m._flow.ptr, x.data_ptr(), y.data_ptr(), B, C.byref(cond), None, None, None, None, wide.data_ptr(),
                      L.NF_ACCUMULATE | L.NF_SUMS_WIDE, st) == 0
for i in range(3000):
    step(i)
torch.cuda.synchronize()
for rnd in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(K):
        step(i)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    print("%s B=%d round %d: %.4f ms  %.3e patches/s" % (os.path.basename(L.LIB_PATH), B, rnd, ms, B / (ms * 1e-3)))
