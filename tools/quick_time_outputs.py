"""Tuning aid: cost of the NLL kernel's output modes (per-patch arrays, fp64 sums, latent)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_flow_amd import _lib, NoiseFlow, default_hps
from noise_flow_amd.ckpt import load_checkpoint
from noise_flow_amd.patches import synth_patches
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
v = load_checkpoint("models/NoiseFlow/ckpt/model.ckpt.best")
m = NoiseFlow([32, 32, 4], False, default_hps(), variables=v)
lib = _lib.load()
x, y = synth_patches(0, 0, B)
nll = torch.empty(B, device="cuda"); sd = torch.empty(B, device="cuda"); ld = torch.empty(B, device="cuda")
z = torch.empty_like(x); sums = torch.zeros(3, dtype=torch.float64, device="cuda")
cond = _lib.nf_cond(100.0, 2.0, 0.0, 0.0)
st = torch.cuda.current_stream().cuda_stream
modes = {"nll_out only": (nll.data_ptr(), None, None, None, None, 0),
         "sums only (accumulate)": (None, None, None, None, sums.data_ptr(), _lib.NF_ACCUMULATE),
         "sums only (memset each call)": (None, None, None, None, sums.data_ptr(), 0),
         "nll+sd+ld": (nll.data_ptr(), sd.data_ptr(), ld.data_ptr(), None, None, 0),
         "z_out only": (None, None, None, z.data_ptr(), None, 0),
         "all": (nll.data_ptr(), sd.data_ptr(), ld.data_ptr(), z.data_ptr(), sums.data_ptr(), _lib.NF_ACCUMULATE)}
def _warm():
    for _ in range(3000):   # ~0.2 s: leave the idle clocks before the first measurement
        lib.nf_nll(m._flow.ptr, x.data_ptr(), y.data_ptr(), B, C.byref(cond), nll.data_ptr(), None, None, None, None, 0, st)
    torch.cuda.synchronize()
_warm()
for rnd in range(3):
  for name, (a, b, c, d, e, fl) in modes.items():
    def fn():
        rc = lib.nf_nll(m._flow.ptr, x.data_ptr(), y.data_ptr(), B, C.byref(cond), a, b, c, d, e, fl, st)
        assert rc == 0
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("round %d B=%d %-30s %.4f ms  %.3e patches/s" % (rnd, B, name, ms, B / (ms * 1e-3)))
