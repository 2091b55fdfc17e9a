"""Debug aid: cycle stamps of wave 0 / workgroup 0.  Needs a library built from a kernel source
instrumented with NF_STAMP() cycle-counter writes into the logdet buffer (see git history of
this file's commit); pass its path as NF_TIMELINE_LIB — it is loaded INSTEAD of the product
library for this tool only."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from noise_flow_amd import _lib
if os.environ.get("NF_TIMELINE_LIB"):
    _lib.LIB_PATH = os.environ["NF_TIMELINE_LIB"]
from noise_flow_amd import NoiseFlow, default_hps
from noise_flow_amd.ckpt import load_checkpoint
from noise_flow_amd.patches import synth_patches
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = NoiseFlow([32, 32, 4], False, default_hps(), variables=load_checkpoint(os.path.join(root, "models/NoiseFlow/ckpt/model.ckpt.best")))
x, y = synth_patches(0, 0, B)
lib = _lib.load()
dbg = torch.zeros(4096, dtype=torch.int64, device="cuda")
nll = torch.empty(B, device="cuda")
cond = _lib.nf_cond(100, 2, 0, 0)
for _ in range(3):
    dbg.zero_()
    _lib.check(lib.nf_nll(m._flow.ptr, x.data_ptr(), y.data_ptr(), B, C.byref(cond), nll.data_ptr(), None, dbg.data_ptr(), None, None, 0, None))
    torch.cuda.synchronize()
s = dbg.cpu().numpy()
s = s[s != 0]
d = np.diff(s)
print("stamps", len(s), "total cycles", s[-1] - s[0])
print("prologue(load x)", d[0])
i = 1
names = ["op-start->", "step1(z0 store)", "barrier1", "l_1 mfma", "l_2+store", "barrier2", "l_last mfma", "tail"]
print(list(d[:60]))
