"""Profiling target: a handful of launches of the fused NLL kernel.
rocprofv3 ... -- python tools/prof_nll.py [B] [n] [H] [cnn_dtype] [nll|sample]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_flow_amd import _lib as _nf_lib
if os.environ.get("NF_TOOL_LIB"):
    _nf_lib.LIB_PATH = os.environ["NF_TOOL_LIB"]
from noise_flow_amd import NoiseFlow, default_hps
from noise_flow_amd.ckpt import load_checkpoint
from noise_flow_amd.patches import synth_patches
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H = int(sys.argv[3]) if len(sys.argv) > 3 else 32
mode = sys.argv[4] if len(sys.argv) > 4 else "fp32"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
v = load_checkpoint(os.path.join(root, "models/NoiseFlow/ckpt/model.ckpt.best"))
m = NoiseFlow([H, H, 4], False, default_hps(), variables=v, cnn_dtype=mode)
x, y = synth_patches(0, 0, B, H, H)
what = sys.argv[5] if len(sys.argv) > 5 else "nll"
for i in range(n):
    if what == "sample":      # in-kernel Philox eps (BASELINE configs[2])
        m.sample(y, 1.0, y, [0], [0], [100], [2], seed=i)
    else:
        m.nll_sums(x, y, [0], [0], [100], [2])
torch.cuda.synchronize()
