"""Profiling target: a handful of launches of the fused NLL kernel (rocprofv3 -- python tools/prof_nll.py [B] [n])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_flow_amd import NoiseFlow, default_hps
from noise_flow_amd.ckpt import load_checkpoint
from noise_flow_amd.patches import synth_patches
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
v = load_checkpoint(os.path.join(root, "models/NoiseFlow/ckpt/model.ckpt.best"))
m = NoiseFlow([32, 32, 4], False, default_hps(), variables=v)
x, y = synth_patches(0, 0, B)
for _ in range(n):
    m.nll_sums(x, y, [0], [0], [100], [2])
torch.cuda.synchronize()
