"""Profiling target: launches of the fused NLL / sampling kernel — first an untimed clock ramp of NF_PROF_RAMP_MS milliseconds
(default 250, like bench.py --ramp-ms: a kernel trace that starts on idle clocks reads 10 - 15 % slow), then n launches.
rocprofv3 ... -- python tools/prof_nll.py [B] [n] [H] [cnn_dtype] [nll|sample]
Under --kernel-trace --stats the ramp's launches are in the average too: they are the same kernel on the same data, and after the
first few milliseconds they run on the clocks the n launches see."""
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
import time


def launch(i):
    if what == "sample":      # in-kernel Philox eps (BASELINE configs[2])
        m.sample(y, 1.0, y, [0], [0], [100], [2], seed=i)
    else:
        m.nll_sums(x, y, [0], [0], [100], [2])


ramp_ms = float(os.environ.get("NF_PROF_RAMP_MS", "250"))
t0, i = time.perf_counter(), 0
while (time.perf_counter() - t0) * 1e3 < ramp_ms:
    launch(i)
    i += 1
    if i % 16 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
for k in range(n):
    launch(i + k)
torch.cuda.synchronize()
