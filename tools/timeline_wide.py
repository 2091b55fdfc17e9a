"""Per-workgroup phase timeline of the width-32 kernel (nf_wide.hip): where do the microseconds of a patch go?

Needs a library whose nf_wide.hip was built with -DNF_TIMELINE (thread 0 of every workgroup stamps the 100 MHz s_memrealtime
counter at the phase boundaries of its middle patch into the sd_out buffer):
    SRC=noise_flow_amd/csrc/nf_wide.hip OBJ=nf_wide bash tools/build_variant.sh wtl -DNF_TIMELINE
    NF_TOOL_LIB=build/variants/lib_wtl.so python tools/timeline_wide.py [B] [fp32|fp16] [H]
Stamps (nf_wide.hip): 0 patch start, 1 inputs in registers, per coupling c: 2+4c weights staged, +1 phase B done (wavefront 0),
+2 every wavefront's phase B done (barrier), +3 phase C done; 40 outputs written, 41 patch done."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from noise_flow_amd import _lib
if os.environ.get("NF_TOOL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NF_TOOL_LIB"])
from noise_flow_amd import NoiseFlow, default_hps, params as _params
from noise_flow_amd.patches import synth_patches

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
mode = sys.argv[2] if len(sys.argv) > 2 else "fp16"
H = int(sys.argv[3]) if len(sys.argv) > 3 else 32
hps = default_hps(width=32)
var = _params.init_variables(hps.arch, 32, 4, 1234)
m = NoiseFlow([H, H, 4], False, hps, variables=var, cnn_dtype=mode)
x, y = synth_patches(0, 0, B, H, H)
lib = _lib.load()
G = 4096                                            # at least the launch's grid
dbg = torch.zeros(G * 64, dtype=torch.int64, device="cuda")
nll = torch.empty(B, device="cuda")
wide = torch.zeros(_lib.NF_SUMS_SLOTS * _lib.NF_SUMS_STRIDE, dtype=torch.float64, device="cuda")
cond = _lib.nf_cond(100, 2, 0, 0)
def run():
    _lib.check(lib.nf_nll(m._flow.ptr, x.data_ptr(), y.data_ptr(), B, C.byref(cond), nll.data_ptr(), dbg.data_ptr(), None, None,
                          wide.data_ptr(), _lib.NF_ACCUMULATE | _lib.NF_SUMS_WIDE, None))
for _ in range(20):
    run()
torch.cuda.synchronize()
dbg.zero_()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
ev0.record(); run(); ev1.record()
torch.cuda.synchronize()
s = dbg.cpu().numpy().reshape(G, 64)
s = s[s[:, 41] != 0].astype(np.float64) * 0.01     # us
print("B=%d %s %dx%d: launch %.1f us, %d workgroups stamped" % (B, mode, H, H, ev0.elapsed_time(ev1) * 1e3, len(s)))
def d(i, j):
    v = s[:, j] - s[:, i]
    return "%7.2f us (p5 %.2f, p95 %.2f)" % (v.mean(), np.percentile(v, 5), np.percentile(v, 95))
print("patch start -> inputs in registers      ", d(0, 1))
print("inputs -> first coupling's weights staged", d(1, 2))
for c in range(8):
    b = 2 + 4 * c
    print("coupling %d: phase B (wavefront 0) %s | wait for the others %s | phase C %s | to the next coupling's staged weights %s"
          % (c, d(b, b + 1), d(b + 1, b + 2), d(b + 2, b + 3), d(b + 3, b + 4) if c < 7 else d(b + 3, 40)))
print("epilogue                                ", d(40, 41))
print("whole patch                             ", d(0, 41))
