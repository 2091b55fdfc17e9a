"""Per-workgroup phase timeline of the fused NLL kernel (where do the microseconds of a B = 1024 launch go?).

Needs a library built with -DNF_TIMELINE (csrc/nf_kernels.hip: thread 0 of every workgroup stamps the 100 MHz
s_memrealtime counter at the phase boundaries of its first patch into the sd_out buffer):

    hipcc ... -DNF_TIMELINE -c nf_kernels.hip -o k_tl.o ; hipcc -shared k_tl.o nf_wide.o nf_host.o nf_train.o -o libnf_timeline.so
    NF_TIMELINE_LIB=noise_flow_amd/csrc/libnf_timeline.so python tools/timeline.py [B] [out.json] [H] [fp32|fp16]
(bash tools/build_variant.sh tl -DNF_TIMELINE builds one as build/variants/lib_tl.so)

Stamps: 0 entry, 1 after LDS set-up (+weight image) and the first barrier, 2 inputs arrived (after the sdn layer),
3..10 after each of the 8 couplings, 11 after the epilogue — of the workgroup's MIDDLE patch (its first when B <= grid) —, 12 kernel exit.  The library is
loaded INSTEAD of the product library for this tool only."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from noise_flow_amd import _lib
if os.environ.get("NF_TIMELINE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NF_TIMELINE_LIB"])
from noise_flow_amd import NoiseFlow, default_hps
from noise_flow_amd.ckpt import load_checkpoint
from noise_flow_amd.patches import synth_patches

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
out_path = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
H = int(sys.argv[3]) if len(sys.argv) > 3 else 32
mode = sys.argv[4] if len(sys.argv) > 4 else "fp32"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = NoiseFlow([H, H, 4], False, default_hps(), variables=load_checkpoint(os.path.join(root, "models/NoiseFlow/ckpt/model.ckpt.best")),
              cnn_dtype=mode)
x, y = synth_patches(0, 0, B, H, H)
lib = _lib.load()
grid = min(B, 1024)
dbg = torch.zeros(grid * 16, dtype=torch.int64, device="cuda")
nll = torch.empty(B, device="cuda")
wide = torch.zeros(_lib.NF_SUMS_SLOTS * _lib.NF_SUMS_STRIDE, dtype=torch.float64, device="cuda")
cond = _lib.nf_cond(100, 2, 0, 0)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(200):      # leave the idle clocks
    _lib.check(lib.nf_nll(m._flow.ptr, x.data_ptr(), y.data_ptr(), B, C.byref(cond), nll.data_ptr(), dbg.data_ptr(), None, None,
                          wide.data_ptr(), _lib.NF_ACCUMULATE | _lib.NF_SUMS_WIDE, None))
torch.cuda.synchronize()
runs = []
for _ in range(5):
    dbg.zero_()
    torch.cuda.synchronize()
    ev0.record()
    _lib.check(lib.nf_nll(m._flow.ptr, x.data_ptr(), y.data_ptr(), B, C.byref(cond), nll.data_ptr(), dbg.data_ptr(), None, None,
                          wide.data_ptr(), _lib.NF_ACCUMULATE | _lib.NF_SUMS_WIDE, None))
    ev1.record()
    torch.cuda.synchronize()
    runs.append((ev0.elapsed_time(ev1) * 1e3, dbg.cpu().numpy().reshape(grid, 16).copy()))
runs.sort(key=lambda r: r[0])
ev_us, s = runs[len(runs) // 2]
s = s[s[:, 0] != 0]                               # the persistent grid can be smaller than min(B, 1024)
grid = len(s)
s = s[:, :13].astype(np.float64) * 0.01          # 100 MHz ticks -> us
t0 = s[:, 0].min()
s -= t0
names = ["entry", "lds set-up + barrier", "inputs arrived (sdn done)"] + ["coupling %d" % i for i in range(1, 9)] + ["epilogue", "exit"]
rep = {"B": B, "grid": grid, "event_us": ev_us, "kernel_span_us": float(s[:, 12].max()),
       "first_entry_to_last_entry_us": float(s[:, 0].max()), "phases": []}
print("B=%d grid=%d: HIP-event time %.1f us; first entry -> last exit %.1f us; last workgroup enters at %.1f us"
      % (B, grid, ev_us, s[:, 12].max(), s[:, 0].max()))
print("%-28s %10s %10s %10s | %s" % ("phase (per workgroup)", "mean us", "p5", "p95", "ends at (mean / max) us"))
for i in range(1, 13):
    d = s[:, i] - s[:, i - 1]
    rep["phases"].append({"phase": names[i], "mean_us": float(d.mean()), "p5_us": float(np.percentile(d, 5)),
                          "p95_us": float(np.percentile(d, 95)), "ends_mean_us": float(s[:, i].mean()), "ends_max_us": float(s[:, i].max())})
    print("%-28s %10.2f %10.2f %10.2f | %8.2f / %8.2f" % (names[i], d.mean(), np.percentile(d, 5), np.percentile(d, 95), s[:, i].mean(), s[:, i].max()))
busy = s[:, 12] - s[:, 0]
print("workgroup lifetime: mean %.2f us (p5 %.2f, p95 %.2f)" % (busy.mean(), np.percentile(busy, 5), np.percentile(busy, 95)))
rep["workgroup_lifetime_mean_us"] = float(busy.mean())
if out_path:
    with open(out_path, "w") as f:
        json.dump(rep, f, indent=1)
