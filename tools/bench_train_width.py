"""Training-step timing at other coupling widths (fresh initialisation): `python tools/bench_train_width.py 32 138`."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noise_flow_amd import _lib as _nf_lib
if os.environ.get("NF_TOOL_LIB"):   # A/B a differently-built library (this tool only)
    _nf_lib.LIB_PATH = os.path.abspath(os.environ["NF_TOOL_LIB"])
from noise_flow_amd import default_hps, patches
from noise_flow_amd.train import Trainer

width = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for B in [int(b) for b in (sys.argv[2] if len(sys.argv) > 2 else "138").split(",")]:
    x, y = patches.synth_patches(0, 0, B, nlf=(0.003696, 2e-6))
    tr = Trainer([32, 32, 4], default_hps(width=width), max_batch=B)
    for _ in range(3):
        tr.step(x, y, [0], [0], [800], [2], lr=1e-4, sync=False)
    torch.cuda.synchronize()
    steps = int(os.environ.get("NF_TOOL_STEPS", "30"))
    t = time.perf_counter()
    for _ in range(steps):
        tr.step(x, y, [0], [0], [800], [2], lr=1e-4, sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    print(json.dumps({"what": "training step", "width": width, "B": B, "ms_per_step": round(dt * 1e3, 3), "patches_per_s": round(B / dt, 1)}))
