"""Training-step timing: one `sess.run([train_op, loss, sd_z])` equivalent (forward with batch BN,
backward, BN EMA, Adam) on the shipped architecture, minibatch sizes of job_noise_flow.sh (138)
and larger.  Prints one JSON line per batch size; `--cpu` also times the torch-CPU autograd port
(oracle/nf_grad_oracle.py, float64, host cores) on a bounded number of steps."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noise_flow_amd import default_hps, patches
from noise_flow_amd.train import Trainer
from noise_flow_amd.ckpt import load_checkpoint

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="138,1024,4096")
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--cpu", action="store_true")
a = ap.parse_args()
v = load_checkpoint("models/NoiseFlow/ckpt/model.ckpt.best")
for B in [int(b) for b in a.batches.split(",")]:
    x, y = patches.synth_patches(0, 0, B, nlf=(0.003696, 2e-6))
    tr = Trainer([32, 32, 4], default_hps(), variables=v, max_batch=B)
    for _ in range(5):
        tr.step(x, y, [0], [0], [800], [2], lr=1e-4, sync=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(a.steps):
        tr.step(x, y, [0], [0], [800], [2], lr=1e-4, sync=False)
    host = (time.perf_counter() - t) / a.steps       # host time to enqueue one step
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / a.steps
    out = {"what": "training step (fwd batch-BN + bwd + EMA + Adam)", "B": B, "ms_per_step": round(dt * 1e3, 4),
           "patches_per_s": round(B / dt, 1), "steps": a.steps,
           "host_enqueue_ms_per_step": round(host * 1e3, 4)}
    if a.cpu and B <= 138:
        from oracle.nf_grad_oracle import train_step
        xs, ys = x.cpu().numpy(), y.cpu().numpy()
        st, vv = {}, dict(v)
        t = time.perf_counter()
        n = 0
        while time.perf_counter() - t < 10.0:
            vv, _, _ = train_step("sdn5|unc|unc|unc|unc|gain4|unc|unc|unc|unc", vv, xs, ys, 800, 2, st, 1e-4)
            n += 1
        cdt = (time.perf_counter() - t) / n
        out["cpu_port_ms_per_step"] = round(cdt * 1e3, 2)
        out["cpu_port"] = "torch-CPU float64 autograd, %d threads" % torch.get_num_threads()
    print(json.dumps(out), flush=True)
