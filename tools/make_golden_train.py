"""Generate tests/golden/train_step_shipped.npz: one training step of the reference's shipped model
on seeded inputs, from the fp64 autograd oracle (oracle/nf_grad_oracle.py) — loss, sd_z, the gradient of
every trainable variable, the EMA-updated BN statistics, and the variables after one Adam step.
TensorFlow 1.12 cannot run here, so these vectors freeze the ORACLE (CPU tier) and anchor the HIP
trainer (GPU tier).  Run from the repo root:  python tools/make_golden_train.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noise_flow_amd.ckpt import load_checkpoint                      # noqa: E402
from oracle.nf_grad_oracle import GradOracle, adam_step, is_trainable  # noqa: E402

ARCH = "sdn5|unc|unc|unc|unc|gain4|unc|unc|unc|unc"


def main():
    v = load_checkpoint(os.path.join(ROOT, "models", "NoiseFlow", "ckpt", "model.ckpt.best"))
    rng = np.random.RandomState(20190828)
    B, iso, cam, lr = 4, 800, 2, 1e-4
    y = rng.rand(B, 32, 32, 4).astype(np.float32)
    x = (rng.randn(B, 32, 32, 4) * np.sqrt(0.003696 * y + 2e-6)).astype(np.float32)
    loss, sd_z, grads, new_running = GradOracle(ARCH, v).loss_and_grads(x, y, iso, cam)
    after = adam_step(v, grads, {}, lr)
    out = {"arch": np.asarray(ARCH), "x": x, "y": y, "iso": np.asarray(iso), "cam": np.asarray(cam), "lr": np.asarray(lr),
           "loss": np.asarray(loss), "sd_z": np.asarray(sd_z)}
    for k, g in grads.items():
        out["grad/" + k] = np.asarray(g, np.float32)
    for k, a in new_running.items():
        out["bn/" + k] = np.asarray(a, np.float32)
    for k in grads:
        if is_trainable(k):
            out["adam/" + k] = np.asarray(after[k], np.float32)
    path = os.path.join(ROOT, "tests", "golden", "train_step_shipped.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(grads), "gradient tensors")


if __name__ == "__main__":
    main()
