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


def main_wide():
    """tests/golden/train_step_width48.npz: the same for a model at a coupling width the trainer runs on its library-GEMM path
    (csrc/nf_train_gemm.h).  The model is drawn here (seeded numpy) and stored IN the fixture, so the tests need no generator."""
    from noise_flow_amd import params
    arch, width, H, W, B, iso, cam, lr = "sdn5|unc|gain4|unc", 48, 8, 6, 3, 400, 1, 1e-3
    seed = int(os.environ.get("NF_GOLDEN_SEED", "50"))     # 48, 49: an activation within 32 ulp of its ReLU kink
    rng = np.random.RandomState(seed)
    v = params.init_variables(arch, width, 4, 48)
    for k in list(v):
        a = np.asarray(v[k])
        if k.endswith("l_1/W"):
            v[k] = (rng.randn(*a.shape) * 0.4).astype(np.float32)
        elif k.endswith("l_2/W"):
            v[k] = (rng.randn(*a.shape) * 0.4 * (4.0 / width) ** 0.5).astype(np.float32)
        elif k.endswith("l_last/W"):
            v[k] = (rng.randn(*a.shape) * 0.15 * (4.0 / width) ** 0.5).astype(np.float32)
        elif k.endswith("/b") or k.endswith("l_last/logs"):
            v[k] = (rng.randn(*a.shape) * 0.1).astype(np.float32)
        elif k.endswith("/mean"):
            v[k] = (rng.randn(*a.shape) * 0.2).astype(np.float32)
        elif k.endswith("/var"):
            v[k] = (0.5 + rng.rand(*a.shape)).astype(np.float32)
        elif "rescaling_scale" in k:
            v[k] = np.float32(0.3 + 0.6 * rng.rand())
        elif "log_S" in k or "L_vec" in k or "U_vec" in k:
            v[k] = (a + rng.randn(*a.shape).astype(np.float32) * 0.1).astype(np.float32)
    y = rng.rand(B, H, W, 4).astype(np.float32)
    x = (rng.randn(B, H, W, 4) * np.sqrt(0.003696 * y + 2e-6)).astype(np.float32)
    o = GradOracle(arch, v)
    loss, sd_z, grads, new_running = o.loss_and_grads(x, y, iso, cam)
    assert not o.kinks, "an activation sits on its ReLU kink: pick another seed"
    after = adam_step(v, grads, {}, lr)
    out = {"arch": np.asarray(arch), "width": np.asarray(width), "x": x, "y": y, "iso": np.asarray(iso), "cam": np.asarray(cam),
           "lr": np.asarray(lr), "loss": np.asarray(loss), "sd_z": np.asarray(sd_z)}
    for k, a in v.items():
        out["var/" + k] = np.asarray(a, np.float32)
    for k, g in grads.items():
        out["grad/" + k] = np.asarray(g, np.float32)
        out["abs/" + k] = np.asarray(o.grad_abs_terms[k], np.float32)     # round-off allowance of each entry (conftest.py)
    for k, a in new_running.items():
        out["bn/" + k] = np.asarray(a, np.float32)
    for k in grads:
        if is_trainable(k):
            out["adam/" + k] = np.asarray(after[k], np.float32)
    path = os.path.join(ROOT, "tests", "golden", "train_step_width48.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(grads), "gradient tensors;", sum(int(np.asarray(g).size) for g in grads.values()), "entries")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "wide":
        main_wide()
    else:
        main()
