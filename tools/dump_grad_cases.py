"""Dump the trainer's loss / gradients of some draws of test_random_model_training_gradients for offline analysis:
    python tools/dump_grad_cases.py out.npz seed [seed ...]"""
import os
import sys

R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, R)
import numpy as np  # noqa: E402
import test_gpu_random_sweep as S  # noqa: E402


def case(seed):
    from noise_flow_amd import params
    from conftest import make_inputs, trained_like_variables
    arch, width, (H, W), fp, decomp, iso, cam, B = S._draw_case(5000 + seed)
    if "unc" not in arch.split("|"):
        arch = arch + "|unc"
    H, W = min(H, 32), min(W, 32)
    B = max(B, 2) + seed % 4
    if H * W * B < 32:
        H, W = H + 4, W + 4
    rng = np.random.RandomState(seed)
    v = params.init_variables(arch, width, 4, seed, fp, decomp)
    base = trained_like_variables(arch, width, seed=seed)
    for k in v:
        if k in base:
            v[k] = base[k]
    v = S._condition(v, arch, width, iso, rng)
    x, y = make_inputs(B, H, W, seed=seed)
    return arch, width, (H, W), fp, decomp, iso, cam, B, v, x, y


if __name__ == "__main__":
    from noise_flow_amd import default_hps
    from noise_flow_amd.train import Trainer
    out = {}
    for seed in (int(a) for a in sys.argv[2:]):
        arch, width, (H, W), fp, decomp, iso, cam, B, v, x, y = case(seed)
        tr = Trainer([H, W, 4], default_hps(arch=arch, width=width, flow_permutation=fp, decomp=decomp), variables=v, optim="adam", max_batch=16)
        grads, loss = tr.forward_backward(x, y, [0.0], [0.0], [iso], [cam])
        got = tr.raw_to_variables(grads.cpu().numpy())
        out["loss_%d" % seed] = loss.cpu().numpy()
        for k, a in got.items():
            out["g_%d_%s" % (seed, k.replace("/", "!"))] = np.asarray(a)
        # the same step on the OTHER kernel paths (layer kernels instead of the tiled / matrix-core stages): a different fp32
        # summation order of the same arithmetic
        os.environ["NF_TRAIN_TILED"] = "0"
        os.environ["NF_TRAIN_WIDE_MFMA"] = "0"
        tr2 = Trainer([H, W, 4], default_hps(arch=arch, width=width, flow_permutation=fp, decomp=decomp), variables=v, optim="adam", max_batch=16)
        del os.environ["NF_TRAIN_TILED"], os.environ["NF_TRAIN_WIDE_MFMA"]
        ga, la = tr2.forward_backward(x, y, [0.0], [0.0], [iso], [cam])
        for k, a in tr2.raw_to_variables(ga.cpu().numpy()).items():
            out["alt_%d_%s" % (seed, k.replace("/", "!"))] = np.asarray(a)
        out["altloss_%d" % seed] = la.cpu().numpy()
        # a second evaluation: run-to-run reproducibility of the trainer on this case
        grads2, loss2 = tr.forward_backward(x, y, [0.0], [0.0], [iso], [cam])
        out["same_%d" % seed] = np.asarray([float((grads2 == grads).all()), float((loss2 == loss).all())])
    np.savez_compressed(sys.argv[1], **out)
    print("dumped", sys.argv[2:])
