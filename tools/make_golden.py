"""Generate the committed golden vectors under tests/golden/.

The reference cannot run anywhere in this project (TensorFlow 1.12 is absent), so
the vectors come from the fp64 numpy oracle (oracle/nf_oracle.py) applied to the
reference's SHIPPED checkpoint on seeded inputs: they freeze the oracle (CPU
tier) and anchor the HIP path (GPU tier).  Run from the repo root:
    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noise_flow_amd.ckpt import load_checkpoint          # noqa: E402
from oracle.nf_oracle import NoiseFlowOracle, layer_names  # noqa: E402

ARCH = "sdn5|unc|unc|unc|unc|gain4|unc|unc|unc|unc"


def main():
    v = load_checkpoint(os.path.join(ROOT, "models", "NoiseFlow", "ckpt", "model.ckpt.best"))
    o = NoiseFlowOracle(ARCH, v)
    rng = np.random.RandomState(20190827)
    B = 4
    y = rng.rand(B, 32, 32, 4).astype(np.float32)
    out = {"arch": np.asarray(ARCH), "layer_names": np.asarray(layer_names(ARCH)), "y": y}
    for iso, cam, b1 in ((100, 2, 0.000479), (800, 2, 0.003696), (3200, 1, 0.02)):
        x = (rng.randn(B, 32, 32, 4) * np.sqrt(b1 * y + 2e-6)).astype(np.float32)
        z, obj, per = o.inverse(x, y, iso, cam, return_layers=True)
        nll, sd, _ = o.nll(x, y, iso, cam)
        tag = "iso%d_cam%d" % (iso, cam)
        out["x_" + tag] = x
        out["nll_" + tag] = nll
        out["sdz_" + tag] = np.asarray(sd)
        out["logdet_" + tag] = obj
        out["layer_ld_" + tag] = np.stack([ld for _, _, ld in per])
        out["z_" + tag] = z.astype(np.float32)
    eps = rng.randn(B, 32, 32, 4).astype(np.float32)
    out["eps"] = eps
    for temp in (1.0, 0.6):
        out["sample_t%.1f_iso100_cam2" % temp] = o.sample(eps, temp, y, 100, 2).astype(np.float32)
    path = os.path.join(ROOT, "tests", "golden", "full_arch_shipped.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
