"""Generate tests/golden/arch_variants.npz: fp64-oracle outputs for the parts of ``noise_flow_arch`` the shipped model does
not exercise — every sdn / gain layer key and the other settings of hps.flow_permutation / hps.decomp — on seeded inputs and
seeded variables (8x8 patches, coupling width 4).  Freezes the oracle (CPU tier) and anchors the HIP path (GPU tier).
Run from the repo root:   python tools/make_golden_variants.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.nf_oracle import NoiseFlowOracle, fresh_variables  # noqa: E402

CASES = [  # (arch, flow_permutation, decomp, iso, cam)
    ("sdn|unc|gain", 1, "LU", 400, 0),
    ("sdn1|unc|gain1", 1, "LU", 800, 1),
    ("sdn2|unc|gain2|unc", 1, "LU2", 1600, 2),
    ("sdn3|unc|gain3", 1, "NONE", 100, 3),
    ("sdn4|unc|gain4|unc", 0, "LU", 3200, 4),
    ("sdn6|unc|unc|gain4", 2, "LU", 250, 2),
    ("sdn5|unc|gain4|unc", 1, "NONE", 800, 2),
]


def variables_for(arch, fp, decomp, seed, iso):
    v = fresh_variables(arch, 4, 4, seed, fp, decomp)
    rng = np.random.RandomState(1000 + seed)
    kinds = set(arch.split("|"))
    for k in list(v):
        a = np.asarray(v[k])
        if k.endswith("l_1/W") or k.endswith("l_2/W"):
            v[k] = (rng.randn(*a.shape) * 0.4).astype(np.float32)
        elif k.endswith("l_last/W"):
            v[k] = (rng.randn(*a.shape) * 0.15).astype(np.float32)
        elif k.endswith("/b") or k.endswith("l_last/logs"):
            v[k] = (rng.randn(*a.shape) * 0.1).astype(np.float32)
        elif k.endswith("/mean"):
            v[k] = (rng.randn(*a.shape) * 0.2).astype(np.float32)
        elif k.endswith("/var"):
            v[k] = (0.5 + rng.rand(*a.shape)).astype(np.float32)
        elif "rescaling_scale" in k:
            v[k] = np.float32(0.3 + 0.6 * rng.rand())
        elif "Conv2d_1x1" in k and not ("/P_" in k or "sign_S" in k):
            v[k] = (a + 0.1 * rng.randn(*a.shape)).astype(np.float32)
        elif "r_gain_param_" in k:
            v[k] = np.asarray([(-np.log(iso) + 0.3 * rng.randn()) / 1e-2], np.float32)
        elif "gain_param_" in k:
            v[k] = np.asarray([(-np.log(iso) + 0.3 * rng.randn()) / 1e-1] if kinds & {"sdn2", "sdn3", "gain2"} else
                              [0.3 * rng.randn() / 1e-5], np.float32)
        elif k in ("model/b1", "model/b2"):
            v[k] = np.asarray([rng.randn()], np.float32)
        elif k == "model/g1":
            v[k] = np.asarray([-np.log(iso) / 1e-5 if "gain1" in kinds else -np.log(iso)], np.float32)
        elif k == "model/g2":
            v[k] = np.asarray([-0.7 / 1e-5 if "gain1" in kinds else -0.7], np.float32)
        elif k == "model/sdn_gain/cam_params":
            shape = (1, 5) if "sdn6" in kinds else a.shape
            v[k] = (1.0 + 0.2 * rng.randn(*shape)).astype(np.float32)
        elif k == "model/sdn_gain/gain_params":
            v[k] = (-np.log(np.asarray([100, 400, 800, 1600, 3200.0])) * 0.8 + 0.1 * rng.randn(5)).astype(np.float32)
        elif k in ("model/sdn_gain/beta1", "model/sdn_gain/beta2"):
            v[k] = np.asarray([-1.0 + 0.3 * rng.randn()], np.float32)
        elif k == "model/sdn_gain/gain_val":
            v[k] = np.asarray([1.3], np.float32)
    return v


def main():
    out, meta = {}, []
    rng = np.random.RandomState(20240927)
    for i, (arch, fp, decomp, iso, cam) in enumerate(CASES):
        v = variables_for(arch, fp, decomp, 50 + i, iso)
        o = NoiseFlowOracle(arch, v, flow_permutation=fp, decomp=decomp)
        y = rng.rand(3, 8, 8, 4).astype(np.float32)
        x = (rng.randn(3, 8, 8, 4) * np.sqrt(0.003 * y + 2e-6)).astype(np.float32)
        eps = rng.randn(3, 8, 8, 4).astype(np.float32)
        nll, sd, z = o.nll(x, y, iso, cam)
        tag = "c%d_" % i
        for name, arr in v.items():
            out[tag + "var:" + name] = np.asarray(arr)
        out[tag + "x"], out[tag + "y"], out[tag + "eps"] = x, y, eps
        out[tag + "nll"], out[tag + "sdz"], out[tag + "z"] = nll, np.asarray(sd), z
        out[tag + "sample"] = o.sample(eps, 0.7, y, iso, cam)
        meta.append({"arch": arch, "flow_permutation": fp, "decomp": decomp, "iso": iso, "cam": cam,
                     "layer_names": [L["name"] for L in o.layers]})
    out["meta"] = np.asarray(json.dumps(meta))
    path = os.path.join(ROOT, "tests", "golden", "arch_variants.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
