"""Patches beyond 64x64 (overlapping tiles, csrc/nf_device.h) against the fp64 oracle on the whole image.

    python tools/check_large_patches.py H W [B] [arch] [width] [fp32|fp16]     # NF_KERNEL=valu: the scalar-weight kernel

Prints one JSON line with the worst relative errors and exits 1 when one is above the parity tolerances of
tests/test_gpu_parity.py (NLL 1e-5 relative, tensors 1e-5 of their scale).  Test infrastructure (imports oracle/)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))


def check(H, W, B=2, arch=None, seed=0, variables=None, strict=False, width=4, cnn_dtype="fp32"):
    import torch
    from conftest import FULL_ARCH, make_inputs, trained_like_variables
    import ctypes as C
    from noise_flow_amd import NoiseFlow, default_hps, _lib, params
    from oracle import philox
    from oracle.nf_oracle import NoiseFlowOracle

    arch = arch or FULL_ARCH
    v = variables if variables is not None else trained_like_variables(arch, width, seed=H * 1000 + W + seed)
    x, y = make_inputs(B, H, W, seed=seed + 7)
    m = NoiseFlow([H, W, 4], False, default_hps(arch=arch, width=width), variables=v, cnn_dtype=cnn_dtype)
    # fp16-CNN mode: the oracle emulates the library's rounding points (tests/test_gpu_parity.py: NLL 1e-4, tensors 2e-3)
    o = NoiseFlowOracle(arch, v, "loss_first", cnn_dtype=cnn_dtype)
    out = {"H": H, "W": W, "B": B, "arch": arch, "kernel_path": int(m._flow.lib.nf_kernel_path(m._flow.ptr, 0))}
    _, descs, flat = params.pack(arch, v, width)
    out["width"], out["cnn_dtype"] = width, cnn_dtype
    out["segments"] = int(m._flow.lib.nf_tile_segments(C.byref(_lib.nf_config(H, W, 4, len(descs), -1, _lib.NF_CFG_FP16_CNN if cnn_dtype == "fp16" else 0)), descs,
                                                       flat.ctypes.data_as(C.POINTER(C.c_float)), flat.size, 0, None, 0))

    def rel(a, ref):
        ref = np.asarray(ref, np.float64)
        return float(np.abs(np.asarray(a, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))

    ref_nll, ref_sd, ref_z = o.nll(x, y, 100, 2)
    # host-fed call (numpy in / numpy out) and the device-resident call must be the same numbers
    nll, sd = m._loss(x, y, [0.0], [0.0], [100], [2])
    xt, yt = torch.as_tensor(x).cuda(), torch.as_tensor(y).cuda()
    nll_d, sd_d = m._loss(xt, yt, [0.0], [0.0], [100], [2])
    out["host_fed_equals_resident"] = bool(np.array_equal(nll, nll_d.cpu().numpy()) and np.float32(sd) == np.float32(float(sd_d)))
    out["nll"] = float(np.abs(nll / ref_nll - 1.0).max())
    out["sd"] = abs(sd / ref_sd - 1.0)
    mean_nll, mean_sd = m.loss(xt, yt, [0.0], [0.0], [100], [2])
    out["mean_nll"] = abs(float(mean_nll) / float(np.mean(ref_nll)) - 1.0)
    z, obj = m.inverse(x, None, y, [0.0], [0.0], [100], [2])
    out["z"] = rel(z, ref_z)
    out["logdet"] = float(np.abs(np.asarray(obj, np.float64) / o.inverse(x, y, 100, 2)[1] - 1.0).max())
    x2 = m.forward(z, None, y, [0.0], [0.0], [100], [2])
    out["round_trip"] = rel(x2, x)
    eps = np.random.RandomState(seed + 4).randn(B, H, W, 4).astype(np.float32)
    xs = m.sample(y, 0.8, y, [0.0], [0.0], [100], [2], eps=eps)
    out["sample"] = rel(xs, o.sample(eps, 0.8, y, 100, 2))
    # the in-kernel Philox draw is keyed by (patch, pixel of the IMAGE), whatever tile evaluates the pixel
    xp = m.sample(y, 0.6, y, [0.0], [0.0], [100], [2], seed=99)
    out["philox_sample"] = rel(xp, o.sample(philox.sample_eps(99, 0, B, H, W), 0.6, y, 100, 2))
    # The NLL, sd_z and log-det are held to the parity tolerances as they are.  Tensors of a randomly perturbed (and, in the
    # deep-stack case, 16 couplings deep) model can be ill-conditioned in fp32 (exp(+-log-scale) amplifies round-off), tiles
    # or no tiles: their yardstick is what the ORACLE ITSELF loses when it runs the whole image in float32 instead of
    # float64 on the same inputs.
    o32 = NoiseFlowOracle(arch, v, "loss_first", dtype=np.float32, cnn_dtype=cnn_dtype)
    cond = {"z": rel(o32.inverse(x, y, 100, 2)[0], ref_z),
            "sample": rel(o32.sample(eps, 0.8, y, 100, 2), o.sample(eps, 0.8, y, 100, 2)),
            "round_trip": rel(o32.forward(o32.inverse(x, y, 100, 2)[0], y, 100, 2), x)}
    eps_p = philox.sample_eps(99, 0, B, H, W)
    cond["philox_sample"] = rel(o32.sample(eps_p, 0.6, y, 100, 2), o.sample(eps_p, 0.6, y, 100, 2))
    out["fp32_oracle_deviation"] = cond
    half = cnn_dtype == "fp16"
    tol = {"nll": 1e-4, "sd": 1e-4, "mean_nll": 1e-4, "logdet": 1e-4} if half else {"nll": 1e-5, "sd": 1e-5, "mean_nll": 1e-5, "logdet": 1e-5}
    for k, base in (("z", 2e-3), ("round_trip", 4e-3), ("sample", 2e-3), ("philox_sample", 2e-3)) if half else \
            (("z", 1e-5), ("round_trip", 1e-5), ("sample", 1e-5), ("philox_sample", 2e-5)):
        tol[k] = base if strict else max(base, 4.0 * cond[k])
    out["ok"] = bool(out["host_fed_equals_resident"] and all(out[k] <= t for k, t in tol.items() if k in out))
    return out


if __name__ == "__main__":
    H, W = int(sys.argv[1]), int(sys.argv[2])
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    arch = (sys.argv[4] if len(sys.argv) > 4 else None) or None
    r = check(H, W, B, arch, width=int(sys.argv[5]) if len(sys.argv) > 5 else 4, cnn_dtype=sys.argv[6] if len(sys.argv) > 6 else "fp32")
    print(json.dumps(r))
    sys.exit(0 if r["ok"] else 1)
