"""GPU tier: the C ABI driven by a plain C program (no Python/torch in that process) gives the
same numbers as the Python surface, on identical Philox-keyed inputs."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import FULL_ARCH, ROOT

pytestmark = pytest.mark.gpu


def test_standalone_c_client(tmp_path, shipped_variables):
    from noise_flow_amd import NoiseFlow, default_hps, params
    from noise_flow_amd.patches import synth_patches
    layers, descs, flat = params.pack(FULL_ARCH, shipped_variables, 4)
    model = tmp_path / "model.bin"
    with open(model, "wb") as f:
        f.write(struct.pack("<i", len(layers)))
        for d in descs:
            f.write(struct.pack("<iiq", d.type, d.width, d.param_offset))
        f.write(struct.pack("<q", flat.size))
        f.write(flat.tobytes())
    exe = tmp_path / "c_abi_demo"
    csrc = os.path.join(ROOT, "noise_flow_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c99", "-O2", os.path.join(ROOT, "examples", "c_abi_demo.c"),
                           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           "-L" + csrc, "-lnoiseflow_hip", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    B = 37
    out = subprocess.check_output([str(exe), str(model), str(B)], text=True).strip().splitlines()
    vals = np.array([[float(t) for t in ln.split()] for ln in out[:B]])
    sums = [float(t) for t in out[B].split()[1:]]
    checksum = float(out[B + 1].split()[1])
    assert out[B + 2].split()[1:] == ["-3", "NF_ECOND"]

    m = NoiseFlow([32, 32, 4], False, default_hps(), variables=shipped_variables)
    x, y = synth_patches(7, 0, B)
    nll, _ = m._loss(x, y, [0], [0], [100], [2])
    np.testing.assert_allclose(vals[:, 0], nll.cpu().numpy(), rtol=1e-6)      # same kernel, same inputs
    assert sums[2] == B and abs(sums[0] - vals[:, 0].astype(np.float64).sum()) < 1e-3
    xs = m.sample(y, 0.6, y, [0], [0], [100], [2], seed=99)
    assert abs(checksum - float(xs.double().sum())) <= 1e-6 * float(xs.abs().double().sum())
