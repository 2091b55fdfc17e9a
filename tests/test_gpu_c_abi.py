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


def test_standalone_c_training_client(tmp_path):
    """The trainer driven by a plain C program: same losses and (bit for bit) the same parameters as
    the Python Trainer on the same Philox-keyed minibatches."""
    from conftest import trained_like_variables
    from noise_flow_amd import default_hps, params
    from noise_flow_amd.patches import synth_patches
    from noise_flow_amd.train import Trainer
    arch = "sdn5|unc|unc|gain4|unc"
    v = trained_like_variables(arch, 4, seed=13)
    layers, descs, flat = params.pack(arch, v, 4)
    model = tmp_path / "model.bin"
    with open(model, "wb") as f:
        f.write(struct.pack("<i", len(layers)))
        for d in descs:
            f.write(struct.pack("<iiq", d.type, d.width, d.param_offset))
        f.write(struct.pack("<q", flat.size))
        f.write(flat.tobytes())
    exe = tmp_path / "c_abi_train_demo"
    csrc = os.path.join(ROOT, "noise_flow_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c99", "-O2", os.path.join(ROOT, "examples", "c_abi_train_demo.c"),
                           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           "-L" + csrc, "-lnoiseflow_hip", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    B, K, lr = 24, 4, 1e-3
    outp = tmp_path / "trained.bin"
    out = subprocess.check_output([str(exe), str(model), str(B), str(K), repr(lr), str(outp)], text=True).strip().splitlines()
    losses = [(float(l.split()[3]), float(l.split()[5])) for l in out[:K]]
    assert out[K].split()[1:] == ["-1", "NF_EINVAL"]
    trained = np.fromfile(str(outp), np.float32)

    tr = Trainer([32, 32, 4], default_hps(arch=arch), variables=v, max_batch=B)
    for k in range(K):
        x, y = synth_patches(11, k * B, B, nlf=(0.003696, 0.000002))
        loss, sd = tr.step(x, y, [0.0], [0.0], [800.0], [2.0], lr=lr)
        assert abs(loss - losses[k][0]) <= 1e-6 * abs(loss) and abs(sd - losses[k][1]) <= 1e-6 * sd
    assert np.array_equal(tr.raw_params(), trained)
    assert losses[-1][0] < losses[0][0]
