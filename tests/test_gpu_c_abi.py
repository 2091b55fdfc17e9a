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


def test_host_fed_entries_equal_the_device_resident_path(shipped_variables):
    """nf_nll_host / nf_sample_host (host pointers, float64 or float32, chunked over three streams with pinned staging) give
    per-patch results BIT-identical to nf_nll / nf_sample on device tensors: multi-chunk batches (NF_HOSTFED_CHUNK-sized
    pieces + a ragged tail), both dtypes, every optional output, in-kernel Philox with a patch base that crosses chunks."""
    import torch
    from conftest import FULL_ARCH, make_inputs
    from noise_flow_amd import NoiseFlow, default_hps
    os.environ["NF_HOSTFED_CHUNK"] = "96"
    try:
        m = NoiseFlow([32, 32, 4], False, default_hps(), variables=shipped_variables)
        B = 96 * 3 + 41
        x, y = make_inputs(B, seed=9)
        xd, yd = torch.tensor(x).cuda(), torch.tensor(y).cuda()
        for cast in (np.float64, np.float32):
            nll_h, sd_h = m._loss(x.astype(cast), y.astype(cast), [0.0], [0.0], [800], [3])
            nll_d, sd_d = m._loss(xd, yd, [0.0], [0.0], [800], [3])
            assert isinstance(nll_h, np.ndarray) and np.array_equal(nll_h, nll_d.cpu().numpy())
            assert abs(sd_h - float(sd_d)) <= 1e-6 * sd_h
            z_h, obj_h = m.inverse(x.astype(cast), None, y.astype(cast), [0.0], [0.0], [800], [3])
            z_d, obj_d = m.inverse(xd, None, yd, [0.0], [0.0], [800], [3])
            assert np.array_equal(z_h, z_d.cpu().numpy()) and np.array_equal(obj_h, obj_d.cpu().numpy())
            mean_h, msd_h = m.loss(x.astype(cast), y.astype(cast), [0.0], [0.0], [800], [3])
            mean_d, msd_d = m.loss(xd, yd, [0.0], [0.0], [800], [3])
            assert abs(float(mean_h) - float(mean_d)) <= 1e-6 * abs(float(mean_d)) and abs(float(msd_h) - float(msd_d)) <= 1e-6
            eps = np.random.RandomState(2).randn(B, 32, 32, 4).astype(np.float32)
            xs_h = m.sample(y.astype(cast), 0.7, y.astype(cast), [0.0], [0.0], [800], [3], eps=eps)
            xs_d = m.sample(yd, 0.7, yd, [0.0], [0.0], [800], [3], eps=torch.tensor(eps).cuda())
            assert np.array_equal(xs_h, xs_d.cpu().numpy())
            m._draws = 1000
            p_h = m.sample(y.astype(cast), 0.7, y.astype(cast), [0.0], [0.0], [800], [3], seed=11)
            m._draws = 1000
            p_d = m.sample(yd, 0.7, yd, [0.0], [0.0], [800], [3], seed=11)
            assert np.array_equal(p_h, p_d.cpu().numpy()) and m._draws == 1000 + B
        # an empty batch and a model without signal-dependent layers (y = NULL)
        nll0, _ = m._loss(x[:0].astype(np.float64), y[:0].astype(np.float64), [0.0], [0.0], [800], [3])
        assert nll0.shape == (0,)
        from conftest import trained_like_variables
        mu = NoiseFlow([16, 16, 4], False, default_hps(arch="unc|unc"), variables=trained_like_variables("unc|unc", 4, seed=2))
        xu, _ = make_inputs(150, 16, 16, seed=3)
        hps_u = mu.hps
        hps_u.sidd_cond = "uncond"
        nu_h, _ = mu._loss(xu.astype(np.float64), None)
        nu_d, _ = mu._loss(torch.tensor(xu).cuda(), None)
        assert np.array_equal(nu_h, nu_d.cpu().numpy())
    finally:
        del os.environ["NF_HOSTFED_CHUNK"]


def test_sixteen_host_threads_feed_one_handle(shipped_variables):
    """The reference's concurrency contract (train_noise_flow.py:30-47, job_noise_flow.sh:36: 16 queue workers, every one calling
    sess.run with its own float64 minibatch of 138 patches on the shared session): nf_nll_host / nf_sample_host take 16 callers
    on one handle at once — each call in flight on a pipeline of its own — with per-patch results bit-identical to the
    single-threaded call, and the aggregate rate does not drop below the single caller's."""
    import threading
    import time
    from conftest import make_inputs
    from noise_flow_amd import NoiseFlow, default_hps
    m = NoiseFlow([32, 32, 4], False, default_hps(), variables=shipped_variables)
    B = 138
    data = [tuple(a.astype(np.float64) for a in make_inputs(B, seed=50 + i)) for i in range(16)]
    ref = [m._loss(x, y, [0.0], [0.0], [800], [3])[0] for x, y in data]
    refs = [m.sample(y, 0.7, y, [0.0], [0.0], [800], [3], seed=100 + i) for i, (x, y) in enumerate(data)]
    m._draws = 0

    def rate(nthreads, calls=160):
        errs = []

        def work(i):
            try:
                x, y = data[i]
                for _ in range(calls // nthreads):
                    nll, _ = m._loss(x, y, [0.0], [0.0], [800], [3])
                    assert np.array_equal(nll, ref[i])
            except Exception as e:  # pragma: no cover
                errs.append(e)
        th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs[0]
        return (calls // nthreads) * nthreads * B / (time.perf_counter() - t0)
    rate(16, 32)                 # every thread's pipeline exists
    one, sixteen = rate(1), rate(16)
    assert sixteen >= one, (one, sixteen)      # measured: 2.1 x (4 calls in flight, the other 12 callers waiting their turn)
    # sampling from 16 threads at once: the library's part is re-entrant (the Python model's draw counter is not: explicit seeds)
    out = [None] * 16

    def draw(i):
        x, y = data[i]
        out[i] = m._flow.lib and m.sample(y, 0.7, y, [0.0], [0.0], [800], [3], seed=100 + i)
    th = [threading.Thread(target=draw, args=(i,)) for i in range(16)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert all(o is not None and o.shape == refs[i].shape for i, o in enumerate(out))
