"""The two driver scripts executed end to end, as subprocesses:

* ``sample_noise_flow_amd.py`` — the data-free core of the reference's ``sample_noise_flow.py:27-101`` (wrapper ->
  sample_noise_nf per patch -> crop 1 px -> clip(clean + noise) -> unpack_raw -> marginal KL).  Its output is held to the
  oracle evaluated on the SAME Philox epsilon, both with the trained model's semantics and in the literal-reference
  mode (``--compat reference``: sampling-graph-first binding + batch-statistics BN, quirks Q1/Q2);
* ``train_noise_flow_amd.py`` — the epoch loop of ``train_noise_flow.py:240-530`` on synthetic patches: log files in the
  reference's TSV format, ``hps.txt``, TF-bundle checkpoints that reload to the logged test NLL.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import FULL_ARCH, ROOT, SHIPPED_DIR

pytestmark = pytest.mark.gpu


def _run(args, timeout=600):
    out = subprocess.run([sys.executable] + args, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    return out.stdout.decode()


@pytest.mark.parametrize("compat", [None, "reference"])
def test_demo_sampler_script_matches_oracle(tmp_path, shipped_variables, compat):
    from noise_flow_amd.patches import unpack_raw
    from oracle import philox
    from oracle.nf_oracle import NoiseFlowOracle
    n, iso, cam, temp, seed = 8, 800.0, 2.0, 0.6, 4321
    out = str(tmp_path / "samples.npz")
    args = [os.path.join(ROOT, "sample_noise_flow_amd.py"), "--n", str(n), "--iso", str(iso), "--cam", str(cam), "--temp", str(temp),
            "--seed", str(seed), "--out", out]
    if compat:
        args += ["--compat", compat]
    stdout = _run(args)
    d = np.load(out)
    assert d["clean"].shape == (n, 32, 32, 4) and d["noise_syn"].shape == (n, 32, 32, 4) and d["noisy_syn"].shape == (n, 60, 60)
    assert "Mean KL divergence = " in stdout and abs(float(stdout.split("=")[-1]) - float(d["kld"].mean())) < 1e-9
    # the script's inputs are reproducible (np.random.seed(12345), sample_noise_flow.py:58)
    np.random.seed(12345)
    clean = np.random.rand(n, 32, 32, 4).astype(np.float32)
    np.testing.assert_array_equal(d["clean"], clean)
    # oracle on the same epsilon: patch p of the run is Philox patch index p (batch size 1 per call, running counter)
    eps = philox.sample_eps(seed, 0, n)
    o = NoiseFlowOracle(FULL_ARCH, shipped_variables, "sample_first" if compat else "loss_first")
    for p in range(n):
        ref = o.sample(eps[p:p + 1], temp, clean[p:p + 1], iso, cam, training=bool(compat))[0]
        scale = np.abs(ref).max()
        assert np.abs(d["noise_syn"][p] - ref).max() <= 5e-5 * scale      # in-kernel Box-Muller: ~1e-6 absolute on eps
        want = unpack_raw(np.clip(clean[p, 1:-1, 1:-1, :] + ref[1:-1, 1:-1, :], 0.0, 1.0).astype(np.float32))
        assert np.abs(d["noisy_syn"][p] - want).max() <= 5e-5 * max(scale, 1e-3)
    # the trained binding reproduces the camera noise far better than the literal wrapper's reversed binding (quirk Q1)
    assert np.isfinite(d["kld"]).all()
    if not compat:
        assert float(d["kld"].mean()) < 0.5


def test_training_script_writes_reference_formats_and_reloads(tmp_path):
    from noise_flow_amd import NoiseFlow, default_hps, patches
    from noise_flow_amd.ckpt import load_checkpoint
    from noise_flow_amd.harness import S6_NLF
    logdir = str(tmp_path / "run")
    _run([os.path.join(ROOT, "train_noise_flow_amd.py"), "--logdir", logdir, "--epochs", "2", "--n_train", "276", "--n_test", "138",
          "--n_batch_train", "138", "--n_batch_test", "138", "--epochs_full_valid", "1", "--lr", "1e-3"])
    for f in ("train.txt", "test.txt", "hps.txt"):
        assert os.path.exists(os.path.join(logdir, f)), f
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(logdir, "test.txt"))]
    assert rows[0][0] == "epoch" and "NLL" in rows[0] and len(rows) >= 3              # header + one row per evaluated epoch
    col = rows[0].index("NLL")
    nlls = [float(r[col]) for r in rows[1:]]
    assert all(np.isfinite(nlls)) and nlls[-1] < nlls[0]                               # it learns
    # the checkpoint written after the last epoch reloads (TF-bundle reader) and evaluates to the logged test NLL
    ck = os.path.join(logdir, "ckpt", "model.ckpt-2")
    assert os.path.exists(ck + ".index") and os.path.exists(os.path.join(logdir, "ckpt", "model.ckpt.best.index"))
    v = load_checkpoint(ck)
    m = NoiseFlow([32, 32, 4], False, default_hps(), variables=v)
    nlf = S6_NLF[800]
    x, y = patches.synth_patches(0, 276, 138, nlf=nlf)
    nll, _ = m.loss(x, y, [nlf[0]], [nlf[1]], [800.0], [2.0])
    assert abs(float(nll) - nlls[-1]) <= 1e-4 * abs(nlls[-1])


def test_training_script_consumes_the_sampler_queues(tmp_path):
    """`--pipeline queues`: image tuples -> PatchSampler -> MiniBatchSampler queues (the reference's host pipeline) feed float64
    minibatch dicts to the trainer; sample.txt carries the four KLD columns of the reference's recipe."""
    logdir = str(tmp_path / "runq")
    _run([os.path.join(ROOT, "train_noise_flow_amd.py"), "--logdir", logdir, "--epochs", "2", "--n_train", "128", "--n_test", "64",
          "--n_batch_train", "32", "--n_batch_test", "32", "--epochs_full_valid", "1", "--lr", "1e-3", "--pipeline", "queues"])
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(logdir, "train.txt"))]
    assert len(rows) == 3 and all(np.isfinite(float(r[rows[0].index("NLL")])) for r in rows[1:])
    srows = [l.rstrip("\n").split("\t") for l in open(os.path.join(logdir, "sample.txt"))]
    assert srows[0][-4:] == ["KLD_G", "KLD_NLF", "KLD_NF", "KLD_R"] and len(srows) == 3
    g, nlf, nf, r = (float(v) for v in srows[-1][-4:])
    assert r == 0.0 and g > 0 and nlf > 0 and nf > 0


def test_wrapper_default_is_the_mode_that_reproduces_the_camera_noise():
    """Why `NoiseFlowWrapper(path)` defaults to the trained model's semantics and keeps upstream's literal graph behind
    `compat='reference'`.  Pinned on the shipped checkpoint against
    heteroscedastic Gaussian noise with the S6 camera NLF (cam_iso_nlf.txt:8-12): with the binding the model was TRAINED
    with, samples at temperature 1.0 carry 1.1 - 1.3 x the NLF's standard deviation — the "too-high noise variance" that
    sample_noise_flow.py:36-39 tempers with 0.6 — and a marginal KL of a few 1e-2; with the sampling-graph-only binding
    (`binding='sample_first'`, what TF-1.12 template semantics would give the upstream wrapper) they carry 3 - 40 x, at either
    temperature.  Batch-statistics BN alone (quirk Q2) moves the ratio by < 3 %.  (tools/wrapper_modes.py prints the table.)"""
    from noise_flow_amd import NoiseFlowWrapper, metrics
    rng = np.random.RandomState(0)
    y = rng.rand(64, 32, 32, 4).astype(np.float32)
    edges = metrics.noise_bin_edges()
    for iso, b1, b2 in ((100.0, 0.000479, 0.000002), (800.0, 0.003696, 0.00001)):
        sd = np.sqrt(b1 * y + b2)
        real = (rng.randn(*y.shape) * sd).astype(np.float32)

        def run(temp, **kw):
            x = np.asarray(NoiseFlowWrapper(SHIPPED_DIR, sampling_temperature=temp, seed=1, **kw).sample_noise_nf(y, 0.0, 0.0, iso, 2.0))
            return float(np.sqrt(np.mean((x / sd) ** 2))), metrics.kl_div_3_data(real, x, bin_edges=edges)[0]

        r_def, kl_def = run(1.0)
        r_bn, kl_bn = run(1.0, bn_mode="batch")
        r_ref1, kl_ref1 = run(1.0, compat="reference")
        r_ref6, kl_ref6 = run(0.6, compat="reference")
        r_def6, _ = run(0.6)
        assert 1.0 < r_def < 1.4 and kl_def < 0.06, (iso, r_def, kl_def)            # a little too wide: hence upstream's 0.6
        assert 0.55 < r_def6 < 0.85, (iso, r_def6)
        assert abs(r_bn - r_def) < 0.03 * r_def and kl_bn < 0.06, (iso, r_bn, kl_bn)  # Q2 alone changes little
        assert r_ref1 > 2.5 and r_ref6 > 2.5, (iso, r_ref1, r_ref6)                  # Q1: the trained filters on the wrong couplings
        assert kl_ref1 > 5 * kl_def, (iso, kl_ref1, kl_def)
