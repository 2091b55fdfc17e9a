"""CPU tier: pin the oracle (the reference ships no tests / golden vectors and
cannot run here — SURVEY §8c lists what pins the restatement instead)."""
import os

import numpy as np
import pytest

from conftest import FULL_ARCH, GOLDEN_DIR, make_inputs, trained_like_variables
from oracle import nf_oracle as O


def test_fill_triangular_layout():
    v = np.arange(6.0)
    assert np.array_equal(O.fill_triangular(v, True), [[0, 1, 2], [0, 4, 5], [0, 0, 3]])
    assert np.array_equal(O.fill_triangular(v, False), [[3, 0, 0], [5, 4, 0], [2, 1, 0]])
    u = O.vec2stricttri(v + 1, True)
    l = O.vec2stricttri(v + 1, False)
    assert np.all(np.tril(u) == 0) and np.all(np.triu(l) == 0)
    assert np.array_equal(O.stricttri2vec(u, True), v + 1) and np.array_equal(O.stricttri2vec(l, False), v + 1)


def test_lu_param_reconstructs_initial_matrix_and_slogdet(shipped_variables):
    import scipy.linalg as sla
    q = sla.qr(np.random.RandomState(1).randn(4, 4))[0]
    p = O.lu_init_from_matrix(q)
    A, A_inv, lad = O.matrix_param_lu(p["P"], p["sign_S"], p["log_S"], p["L_vec"], p["U_vec"])
    np.testing.assert_allclose(A, q, atol=1e-6)
    np.testing.assert_allclose(A @ A_inv, np.eye(4), atol=1e-6)
    assert abs(lad) < 1e-6                                   # orthogonal -> sum log_S = 0
    m = O.NoiseFlowOracle(FULL_ARCH, shipped_variables)
    for L in m.layers:
        if L["type"] == "conv1x1":
            assert abs(np.linalg.slogdet(L["A"])[1] - L["log_abs_det"]) < 1e-12
            np.testing.assert_allclose(L["A"] @ L["A_inv"], np.eye(4), atol=1e-12)


def test_fresh_init_known_answers():
    """Zero-init last conv + orthogonal A: NLL = 1/2 HWC log 2pi + 1/2 ||x||^2 exactly."""
    x = np.random.RandomState(0).randn(5, 32, 32, 4)
    m = O.NoiseFlowOracle("unc|unc|unc", O.fresh_variables("unc|unc|unc", seed=3))
    nll, sd, z = m.nll(x)
    np.testing.assert_allclose(nll, 0.5 * 4096 * np.log(2 * np.pi) + 0.5 * (x ** 2).sum((1, 2, 3)), rtol=1e-6)
    assert abs(0.5 * 4096 * np.log(2 * np.pi) - 3763.97) < 0.01
    # fresh sdn5: scale = sqrt(exp(-5e) * y / (exp(-5e) * iso) + 1) = sqrt(y/iso + 1)
    y = np.random.RandomState(1).rand(2, 8, 8, 4)
    v = O.fresh_variables("sdn5|unc")
    L = O.bind_variables("sdn5|unc", v)[0]
    np.testing.assert_allclose(O.sdn_ex5_scale(y, L["p"], 400.0, 3.0), np.sqrt(y / 400.0 + 1.0), rtol=1e-6)


def test_unknown_iso_and_cam(shipped_variables):
    L = O.bind_variables(FULL_ARCH, shipped_variables)[0]
    b1, b2, g = O.sdn_ex5_scalars(L["p"], 250.0, 2.0)
    assert abs(g - 250.0) < 1e-9                             # g = exp(0) * iso  (cond_utils.py:227-230)
    with pytest.raises(IndexError):
        O.sdn_ex5_scalars(L["p"], 100.0, 9.0)


def test_round_trip_and_additivity(oracle_full):
    x, y = make_inputs(3, seed=5)
    z, obj, per = oracle_full.inverse(x, y, 800, 2, return_layers=True)
    np.testing.assert_allclose(sum(ld for _, _, ld in per), obj, rtol=1e-12)
    assert [n for n, _, _ in per] == O.layer_names(FULL_ARCH)
    back = oracle_full.forward(z, y, 800, 2)
    assert np.abs(back - x).max() <= 1e-12 * np.abs(x).max()


def test_logdet_equals_jacobian_on_toy_patch():
    """log|det J| of the whole chain vs a finite-difference Jacobian (4x4x4 patch)."""
    arch = "sdn5|unc|gain4|unc"
    v = trained_like_variables(arch, 4, seed=2)
    m = O.NoiseFlowOracle(arch, v)
    rng = np.random.RandomState(0)
    y = rng.rand(1, 4, 4, 4)
    x = rng.randn(1, 4, 4, 4) * 0.1
    _, obj = m.inverse(x, y, 100, 2)
    n = x.size
    J = np.zeros((n, n))
    h = 1e-6
    for i in range(n):
        d = np.zeros(n)
        d[i] = h
        zp, _ = m.inverse(x + d.reshape(x.shape), y, 100, 2)
        zm, _ = m.inverse(x - d.reshape(x.shape), y, 100, 2)
        J[:, i] = (zp - zm).reshape(-1) / (2 * h)
    assert abs(np.linalg.slogdet(J)[1] - obj[0]) < 1e-5


def test_shipped_model_plausibility_band(oracle_full):
    """the trained model is within 0.05 nat/dim of the generating
    density and whitens to sd_z in [0.8, 1.0]; the reversed binding does not."""
    for iso, b1 in ((100, 0.000479), (800, 0.003696), (3200, 0.01993)):
        x, y = make_inputs(16, seed=iso, b1=b1)
        nll, sd, _ = oracle_full.nll(x, y, iso, 2)
        exact = O.nll_sdn(x, y, b1, 2e-6)
        assert abs(nll.mean() - exact.mean()) / 4096 < 0.05
        assert 0.8 < sd < 1.0


def test_wrong_binding_is_drastically_different(shipped_variables):
    x, y = make_inputs(8, seed=1)
    bad = O.NoiseFlowOracle(FULL_ARCH, shipped_variables, binding="sample_first")
    nll, sd, _ = bad.nll(x, y, 100, 2)
    assert sd > 2.0 and nll.mean() / 4096 > 0.0


def test_closed_form_baselines():
    x = np.random.RandomState(0).randn(3, 8, 8, 4) * 0.3
    y = np.full_like(x, 0.5)
    g = O.nll_gauss(x, 0.3)
    s = O.nll_sdn(x, y, 0.0, 0.09)
    np.testing.assert_allclose(g, s, rtol=1e-12)            # same density, two formulas


def test_fp32_flavour_tracks_fp64(shipped_variables, oracle_full):
    x, y = make_inputs(4, seed=2)
    o32 = O.NoiseFlowOracle(FULL_ARCH, shipped_variables, dtype=np.float32)
    a = o32.nll(x, y, 100, 2)[0]
    b = oracle_full.nll(x, y, 100, 2)[0]
    assert a.dtype == np.float32
    np.testing.assert_allclose(a, b, rtol=2e-6)


def test_torch_cpu_baseline_port_matches_oracle(shipped_variables, oracle_full):
    from oracle.nf_cpu_torch import TorchCpuFlow
    c = TorchCpuFlow(FULL_ARCH, shipped_variables)
    x, y = make_inputs(6, seed=3)
    nll, sd = c.nll(x, y, 100.0, 2.0)
    ref, rsd, _ = oracle_full.nll(x, y, 100, 2)
    np.testing.assert_allclose(nll.numpy(), ref, rtol=2e-6)
    assert abs(sd - rsd) < 1e-5
    eps = np.random.RandomState(1).randn(6, 32, 32, 4).astype(np.float32)
    xs = c.sample(eps, 0.6, y, 100.0, 2.0).numpy()
    r = oracle_full.sample(eps, 0.6, y, 100, 2)
    assert np.abs(xs - r).max() <= 1e-5 * np.abs(r).max()


def test_golden_vectors_freeze_the_oracle(oracle_full):
    g = np.load(os.path.join(GOLDEN_DIR, "full_arch_shipped.npz"))
    assert str(g["arch"]) == FULL_ARCH and list(g["layer_names"]) == O.layer_names(FULL_ARCH)
    y = g["y"]
    for iso, cam in ((100, 2), (800, 2), (3200, 1)):
        tag = "iso%d_cam%d" % (iso, cam)
        nll, sd, z = oracle_full.nll(g["x_" + tag], y, iso, cam)
        np.testing.assert_allclose(nll, g["nll_" + tag], rtol=1e-12)
        np.testing.assert_allclose(sd, g["sdz_" + tag], rtol=1e-12)
        np.testing.assert_allclose(z, g["z_" + tag], rtol=0, atol=1e-6 * np.abs(z).max())
        _, obj, per = oracle_full.inverse(g["x_" + tag], y, iso, cam, return_layers=True)
        np.testing.assert_allclose(np.stack([ld for _, _, ld in per]), g["layer_ld_" + tag], rtol=1e-12, atol=1e-9)
    for temp in (1.0, 0.6):
        xs = oracle_full.sample(g["eps"], temp, y, 100, 2)
        np.testing.assert_allclose(xs, g["sample_t%.1f_iso100_cam2" % temp], rtol=0, atol=1e-6 * np.abs(xs).max())


def test_philox_known_answers():
    from oracle import philox
    r = philox.philox4x32_10([0], [0], [0], [0], 0, 0)       # Random123 kat_vectors
    assert [int(v[0]) for v in r] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    r = philox.philox4x32_10([0xFFFFFFFF], [0xFFFFFFFF], [0xFFFFFFFF], [0xFFFFFFFF], 0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(v[0]) for v in r] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    r = philox.philox4x32_10([0x243F6A88], [0x85A308D3], [0x13198A2E], [0x03707344], 0xA4093822, 0x299F31D0)
    assert [int(v[0]) for v in r] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    x, y = philox.synth_patches(0, 0, 64)
    assert 0.0 <= y.min() and y.max() < 1.0 and abs(y.mean() - 0.5) < 0.01
    eps = x / np.sqrt(np.float32(0.000479) * y + np.float32(2e-6))
    assert abs(eps.std() - 1.0) < 0.01 and abs(eps.mean()) < 0.01


def test_secondary_layers_invariants():
    """sdn4 / sdn / gain: fresh-init closed forms, round trip, and the reference's
    once-per-patch log-det of the plain gain layer (AffineCouplingGain.py:113-127)."""
    y = np.random.RandomState(0).rand(2, 8, 8, 4)
    x = np.random.RandomState(1).randn(2, 8, 8, 4) * 0.1
    v = O.fresh_variables("sdn4|gain4")
    m = O.NoiseFlowOracle("sdn4|gain4", v)
    # fresh sdn4: beta1 = exp(-5), beta2 = 1, gain = exp(-5)*iso  ->  scale = sqrt(y/iso + 1)
    np.testing.assert_allclose(O.sdn_ex4_scale(y, m.layers[0]["p"], 400.0), np.sqrt(y / 400.0 + 1.0), rtol=1e-6)
    z, obj = m.inverse(x, y, 400.0, 2)
    assert np.abs(m.forward(z, y, 400.0, 2) - x).max() < 1e-15
    v = O.fresh_variables("sdn|gain")
    m = O.NoiseFlowOracle("sdn|gain", v)
    sig = lambda t: 1 / (1 + np.exp(-t))
    z, obj = m.inverse(x, y, 100.0, 2)
    scale = np.sqrt(sig(-3.0) * y + sig(3.0))
    g = sig(-3.0) * 100.0 + sig(3.0)
    np.testing.assert_allclose(z, x / scale / g, rtol=1e-12)
    np.testing.assert_allclose(obj, -np.log(scale).sum((1, 2, 3)) - np.log(g), rtol=1e-12)   # NOT -HWC*log(g)
    assert O.layer_names("sdn|unc|gain") == ["sdn_0", "Conv2d_1x1_1", "unc_1", "gain_2"]


@pytest.mark.parametrize("arch,width,hw", [(FULL_ARCH, 4, (32, 32)), ("unc|gain4|unc", 8, (9, 6)), ("sdn4|unc|gain|sdn", 4, (16, 16)),
                                           ("unc", 16, (5, 5))])
def test_c_oracle_matches_numpy_oracle(shipped_variables, arch, width, hw):
    """The plain-C restatement (fp32, un-folded, reference op order) against the fp64 numpy oracle."""
    from oracle.nf_oracle_c import COracle
    v = shipped_variables if arch == FULL_ARCH else trained_like_variables(arch, width, seed=4)
    if "model/g1" in v:
        v["model/g1"] = np.asarray([-6.0], np.float32)
    H, W = hw
    x, y = make_inputs(5, H, W, seed=7)
    o = O.NoiseFlowOracle(arch, v)
    c = COracle(arch, v)
    nll, sd, z = c.nll(x, y, 800.0, 2.0, want_z=True)
    ref, rsd, rz = o.nll(x, y, 800.0, 2.0)
    np.testing.assert_allclose(nll, ref, rtol=5e-6, atol=1e-3)
    assert np.abs(z - rz).max() <= 5e-6 * np.abs(rz).max()
    assert abs(sd.mean() - rsd) <= 1e-5 * rsd
    eps = np.random.RandomState(1).randn(5, H, W, 4).astype(np.float32)
    xs = c.sample(eps, 0.6, y, 800.0, 2.0)
    r = o.sample(eps, 0.6, y, 800.0, 2.0)
    assert np.abs(xs - r).max() <= 5e-6 * np.abs(r).max()


def test_config_c1_unconditional_flow_256_patches_cpu():
    """BASELINE configs[0]: a single unconditional Conv1x1 + AffineCoupling flow, 256 synthetic
    32x32x4 patches, CPU only.  Fresh init (reference initialisers): the coupling is the identity
    up to rescaling_scale*tanh(0) and A is orthogonal, so NLL = 3763.97 + 1/2 ||x||^2 exactly;
    with a trained-like coupling the plain-C and numpy restatements agree."""
    from oracle.nf_oracle_c import COracle
    from oracle import philox
    x, _ = philox.synth_patches(0, 0, 256)
    v = O.fresh_variables("unc", seed=0)
    nll, sd, _ = COracle("unc", v).nll(x)
    want = 0.5 * 4096 * np.log(2 * np.pi) + 0.5 * (x.astype(np.float64) ** 2).sum((1, 2, 3))
    np.testing.assert_allclose(nll, want, rtol=1e-6)
    np.testing.assert_allclose(O.NoiseFlowOracle("unc", v, sidd_cond="uncond").nll(x[:8])[0], want[:8], rtol=1e-7)   # Q is orthogonal only to fp32 round-off
    v = trained_like_variables("unc", 4, seed=1)
    a = COracle("unc", v).nll(x)[0]
    b = O.NoiseFlowOracle("unc", v).nll(x[:32])[0]
    np.testing.assert_allclose(a[:32], b, rtol=5e-6)
    assert np.isfinite(a).all()


def test_conv1x1_decompositions_agree():
    """matrix_param.py:23-29 / :100-140 / :143-188: the three parameterisations of one matrix give the same A, A^-1
    and log|det A|; tfb.Permute(reversed channels) is an involution with log|det| = 0."""
    import scipy.linalg as sla
    from oracle import nf_oracle as O
    rng = np.random.RandomState(3)
    q = (sla.qr(rng.randn(4, 4))[0] + 0.2 * rng.randn(4, 4)).astype(np.float32)
    v = {}
    for d in ("LU", "LU2", "NONE"):
        v.update(O.conv1x1_init_variables(q, 0, d))
    got = {d: O.conv1x1_from_variables(v, 0, d, np.float64) for d in ("LU", "LU2", "NONE")}
    for d, (A, Ai, lad) in got.items():
        np.testing.assert_allclose(A, q.astype(np.float64), rtol=0, atol=3e-7, err_msg=d)
        np.testing.assert_allclose(A @ Ai, np.eye(4), rtol=0, atol=2e-6, err_msg=d)
        assert abs(lad - np.log(abs(np.linalg.det(q.astype(np.float64))))) < 2e-6, d
    # LU2 ignores what sits outside the strict triangles of its full-matrix L / U variables (matrix_param.py:171-173)
    n = O.conv1x1_variable_names(0, "LU2")
    v2 = dict(v)
    v2[n["L"]] = v[n["L"]] + np.triu(np.ones((4, 4), np.float32))
    v2[n["U"]] = v[n["U"]] + np.tril(np.ones((4, 4), np.float32))
    np.testing.assert_array_equal(O.conv1x1_from_variables(v2, 0, "LU2", np.float64)[0], got["LU2"][0])
    layers = O.bind_variables("unc", O.fresh_variables("unc", flow_permutation=0), flow_permutation=0)
    assert [L["name"] for L in layers] == ["permute", "unc_0"] and layers[0]["log_abs_det"] == 0
    z = rng.randn(1, 2, 2, 4)
    np.testing.assert_array_equal(O.conv1x1_inverse(z, layers[0]["A"], 0.0)[0], z[..., ::-1])
    assert [L["name"] for L in O.bind_variables("unc", O.fresh_variables("unc", flow_permutation=2), flow_permutation=2)] == ["unc_0"]


def _variant_cases():
    import json
    d = np.load(os.path.join(GOLDEN_DIR, "arch_variants.npz"))
    meta = json.loads(str(d["meta"]))
    for i, m in enumerate(meta):
        tag = "c%d_" % i
        v = {k[len(tag) + 4:]: d[k] for k in d.files if k.startswith(tag + "var:")}
        yield m, v, {k: d[tag + k] for k in ("x", "y", "eps", "nll", "sdz", "z", "sample")}


def test_golden_arch_variants_freeze_the_oracle():
    """tests/golden/arch_variants.npz (tools/make_golden_variants.py): every sdn / gain layer key and the other settings of
    hps.flow_permutation / hps.decomp, 8x8 patches — the oracle still reproduces what it produced when they were committed."""
    from oracle.nf_oracle import NoiseFlowOracle
    n = 0
    for m, v, t in _variant_cases():
        o = NoiseFlowOracle(m["arch"], v, flow_permutation=m["flow_permutation"], decomp=m["decomp"])
        assert [L["name"] for L in o.layers] == m["layer_names"]
        nll, sd, z = o.nll(t["x"], t["y"], m["iso"], m["cam"])
        np.testing.assert_allclose(nll, t["nll"], rtol=1e-12)
        np.testing.assert_allclose(z, t["z"], rtol=0, atol=1e-12 * np.abs(t["z"]).max())
        np.testing.assert_allclose(o.sample(t["eps"], 0.7, t["y"], m["iso"], m["cam"]), t["sample"], rtol=0,
                                   atol=1e-12 * np.abs(t["sample"]).max())
        n += 1
    assert n == 7
