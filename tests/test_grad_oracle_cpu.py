"""CPU tier: the training-step oracle (oracle/nf_grad_oracle.py) pinned to the forward oracle and
to central finite differences of it; the raw-layout pack/unpack round trip the trainer relies on."""
import numpy as np

from conftest import trained_like_variables, make_inputs


ARCH = "sdn5|unc|gain4|unc"


def _setup(B=3, hw=(12, 10), width=4, seed=3):
    v = trained_like_variables(ARCH, width, seed=seed)
    x, y = make_inputs(B, hw[0], hw[1], seed=seed + 1, b1=0.003696)
    return v, x, y


def test_forward_value_equals_the_forward_oracle_in_training_mode():
    from oracle.nf_oracle import NoiseFlowOracle
    from oracle.nf_grad_oracle import GradOracle
    v, x, y = _setup()
    ref = NoiseFlowOracle(ARCH, v)
    nll, sd, _ = ref.nll(x, y, 800, 2, training=True)
    loss, sd_z, grads, new_running = GradOracle(ARCH, v).loss_and_grads(x, y, 800, 2)
    assert abs(loss - nll.mean()) <= 1e-12 * abs(nll.mean())
    assert abs(sd_z - sd) <= 1e-12 * sd
    # EMA targets = what the forward oracle records
    for lname, rec in ref.last_batch_moments.items():
        assert set(("new_mean1", "new_var1", "new_mean2", "new_var2")) <= set(rec)
    got = sorted(k for k in new_running)
    assert len(got) == 8 and all(k.endswith("/mean") or k.endswith("/var") for k in got)


def test_gradients_match_central_finite_differences():
    from oracle.nf_oracle import NoiseFlowOracle
    from oracle.nf_grad_oracle import GradOracle, is_trainable
    v, x, y = _setup()
    _, _, grads, _ = GradOracle(ARCH, v).loss_and_grads(x, y, 800, 2)
    f = lambda vv: NoiseFlowOracle(ARCH, vv).nll(x, y, 800, 2, training=True)[0].mean()
    rng = np.random.RandomState(0)
    gmax = max(np.abs(g).max() for g in grads.values())
    names = sorted(k for k in v if is_trainable(k) and k in grads)
    for k in names:
        a = np.asarray(v[k], np.float64)
        idx = tuple(rng.randint(s) for s in a.shape) if a.shape else ()
        h = 1e-7
        vp, vm = dict(v), dict(v)
        ap, am = a.copy(), a.copy()
        ap[idx] += h
        am[idx] -= h
        vp[k], vm[k] = ap, am
        fd = (f(vp) - f(vm)) / (2 * h)
        assert abs(fd - grads[k][idx]) <= 1e-5 * max(abs(fd), 1e-3 * gmax), (k, fd, grads[k][idx])


def test_adam_and_momentum_restatements():
    from oracle.nf_grad_oracle import adam_step, momentum_step
    v = {"a": np.array([1.0, -2.0]), "b": np.array([0.5])}
    g = {"a": np.array([0.1, -0.3])}
    st = {}
    out = adam_step(v, g, st, 0.01)
    # first Adam step moves every entry with a gradient by lr * sign(g) (bias-corrected)
    np.testing.assert_allclose(out["a"], v["a"] - 0.01 * np.sign(g["a"]), rtol=1e-6)
    assert out["b"] is v["b"] and st["t"] == 1
    st2 = {}
    o1 = momentum_step(v, g, st2, 0.1)
    o2 = momentum_step(o1, g, st2, 0.1)
    np.testing.assert_allclose(o2["a"], v["a"] - 0.1 * g["a"] - 0.1 * (1.9 * g["a"]))


def test_raw_layout_pack_unpack_round_trip():
    from noise_flow_amd import params as P
    v = trained_like_variables(ARCH, 8, seed=5)
    layers = P.parse_arch(ARCH)
    tmpl = P.template_binding(layers, "loss_first")
    layers, descs, flat = P.pack_layers(layers, v, 8, tmpl)
    back = P.unpack_layers(layers, flat, v, tmpl)
    for k in v:
        assert np.array_equal(np.asarray(back[k], np.float32).reshape(-1), np.asarray(v[k], np.float32).reshape(-1)), k
        assert np.asarray(back[k]).shape == np.asarray(v[k]).shape
    flat2 = flat + 1.0
    moved = P.unpack_layers(layers, flat2, v, tmpl)
    owned = [nm for L in layers for nm in P.layer_variable_names(L, tmpl) if nm is not None]
    for k in owned:
        np.testing.assert_allclose(np.asarray(moved[k]).reshape(-1), np.asarray(v[k], np.float32).reshape(-1) + 1.0)
    assert sum(int(np.asarray(v[k]).size) for k in owned) + 1 == flat.size   # + sdn5's constant c_i


def test_training_golden_fixture_matches_the_oracle(shipped_variables):
    """tests/golden/train_step_shipped.npz (tools/make_golden_train.py) freezes the training oracle."""
    import os
    from conftest import ROOT
    from oracle.nf_grad_oracle import GradOracle, adam_step
    g = np.load(os.path.join(ROOT, "tests", "golden", "train_step_shipped.npz"))
    arch = str(g["arch"])
    loss, sd_z, grads, new_running = GradOracle(arch, shipped_variables).loss_and_grads(g["x"], g["y"], int(g["iso"]), int(g["cam"]))
    assert abs(loss - float(g["loss"])) <= 1e-10 * abs(loss) and abs(sd_z - float(g["sd_z"])) <= 1e-10 * sd_z
    n = 0
    gmax = max(float(np.abs(g["grad/" + k]).max()) for k in grads)
    for k, v in grads.items():
        ref = g["grad/" + k]
        # l_1/b and l_2/b are analytically zero (BN subtracts the batch mean): what the fixture holds there is fp64 round-off of
        # 1e-12, which moves with the order of the oracle's own sums -> compared on the scale of the step's gradients
        assert np.abs(np.asarray(v, np.float32) - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1e-6 * gmax), k
        n += ref.size
    assert n == 2433
    for k, v in new_running.items():
        assert np.abs(np.asarray(v, np.float32) - g["bn/" + k]).max() <= 1e-6 * max(np.abs(g["bn/" + k]).max(), 1e-6), k
    after = adam_step(shipped_variables, grads, {}, float(g["lr"]))
    for k in grads:
        np.testing.assert_allclose(np.asarray(after[k], np.float32), g["adam/" + k], rtol=1e-6, atol=1e-9)


def test_wide_training_golden_fixture_matches_the_oracle():
    """tests/golden/train_step_width48.npz (tools/make_golden_train.py wide): a step at a coupling width the trainer runs on its
    library-GEMM path; the model is part of the fixture."""
    import os
    from conftest import ROOT
    from oracle.nf_grad_oracle import GradOracle
    g = np.load(os.path.join(ROOT, "tests", "golden", "train_step_width48.npz"))
    v = {k[4:]: g[k] for k in g.files if k.startswith("var/")}
    o = GradOracle(str(g["arch"]), v)
    loss, sd_z, grads, new_running = o.loss_and_grads(g["x"], g["y"], int(g["iso"]), int(g["cam"]))
    assert not o.kinks
    assert abs(loss - float(g["loss"])) <= 1e-10 * abs(loss) and abs(sd_z - float(g["sd_z"])) <= 1e-10 * sd_z
    gmax = max(float(np.abs(g["grad/" + k]).max()) for k in grads)
    for k, a in grads.items():
        ref = g["grad/" + k]
        assert np.abs(np.asarray(a, np.float32) - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1e-6 * gmax), k
        np.testing.assert_allclose(np.asarray(o.grad_abs_terms[k], np.float32), g["abs/" + k], rtol=1e-5, atol=1e-12)
    for k, a in new_running.items():
        assert np.abs(np.asarray(a, np.float32) - g["bn/" + k]).max() <= 1e-6 * max(np.abs(g["bn/" + k]).max(), 1e-6), k


def test_autograd_oracle_covers_the_whole_vocabulary_and_every_permutation_setting():
    """The rest of ``noise_flow_arch``'s layer keys and the other ``hps.flow_permutation`` / ``hps.decomp`` settings in the
    autograd oracle: the forward value equals the numpy forward oracle's (training-mode BN) and a sample of gradient entries
    equals central finite differences of it — which is what pins the gradients the trainer is held to."""
    import json
    import os
    from conftest import GOLDEN_DIR
    from oracle.nf_oracle import NoiseFlowOracle
    from oracle.nf_grad_oracle import GradOracle, is_trainable
    d = np.load(os.path.join(GOLDEN_DIR, "arch_variants.npz"))
    meta = json.loads(str(d["meta"]))
    rng = np.random.RandomState(1)
    for i, m in enumerate(meta):
        tag = "c%d_" % i
        v = {k[len(tag) + 4:]: d[k] for k in d.files if k.startswith(tag + "var:")}
        x, y = d[tag + "x"], d[tag + "y"]
        kw = dict(flow_permutation=m["flow_permutation"], decomp=m["decomp"])
        f = lambda vv: NoiseFlowOracle(m["arch"], vv, **kw).nll(x, y, m["iso"], m["cam"], training=True)[0].mean()
        loss, _, grads, _ = GradOracle(m["arch"], v, **kw).loss_and_grads(x, y, m["iso"], m["cam"])
        assert abs(loss - f(v)) <= 1e-11 * abs(loss), m["arch"]
        gmax = max(np.abs(g).max() for g in grads.values())
        names = [k for k in sorted(v) if is_trainable(k) and k in grads and not ("real_nvp_conv_template" in k)]
        assert names, m["arch"]
        for k in names:                                  # every scalar-layer / mixing-layer variable, one entry each
            a = np.asarray(v[k], np.float64)
            nz = np.argwhere(np.abs(grads[k]) > 0)
            idx = tuple(nz[rng.randint(len(nz))]) if len(nz) else tuple(0 for _ in a.shape)
            h = 1e-6 * max(1.0, abs(float(a[idx])))
            vp, vm = dict(v), dict(v)
            ap, am = a.copy(), a.copy()
            ap[idx] += h
            am[idx] -= h
            vp[k], vm[k] = ap, am
            fd = (f(vp) - f(vm)) / (2 * h)
            assert abs(fd - grads[k][idx]) <= 2e-5 * max(abs(fd), 1e-6 * gmax), (m["arch"], k, fd, grads[k][idx])


def test_relu_kinks_are_reported_and_can_be_flipped():
    """The deterministic replacement of 're-draw the input next to a kink' (tests/conftest.py::grads_match_up_to_kinks)."""
    import pytest
    from conftest import grads_match_up_to_kinks
    from oracle.nf_grad_oracle import GradOracle
    v, x, y = _setup(B=2, hw=(8, 8))
    o = GradOracle(ARCH, v)
    loss, sd, g0, _ = o.loss_and_grads(x, y, 800, 2)
    assert o.kinks == []                                   # a generic input has no activation within 32 ulp of a kink
    o.kink_ulps = 8e3                                      # widen the band artificially to get candidates
    o.loss_and_grads(x, y, 800, 2)
    cands = [(s, k) for s, k, _ in o.kinks]
    assert 1 <= len(cands) <= 8, len(cands)
    assert all(m < 8e3 for _, _, m in o.kinks)
    loss1, sd1, g1, _ = o.loss_and_grads(x, y, 800, 2, relu_flips=[cands[0]])
    assert abs(loss1 - loss) <= 1e-5 * abs(loss)           # the forward value moves by the (tiny) activation only
    diff = max(np.abs(g1[k] - g0[k]).max() / max(np.abs(g0[k]).max(), 1e-30) for k in g0)
    assert diff > 1e-6                                     # ... but the gradient jumps: the branch matters
    # an evaluation that took the other branch at candidate 0 is explained by exactly one flip, and by nothing else
    def compare_to(target):
        def compare(l, s, g):
            for k in target:
                assert np.abs(g[k] - target[k]).max() <= 1e-9 * max(np.abs(target[k]).max(), 1e-30), k
        return compare
    assert grads_match_up_to_kinks(o, x, y, 800, 2, compare_to(g0)) == 0
    assert grads_match_up_to_kinks(o, x, y, 800, 2, compare_to(g1)) == 1
    bogus = {k: a * 1.01 for k, a in g0.items()}
    with pytest.raises(AssertionError):
        grads_match_up_to_kinks(o, x, y, 800, 2, compare_to(bogus))
    o.kink_ulps = 32.0
    with pytest.raises(AssertionError, match="nothing to excuse"):
        grads_match_up_to_kinks(o, x, y, 800, 2, compare_to(g1))


def test_many_kinks_are_solved_for_not_searched():
    """More than 6 candidates: the flipped subset is fitted by least squares in the candidates' gradient moves, rounded,
    and verified by an exact re-evaluation (tests/conftest.py::grads_match_up_to_kinks)."""
    from conftest import grads_match_up_to_kinks
    from oracle.nf_grad_oracle import GradOracle
    v, x, y = _setup(B=2, hw=(8, 8))
    o = GradOracle(ARCH, v)
    o.kink_ulps = 1.2e4
    o.loss_and_grads(x, y, 800, 2)
    c = [(s, k) for s, k, _ in o.kinks]
    assert len(c) > 6
    tgt = o.loss_and_grads(x, y, 800, 2, relu_flips=[c[1], c[4], c[7]])[2]

    def compare(l, s, g):
        for k in tgt:
            assert np.abs(g[k] - tgt[k]).max() <= 1e-9 * max(np.abs(tgt[k]).max(), 1e-30), k
    assert grads_match_up_to_kinks(o, x, y, 800, 2, compare, max_kinks=48, got=tgt) == 3


def test_abs_terms_bound_the_gradient_and_the_round_off_of_an_fp32_evaluation():
    """`GradOracle.grad_abs_terms` — per gradient entry, the sum of |term| over the batch x pixel contributions it is the sum
    of (the two parts of the loss counted separately) — is what the GPU tests' round-off allowance is made of
    (conftest.py::grad_noise_allowance).  Pinned here without a GPU: (i) triangle inequality, |gradient| <= sum |terms|, entry
    by entry; (ii) for a gain layer the closed form: d loss / d g = (C H W - sum_e z_e^2) / g per patch, so the terms of the two
    parts are C H W / g and sum z^2 / g; (iii) the SAME op sequence evaluated in float32 (GradOracle(dtype=float32): an
    fp32 evaluation that is not one of the kernels under test) sits within GRAD_NOISE_C * 2^-24 * terms or 2e-4 of the tensor's
    scale of the fp64 gradient — the tolerance the kernels are held to."""
    import torch
    from conftest import GRAD_NOISE_C
    from oracle.nf_grad_oracle import GradOracle
    for arch, width, hw, seed in ((ARCH, 4, (12, 10), 3), ("sdn5|unc|gain4|unc", 8, (8, 12), 5), ("gain4", 4, (6, 6), 7)):
        v = trained_like_variables(arch, width, seed=seed)
        x, y = make_inputs(4, hw[0], hw[1], seed=seed + 1, b1=0.003696)
        o = GradOracle(arch, v)
        _, _, g, _ = o.loss_and_grads(x, y, 800, 2)
        if o.kinks:          # an activation on its ReLU kink may take the other branch in float32: not what is measured here
            continue
        _, _, g32, _ = GradOracle(arch, v, dtype=torch.float32).loss_and_grads(x, y, 800, 2)
        gmax = max(np.abs(a).max() for a in g.values())
        for k in g:
            T = o.grad_abs_terms[k]
            assert T.shape == g[k].shape and (T >= np.abs(g[k]) * (1 - 1e-9)).all(), k
            tol = np.maximum(2e-4 * max(np.abs(g[k]).max(), 1e-6 * gmax), GRAD_NOISE_C * 2.0 ** -24 * T)
            assert (np.abs(g32[k].astype(np.float64) - g[k]) <= tol).all(), (arch, k)
        if arch == "gain4":
            gv = float(v["model/sdn_gain/gain_val"][0])
            z2 = float((np.asarray(x, np.float64) ** 2).sum()) / gv ** 2
            n = x.shape[0]
            want = (x[0].size * n / gv + z2 / gv) / n          # mean over the batch of C H W / g and sum z^2 / g
            got = float(o.grad_abs_terms["model/sdn_gain/gain_val"].reshape(-1)[0])
            assert abs(got - want) <= 1e-9 * want, (got, want)
