"""GPU parity of the batch-statistics mode (``is_training=True``, layers.py:386-398):
every batch_norm of the coupling CNNs uses the moments of the call's own patches.

The oracle side is ``NoiseFlowOracle(..., training=True)`` (fp64).  Tolerances are
those of the eval path: NLL 1e-5 relative, tensors 1e-5 of their scale; the batch
moments themselves (sums of up to 10^5 fp32 values, accumulated in fp64) 1e-5 of
the activation scale.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import FULL_ARCH, SHIPPED_DIR, make_inputs, trained_like_variables

pytestmark = pytest.mark.gpu

NLL_RTOL = 1e-5
ELEM_RTOL = 1e-5


def _model(arch, variables, x_shape=(32, 32, 4), width=4, binding="loss_first", training=True):
    from noise_flow_amd import NoiseFlow, default_hps
    return NoiseFlow(list(x_shape), training, default_hps(arch=arch, width=width), variables=variables, binding=binding)


def _oracle(arch, variables, binding="loss_first"):
    from oracle.nf_oracle import NoiseFlowOracle
    return NoiseFlowOracle(arch, variables, binding)


def _close_elem(a, ref, rtol=ELEM_RTOL):
    """Scale-relative bound + the masked per-element relative error for the session summary (tests/conftest.py::close_elem)."""
    from conftest import close_elem
    return close_elem(a, ref, rtol)


def _coupling_scopes(m):
    return m._flow.coupling_scopes


def test_nll_batchstats_full_arch(shipped_variables):
    x, y = make_inputs(16, seed=21, b1=0.003696)
    o = _oracle(FULL_ARCH, shipped_variables)
    ref_nll, ref_sd, ref_z = o.nll(x, y, 800, 2, training=True)
    eval_nll, _, _ = o.nll(x, y, 800, 2)
    assert np.abs(ref_nll - eval_nll).max() > 1e-3 * np.abs(eval_nll).max()   # the two modes really differ

    m = _model(FULL_ARCH, shipped_variables)
    before = {k: np.array(v, np.float32) for k, v in m.variables.items() if "bn_nvp_conv" in k}
    nll, sd_z = m._loss(x, y, [0.0], [0.0], [800], [2])
    np.testing.assert_allclose(nll, ref_nll, rtol=NLL_RTOL)
    assert abs(sd_z - ref_sd) <= 1e-5 * ref_sd

    # the running statistics moved by the reference's EMA (layers.py:392-393)
    names = [L["name"] for L in o.layers if L["type"] == "coupling"]
    for scope, lname in zip(_coupling_scopes(m), names):
        rec = o.last_batch_moments[lname]
        for bn, key in (("bn_nvp_conv_1/mean", "new_mean1"), ("bn_nvp_conv_1/var", "new_var1"),
                        ("bn_nvp_conv_2/mean", "new_mean2"), ("bn_nvp_conv_2/var", "new_var2")):
            got = m.variables[scope + "/" + bn]
            want = rec[key]
            scale = max(np.abs(want).max(), 1e-3)
            assert np.abs(got - want).max() <= 1e-5 * scale, (scope, bn)
            assert not np.array_equal(got, before[scope + "/" + bn])

    # latent and objective through inverse()
    m2 = _model(FULL_ARCH, shipped_variables)
    z, obj = m2.inverse(x, None, y, [0.0], [0.0], [800], [2])
    _close_elem(z, ref_z)


def test_sampling_batchstats_full_arch(shipped_variables):
    rng = np.random.RandomState(5)
    _, y = make_inputs(12, seed=9)
    eps = rng.randn(12, 32, 32, 4).astype(np.float32)
    o = _oracle(FULL_ARCH, shipped_variables)
    m = _model(FULL_ARCH, shipped_variables)
    for temp in (1.0, 0.6):
        ref = o.sample(eps, temp, y, 100, 2, training=True)
        xs = m.sample(y, temp, y, [0.0], [0.0], [100], [2], eps=eps)
        _close_elem(xs, ref)
    # in-kernel draw: every statistics pass must regenerate the same epsilon
    from oracle import philox
    e2 = philox.sample_eps(77, 0, 12)
    m3 = _model(FULL_ARCH, shipped_variables)
    xs = m3.sample(y, 0.6, y, [0.0], [0.0], [100], [2], seed=77)
    _close_elem(xs, o.sample(e2, 0.6, y, 100, 2, training=True), rtol=5e-5)


@pytest.mark.parametrize("arch,width,hw,B", [("unc|unc", 8, (24, 40), 6), ("unc|gain4|unc", 16, (16, 16), 9),
                                             ("sdn4|unc|gain4|unc", 4, (64, 64), 3),
                                             ("unc|unc", 5, (16, 20), 5), ("sdn5|unc|unc", 12, (16, 16), 4),      # zero-padded onto
                                             ("unc|gain4|unc", 24, (16, 16), 6), ("unc|unc", 3, (32, 32), 3)])     # 8 / 16 / 32 / 4
def test_batchstats_other_shapes(arch, width, hw, B):
    """Scalar-weight final pass (width != 4 / ragged shapes) and the 64x64 matrix-core one; widths between two kernel widths run
    zero-padded (a padded channel is 0 on every pixel: batch moments 0, its weights and bias stay 0), the moments the call
    reports — and the running statistics they move — are the model's own channels."""
    v = trained_like_variables(arch, width, seed=4)
    shape = (hw[0], hw[1], 4)
    x, y = make_inputs(B, hw[0], hw[1], seed=13)
    o = _oracle(arch, v)
    m = _model(arch, v, shape, width)
    nll, _ = m._loss(x, y, [0.0], [0.0], [400], [1])
    ref, _, _ = o.nll(x, y, 400, 1, training=True)
    np.testing.assert_allclose(nll, ref, rtol=NLL_RTOL)
    rng = np.random.RandomState(2)
    eps = rng.randn(B, *shape).astype(np.float32)
    # running statistics after the NLL call: the EMA of the model's own channels (layers.py:392-393)
    m3 = _model(arch, v, shape, width)
    m3._loss(x, y, [0.0], [0.0], [400], [1])
    o.nll(x, y, 400, 1, training=True)
    names = [L["name"] for L in o.layers if L["type"] == "coupling"]
    for scope, lname in zip(_coupling_scopes(m3), names):
        rec = o.last_batch_moments[lname]
        for bn, key in (("bn_nvp_conv_1/mean", "new_mean1"), ("bn_nvp_conv_1/var", "new_var1"),
                        ("bn_nvp_conv_2/mean", "new_mean2"), ("bn_nvp_conv_2/var", "new_var2")):
            got, want = np.asarray(m3.variables[scope + "/" + bn]).reshape(-1), np.asarray(rec[key]).reshape(-1)
            assert got.shape == want.shape == (width,)
            assert np.abs(got - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-3), (scope, bn)
    xs = m.sample(y, 1.0, y, [0.0], [0.0], [400], [1], eps=eps)
    _close_elem(xs, o.sample(eps, 1.0, y, 400, 1, training=True))


@pytest.mark.parametrize("arch,width,hw,B,iso,cam", [("sdn5|unc|gain4|unc", 64, (32, 32), 3, 800, 2),
                                                     ("unc|unc", 512, (32, 32), 2, 100, 1),        # the reference's default --width
                                                     ("sdn5|unc|unc|gain4|unc", 32, (64, 64), 2, 400, 0),   # configs[4] geometry at the paper's width
                                                     ("unc|unc", 50, (9, 7), 4, 1600, 3),           # not a multiple of 4, ragged shape
                                                     ("unc|gain4|unc", 16, (48, 64), 2, 800, 2),    # beyond the scalar kernel's LDS tiles
                                                     ("unc|unc", 132, (12, 20), 1, 400, 1),         # ONE patch: the batch is its pixels; 128 + 4 channels
                                                     ("|".join(["unc"] * 9), 40, (8, 8), 3, 100, 0),    # 9 couplings
                                                     # the paper's width on the training patch size: the patch-resident forward stages
                                                     # (csrc/nf_train_pr.h; 7 patches on 7 workgroups)
                                                     ("sdn5|unc|unc|gain4|unc", 32, (32, 32), 7, 800, 2)])
def test_batchstats_on_the_gemm_route(arch, width, hw, B, iso, cam):
    """NoiseFlowWrapper.py:86 runs the sampling graph with is_training=True and sidd/ArgParser.py:43 defaults --width to 512: the
    batch-statistics calls take every width and patch size.  Beyond 32 channels, and where a patch outgrows the scalar-weight
    kernel's LDS tiles (64x64 at width 32), the call walks the layers on the trainer's matrix-core GEMM path (csrc/nf_train.hip:
    nf_bs_wide_run).  Per-patch NLL, sd_z, latent, the batch moments' EMA and both sampling inputs against the fp64 oracle."""
    v = trained_like_variables(arch, width, seed=width + 1)
    for k in v:     # activations of O(1) at every width (the helper's weights are tuned for width 4)
        if k.endswith("l_2/W") or k.endswith("l_last/W"):
            v[k] = (v[k] * np.float32((4.0 / width) ** 0.5)).astype(np.float32)
    shape = (hw[0], hw[1], 4)
    x, y = make_inputs(B, hw[0], hw[1], seed=17)
    o = _oracle(arch, v)
    ref_nll, ref_sd, ref_z = o.nll(x, y, iso, cam, training=True)
    m = _model(arch, v, shape, width)
    nll, sd_z = m._loss(x, y, [0.0], [0.0], [iso], [cam])
    np.testing.assert_allclose(nll, ref_nll, rtol=NLL_RTOL)
    assert abs(sd_z - ref_sd) <= 1e-5 * ref_sd
    names = [L["name"] for L in o.layers if L["type"] == "coupling"]
    for scope, lname in zip(_coupling_scopes(m), names):
        rec = o.last_batch_moments[lname]
        for bn, key in (("bn_nvp_conv_1/mean", "new_mean1"), ("bn_nvp_conv_1/var", "new_var1"),
                        ("bn_nvp_conv_2/mean", "new_mean2"), ("bn_nvp_conv_2/var", "new_var2")):
            got, want = np.asarray(m.variables[scope + "/" + bn]).reshape(-1), np.asarray(rec[key]).reshape(-1)
            assert got.shape == want.shape == (width,)
            assert np.abs(got - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-3), (scope, bn)
    m2 = _model(arch, v, shape, width)
    z, _ = m2.inverse(x, None, y, [0.0], [0.0], [iso], [cam])
    _close_elem(z, ref_z)
    rng = np.random.RandomState(3)
    eps = rng.randn(B, *shape).astype(np.float32)
    m3 = _model(arch, v, shape, width)
    for temp in (1.0, 0.6):
        _close_elem(m3.sample(y, temp, y, [0.0], [0.0], [iso], [cam], eps=eps), o.sample(eps, temp, y, iso, cam, training=True))
    if hw == (32, 32):     # the in-kernel draw: the evaluator regenerates the fused kernels' epsilon
        from oracle import philox
        e2 = philox.sample_eps(41, 0, B)
        m4 = _model(arch, v, shape, width)
        _close_elem(m4.sample(y, 0.6, y, [0.0], [0.0], [iso], [cam], seed=41), o.sample(e2, 0.6, y, iso, cam, training=True), rtol=5e-5)


def test_the_two_batchstats_routes_agree(shipped_variables, monkeypatch):
    """The shipped model (width 4) through the fused kernels' statistics passes and, NF_BS_WIDE=1, through the GEMM-route
    evaluator: same per-patch NLL (1e-5), same samples, same moments — two implementations of layers.py:386-398."""
    x, y = make_inputs(6, seed=23, b1=0.003696)
    rng = np.random.RandomState(8)
    eps = rng.randn(6, 32, 32, 4).astype(np.float32)
    res = {}
    for mode in ("0", "1"):
        if mode == "1":
            monkeypatch.setenv("NF_BS_WIDE", "1")
        else:
            monkeypatch.delenv("NF_BS_WIDE", raising=False)
        m = _model(FULL_ARCH, shipped_variables)
        nll, sd = m._loss(x, y, [0.0], [0.0], [800], [2])
        mom = {k: np.array(v_) for k, v_ in m.variables.items() if "bn_nvp_conv" in k}
        xs = _model(FULL_ARCH, shipped_variables).sample(y, 0.6, y, [0.0], [0.0], [100], [2], eps=eps)
        res[mode] = (np.asarray(nll), sd, mom, np.asarray(xs))
    np.testing.assert_allclose(res["1"][0], res["0"][0], rtol=NLL_RTOL)
    assert abs(res["1"][1] - res["0"][1]) <= 1e-5 * res["0"][1]
    for k in res["0"][2]:
        assert np.abs(res["1"][2][k] - res["0"][2][k]).max() <= 1e-5 * max(np.abs(res["0"][2][k]).max(), 1e-3), k
    _close_elem(res["1"][3], res["0"][3].astype(np.float64))


def test_batchstats_single_patch_and_large_batch(shipped_variables):
    """B = 1 (moments over one patch) against the oracle; B = 1024 against the plain-C fp32 oracle's
    structure is not available in training mode, so the large batch is checked through the property
    that holds by construction: feeding the batch moments back as RUNNING statistics into an
    eval-mode model reproduces the batch-mode NLL."""
    import torch
    from noise_flow_amd import patches
    o = _oracle(FULL_ARCH, shipped_variables)
    x, y = make_inputs(1, seed=31)
    m = _model(FULL_ARCH, shipped_variables)
    nll, _ = m._loss(x, y, [0.0], [0.0], [100], [2])
    np.testing.assert_allclose(nll, o.nll(x, y, 100, 2, training=True)[0], rtol=NLL_RTOL)

    B = 1024
    xs, ys = patches.synth_patches(5, 0, B, 32, 32, (0.003696, 1e-5))
    mt = _model(FULL_ARCH, shipped_variables)
    v0 = {k: np.array(v) for k, v in mt.variables.items()}
    nll_b, _ = mt._loss(xs, ys, [0.0], [0.0], [800], [2])
    # recover the batch moments from the EMA: new = old - 0.1 (old - m)  =>  m = old + (new - old) / 0.1
    v1 = dict(v0)
    for k in v0:
        if "bn_nvp_conv" in k:
            v1[k] = (v0[k].astype(np.float64) + (mt.variables[k].astype(np.float64) - v0[k]) / 0.1).astype(np.float32)
    me = _model(FULL_ARCH, v1, training=False)
    nll_e, _ = me._loss(xs, ys, [0.0], [0.0], [800], [2])
    assert torch.is_tensor(nll_b)
    # the moments pass through one fp32 EMA round trip (relative 1e-6 on a 10x amplified difference)
    np.testing.assert_allclose(nll_b.cpu().numpy(), nll_e.cpu().numpy(), rtol=2e-4)


def test_wrapper_batch_mode_is_the_literal_reference_wrapper():
    """NoiseFlowWrapper.py:49,64,86: sampling graph only ('sample_first' binding) + is_training=True."""
    from noise_flow_amd import NoiseFlowWrapper
    from noise_flow_amd.ckpt import load_checkpoint
    import os
    w = NoiseFlowWrapper(SHIPPED_DIR, sampling_temperature=0.6, binding="sample_first", bn_mode="batch", seed=3)
    _, y = make_inputs(8, seed=2)
    out = w.sample_noise_nf(y, 0.0, 0.0, 100.0, 2.0)
    from oracle import philox
    eps = philox.sample_eps(3, 0, 8)
    o = _oracle(FULL_ARCH, load_checkpoint(os.path.join(SHIPPED_DIR, "ckpt", "model.ckpt.best")), "sample_first")
    _close_elem(out, o.sample(eps, 0.6, y, 100, 2, training=True), rtol=5e-5)


def test_batchstats_c_abi_errors(shipped_variables):
    import torch
    from noise_flow_amd import _lib, NoiseFlow, default_hps
    lib = _lib.load()
    m = _model(FULL_ARCH, shipped_variables)
    x = torch.zeros(2, 32, 32, 4, device="cuda")
    cond = _lib.nf_cond(100.0, 2.0, 0.0, 0.0)
    # empty batch: moments undefined
    rc = lib.nf_nll_batchstats(m._flow.ptr, x.data_ptr(), x.data_ptr(), 0, C.byref(cond), None, None, None, None, None,
                               0, None, None)
    assert rc == _lib.NF_EINVAL and b"empty" in lib.nf_last_error()
    # fp16-CNN handles are eval-only
    mh = NoiseFlow([32, 32, 4], False, default_hps(arch=FULL_ARCH), variables=shipped_variables, cnn_dtype="fp16")
    out = torch.empty(2, device="cuda")
    rc = lib.nf_nll_batchstats(mh._flow.ptr, x.data_ptr(), x.data_ptr(), 2, C.byref(cond), out.data_ptr(), None, None,
                               None, None, 0, None, None)
    assert rc == _lib.NF_EINVAL and b"fp32 only" in lib.nf_last_error()
    # moments_out is optional
    rc = lib.nf_nll_batchstats(m._flow.ptr, x.data_ptr(), x.data_ptr(), 2, C.byref(cond), out.data_ptr(), None, None,
                               None, None, 0, None, None)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()


def _bs_sync_worker(rank, world, port, outdir, arch, width, hw, per):
    """One rank of a batch-statistics evaluation with cross-rank moments (NoiseFlow.set_sync_bn -> nf_set_sync)."""
    import os
    import sys
    import torch.distributed as dist
    from conftest import ROOT, make_inputs, trained_like_variables
    sys.path.insert(0, ROOT)
    from noise_flow_amd import NoiseFlow, default_hps
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    v = trained_like_variables(arch, width, seed=4)
    m = NoiseFlow([hw, hw, 4], True, default_hps(arch=arch, width=width), variables=v)
    m.set_sync_bn(True)
    x, y = make_inputs(per * world, hw, hw, seed=31, b1=0.003696)
    sl = slice(per * rank, per * rank + per)
    nll, sd = m._loss(x[sl], y[sl], [0.0], [0.0], [800], [2])
    eps = np.random.RandomState(5).randn(per * world, hw, hw, 4).astype(np.float32)
    xs = m.sample(y[sl], 0.8, y[sl], [0.0], [0.0], [800], [2], eps=eps[sl])
    np.save(os.path.join(outdir, "bs_nll_%d.npy" % rank), nll)
    np.save(os.path.join(outdir, "bs_xs_%d.npy" % rank), xs)
    np.savez(os.path.join(outdir, "bs_bn_%d.npz" % rank), **{k.replace("/", "!"): a for k, a in m.variables.items() if "bn_nvp_conv" in k})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("arch,width,hw,per", [("sdn5|unc|unc|gain4|unc", 4, 32, 6), ("sdn5|unc|gain4|unc", 16, 16, 5),
                                               ("sdn5|unc|gain4|unc", 32, 32, 3)])     # the patch-resident forward stages
def test_two_rank_batch_statistics_evaluation_equals_one_rank_on_the_union(tmp_path, arch, width, hw, per):
    """`is_training=True` forward / sampling across ranks (layers.py:386-398: the moments are those of the whole minibatch):
    with the nf_set_sync hook, 2 ranks x `per` patches give per-patch NLLs, samples and running-statistics updates equal to
    one rank on the 2*per concatenated patches — on the fused width-4 path and on the generic (width 16) path."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_bs_sync_worker, args=(r, 2, port, str(tmp_path), arch, width, hw, per)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(timeout=300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    v = trained_like_variables(arch, width, seed=4)
    x, y = make_inputs(per * 2, hw, hw, seed=31, b1=0.003696)
    one = _model(arch, v, (hw, hw, 4), width)
    nll1, _ = one._loss(x, y, [0.0], [0.0], [800], [2])
    bn_after_nll = {k: np.array(a) for k, a in one.variables.items() if "bn_nvp_conv" in k}
    eps = np.random.RandomState(5).randn(per * 2, hw, hw, 4).astype(np.float32)
    xs1 = one.sample(y, 0.8, y, [0.0], [0.0], [800], [2], eps=eps)
    nll2 = np.concatenate([np.load(str(tmp_path / ("bs_nll_%d.npy" % r))) for r in range(2)])
    xs2 = np.concatenate([np.load(str(tmp_path / ("bs_xs_%d.npy" % r))) for r in range(2)])
    np.testing.assert_allclose(nll2, nll1, rtol=NLL_RTOL)
    _close_elem(xs2, xs1.astype(np.float64))
    # without the hook the shards normalise with their own moments: measurably different
    half = _model(arch, v, (hw, hw, 4), width)
    nll_h, _ = half._loss(x[:per], y[:per], [0.0], [0.0], [800], [2])
    assert np.abs(nll_h - nll1[:per]).max() > 1e-4 * np.abs(nll1).max()
    # both ranks moved their running statistics identically, by the GLOBAL moments (2 EMA steps: _loss, then sample)
    bn0, bn1 = (np.load(str(tmp_path / ("bs_bn_%d.npz" % r))) for r in range(2))
    final = {k: np.array(a) for k, a in one.variables.items() if "bn_nvp_conv" in k}
    for k, a in final.items():
        kk = k.replace("/", "!")
        assert np.array_equal(bn0[kk], bn1[kk]), k
        assert np.abs(bn0[kk] - a).max() <= 1e-5 * max(np.abs(a).max(), 1e-3), k
        assert np.abs(a - bn_after_nll[k]).max() > 0 or "var" in k
