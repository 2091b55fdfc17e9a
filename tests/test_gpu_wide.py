"""GPU parity of the wide coupling CNN (hps.width = 32, the paper-scale width of job_noise_flow.sh:19) on the
f32 matrix cores (csrc/nf_wide.hip: v_mfma_f32_32x32x2_f32, strips of 8 rows per wavefront) against the fp64 oracle:
full and ragged patch shapes up to 64x64 (one and two tiles per row, partial strips), both directions, in-kernel
Philox.  Tolerances as everywhere: per-patch NLL 1e-5 relative, tensors 1e-5 of their scale."""
import numpy as np
import pytest

from conftest import make_inputs, trained_like_variables

pytestmark = pytest.mark.gpu

NLL_RTOL = 1e-5
ELEM_RTOL = 1e-5
ARCH = "sdn5|unc|gain4|unc"


def _model(arch, variables, x_shape, width):
    from noise_flow_amd import NoiseFlow, default_hps
    return NoiseFlow(list(x_shape), False, default_hps(arch=arch, width=width), variables=variables)


def _close_elem(a, ref, rtol=ELEM_RTOL):
    """Scale-relative bound + the masked per-element relative error for the session summary (tests/conftest.py::close_elem)."""
    from conftest import close_elem
    return close_elem(a, ref, rtol)


def _path(m, direction=0):
    return m._flow.lib.nf_kernel_path(m._flow.ptr, direction)


@pytest.mark.parametrize("hw", [(32, 32), (64, 64), (20, 28), (40, 50), (33, 64), (64, 33), (8, 32), (1, 1), (9, 33),
                                (16, 16), (64, 32), (32, 64), (7, 5)])
def test_wide32_matrix_core_kernel_matches_oracle(hw):
    from noise_flow_amd import _lib
    from oracle.nf_oracle import NoiseFlowOracle
    H, W = hw
    v = trained_like_variables(ARCH, 32, seed=H * 100 + W)
    x, y = make_inputs(5, H, W, seed=3)
    m = _model(ARCH, v, (H, W, 4), 32)
    assert _path(m, 0) == _lib.NF_PATH_WIDE32 and _path(m, 1) == _lib.NF_PATH_WIDE32
    o = NoiseFlowOracle(ARCH, v)
    nll, sd = m._loss(x, y, [0.0], [0.0], [100], [2])
    ref_nll, ref_sd, ref_z = o.nll(x, y, 100, 2)
    np.testing.assert_allclose(nll, ref_nll, rtol=NLL_RTOL, atol=1e-4)
    assert abs(sd - ref_sd) <= 1e-5 * ref_sd
    z, obj = m.inverse(x, None, y, [0.0], [0.0], [100], [2])
    _close_elem(z, ref_z)
    eps = np.random.RandomState(4).randn(5, H, W, 4).astype(np.float32)
    xs = m.sample(y, 0.8, y, [0.0], [0.0], [100], [2], eps=eps)
    _close_elem(xs, o.sample(eps, 0.8, y, 100, 2))


@pytest.mark.parametrize("width,hw", [(8, (64, 64)), (8, (60, 64))])
def test_narrower_widths_on_large_patches_use_the_wide_kernel(width, hw):
    """Width 8 beyond the scalar-weight kernel's LDS tile (round 1 rejected it at create): zero-padded to 32 hidden
    channels on the matrix-core kernel — exact, since a padded channel is identically zero."""
    from noise_flow_amd import _lib
    from oracle.nf_oracle import NoiseFlowOracle
    H, W = hw
    v = trained_like_variables(ARCH, width, seed=H + W + width)
    x, y = make_inputs(3, H, W, seed=8)
    m = _model(ARCH, v, (H, W, 4), width)
    assert _path(m, 0) == _lib.NF_PATH_WIDE32
    o = NoiseFlowOracle(ARCH, v)
    nll, sd = m._loss(x, y, [0.0], [0.0], [100], [2])
    ref_nll, ref_sd, ref_z = o.nll(x, y, 100, 2)
    np.testing.assert_allclose(nll, ref_nll, rtol=NLL_RTOL, atol=1e-4)
    z, _ = m.inverse(x, None, y, [0.0], [0.0], [100], [2])
    _close_elem(z, ref_z)
    eps = np.random.RandomState(4).randn(3, H, W, 4).astype(np.float32)
    _close_elem(m.sample(y, 0.8, y, [0.0], [0.0], [100], [2], eps=eps), o.sample(eps, 0.8, y, 100, 2))
    # a shape the scalar kernel holds stays on it
    assert _path(_model(ARCH, trained_like_variables(ARCH, width, seed=1), (32, 32, 4), width), 0) == _lib.NF_PATH_SCALAR


@pytest.mark.parametrize("hw", [(32, 32), (64, 64), (20, 28), (40, 50), (33, 64), (64, 33), (8, 32), (1, 1), (9, 33), (16, 16),
                                (64, 32), (48, 48), (7, 5), (17, 16)])
def test_wide16_matrix_core_kernel_matches_oracle(hw):
    """Width 16 on v_mfma_f32_16x16x4_f32 (csrc/nf_wide16.hip): tiles of 16 pixels, strips of 8 or 16 rows, up to 4 column
    blocks per row with their seams, ragged shapes."""
    from noise_flow_amd import _lib
    from oracle.nf_oracle import NoiseFlowOracle
    H, W = hw
    v = trained_like_variables(ARCH, 16, seed=H * 100 + W)
    x, y = make_inputs(5, H, W, seed=3)
    m = _model(ARCH, v, (H, W, 4), 16)
    assert _path(m, 0) == _lib.NF_PATH_WIDE16 and _path(m, 1) == _lib.NF_PATH_WIDE16
    o = NoiseFlowOracle(ARCH, v)
    nll, sd = m._loss(x, y, [0.0], [0.0], [100], [2])
    ref_nll, ref_sd, ref_z = o.nll(x, y, 100, 2)
    np.testing.assert_allclose(nll, ref_nll, rtol=NLL_RTOL, atol=1e-4)
    assert abs(sd - ref_sd) <= 1e-5 * ref_sd
    z, obj = m.inverse(x, None, y, [0.0], [0.0], [100], [2])
    _close_elem(z, ref_z)
    eps = np.random.RandomState(4).randn(5, H, W, 4).astype(np.float32)
    xs = m.sample(y, 0.8, y, [0.0], [0.0], [100], [2], eps=eps)
    _close_elem(xs, o.sample(eps, 0.8, y, 100, 2))
    xs2 = m.sample(y, 0.7, y, [0.0], [0.0], [100], [2], seed=5)     # in-kernel Philox instantiation
    assert np.isfinite(xs2).all() and xs2.shape == x.shape


def test_wide32_full_arch_batch_and_round_trip():
    """The shipped layer sequence at width 32, a batch larger than the resident grid (persistent stride loop),
    slotted sums, and sample(nll(x)) = x."""
    from conftest import FULL_ARCH
    from oracle.nf_oracle import NoiseFlowOracle
    v = trained_like_variables(FULL_ARCH, 32, seed=11)
    for k in v:   # keep the 8-coupling stack as well conditioned at 32 channels as the helper's weights are at 4
        if k.endswith("l_2/W") or k.endswith("l_last/W"):
            v[k] = (v[k] * np.float32((4.0 / 32.0) ** 0.5)).astype(np.float32)
    B = 1100
    x, y = make_inputs(B, 32, 32, seed=5)
    m = _model(FULL_ARCH, v, (32, 32, 4), 32)
    nll, sd = m._loss(x, y, [0.0], [0.0], [100], [2])
    idx = np.r_[0:8, B - 8:B]
    o = NoiseFlowOracle(FULL_ARCH, v)
    ref_nll, _, ref_z = o.nll(x[idx], y[idx], 100, 2)
    np.testing.assert_allclose(nll[idx], ref_nll, rtol=NLL_RTOL, atol=1e-4)
    mean, _ = m.loss(x, y, [0.0], [0.0], [100], [2])
    assert abs(float(mean) - float(np.mean(nll.astype(np.float64)))) <= 1e-6 * abs(float(mean))
    z, _ = m.inverse(x, None, y, [0.0], [0.0], [100], [2])
    _close_elem(z[idx], ref_z)
    x2 = m.forward(z, None, y, [0.0], [0.0], [100], [2])
    # Sampling direction on the same latents.  16 random wide CNN evaluations deep this stack is ill conditioned in
    # fp32: the oracle's own float32 flavour (same op order as the reference graph) is ~2e-5 of scale away from fp64.
    # The HIP path has to be at least as good as plain fp32 arithmetic, not better than it.
    ref64 = o.sample(z[idx], 1.0, y[idx], 100, 2)
    ref32 = NoiseFlowOracle(FULL_ARCH, v, dtype=np.float32).sample(z[idx], 1.0, y[idx], 100, 2)
    scale = np.abs(ref64).max()
    fp32_err = np.abs(ref32.astype(np.float64) - ref64).max()
    err = np.abs(x2[idx].astype(np.float64) - ref64).max()
    assert err <= max(1e-5 * scale, 2.0 * fp32_err) and err <= 1e-4 * scale, (err, fp32_err, scale)
    assert np.abs(x2 - x).max() <= 1e-4 * np.abs(x).max()          # round trip, same conditioning


def test_wide32_in_kernel_philox_matches_numpy_philox():
    from oracle import philox
    from oracle.nf_oracle import NoiseFlowOracle
    v = trained_like_variables(ARCH, 32, seed=21)
    H, W, B = 24, 40, 4
    _, y = make_inputs(B, H, W, seed=9)
    m = _model(ARCH, v, (H, W, 4), 32)
    xs = m.sample(y, 0.7, y, [0.0], [0.0], [100], [2], seed=77)
    eps = philox.sample_eps(77, 0, B, H, W)
    ref = NoiseFlowOracle(ARCH, v).sample(eps, 0.7, y, 100, 2)
    _close_elem(xs, ref, rtol=3e-5)   # Box-Muller on the hardware transcendental unit: ~1e-6 absolute on eps


@pytest.mark.parametrize("width,hw", [(32, (32, 32)), (32, (64, 64)), (32, (20, 28)), (32, (40, 50)), (16, (32, 32)), (8, (24, 24))])
def test_wide_fp16_cnn_mode(width, hw):
    """NF_CFG_FP16_CNN at widths 8 / 16 / 32 (v_mfma_f32_32x32x16_f16, fp32 accumulate, fp32 log-det): against the oracle's
    emulation of the rounding points — BN and exp(3 logs) folded in fp64, THEN the folded weights and the three CNN inputs
    rounded to half once — 1e-4 relative on the NLL, 2e-3 of scale on tensors (a near-tie at a half-rounding point may flip
    an activation by one fp16 ulp), and the mode stays within 2e-4 of the all-fp32 model on the NLL."""
    from noise_flow_amd import NoiseFlow, default_hps, _lib
    from oracle.nf_oracle import NoiseFlowOracle
    H, W = hw
    v = trained_like_variables(ARCH, width, seed=5 + width)
    for k in v:   # activations of O(1) at every width (the helper's weights are tuned for width 4)
        if k.endswith("l_2/W") or k.endswith("l_last/W"):
            v[k] = (v[k] * np.float32((4.0 / width) ** 0.5)).astype(np.float32)
    x, y = make_inputs(4, H, W, seed=12)
    m = NoiseFlow([H, W, 4], False, default_hps(arch=ARCH, width=width), variables=v, cnn_dtype="fp16")
    assert _path(m, 0) == _lib.NF_PATH_WIDE32_FP16 and _path(m, 1) == _lib.NF_PATH_WIDE32_FP16
    o16 = NoiseFlowOracle(ARCH, v, cnn_dtype="fp16")
    nll, sd = m._loss(x, y, [0.0], [0.0], [100], [2])
    ref, rsd, rz = o16.nll(x, y, 100, 2)
    np.testing.assert_allclose(nll, ref, rtol=1e-4)
    assert abs(sd - rsd) <= 1e-4 * rsd
    z, _ = m.inverse(x, None, y, [0.0], [0.0], [100], [2])
    _close_elem(z, rz, rtol=2e-3)
    eps = np.random.RandomState(3).randn(4, H, W, 4).astype(np.float32)
    _close_elem(m.sample(y, 0.6, y, [0.0], [0.0], [100], [2], eps=eps), o16.sample(eps, 0.6, y, 100, 2), rtol=2e-3)
    np.testing.assert_allclose(nll, NoiseFlowOracle(ARCH, v).nll(x, y, 100, 2)[0], rtol=2e-4)


@pytest.mark.parametrize("width,hw,dt", [(1, (8, 8), "fp32"), (3, (32, 32), "fp32"), (5, (32, 32), "fp32"), (6, (20, 28), "fp32"),
                                         (12, (32, 32), "fp32"), (24, (32, 32), "fp32"), (24, (64, 64), "fp32"), (31, (40, 50), "fp32"),
                                         (24, (32, 32), "fp16"), (5, (32, 32), "fp16"), (12, (100, 70), "fp32")])
def test_coupling_widths_between_the_kernel_widths(width, hw, dt):
    """layers.py:452-498 takes any width; the kernels exist for 4 / 8 / 16 / 32 (and 33 .. 512): a width in between runs on the
    next one up with the extra hidden channels zero-padded at nf_create (exact: relu(0) = 0 in both hidden layers).  Every
    kernel family the padding lands on, fp32 and the fp16-CNN mode, whole patches and a tiled image; the batch-statistics mode
    pads the same way (tests/test_gpu_batchstats.py), the trainer runs such widths on its library-GEMM path (tests/test_gpu_train.py)."""
    from noise_flow_amd import NoiseFlow, default_hps
    from oracle.nf_oracle import NoiseFlowOracle
    H, W = hw
    v = trained_like_variables(ARCH, width, seed=7 * width + H)
    x, y = make_inputs(3, H, W, seed=21)
    m = NoiseFlow([H, W, 4], False, default_hps(arch=ARCH, width=width), variables=v, cnn_dtype=dt)
    o = NoiseFlowOracle(ARCH, v, cnn_dtype=dt)
    nll, sd = m._loss(x, y, [0.0], [0.0], [100], [2])
    ref_nll, ref_sd, ref_z = o.nll(x, y, 100, 2)
    rt, et = (NLL_RTOL, ELEM_RTOL) if dt == "fp32" else (1e-4, 2e-3)
    np.testing.assert_allclose(nll, ref_nll, rtol=rt, atol=1e-4)
    z, _ = m.inverse(x, None, y, [0.0], [0.0], [100], [2])
    _close_elem(z, ref_z, rtol=et)
    eps = np.random.RandomState(4).randn(3, H, W, 4).astype(np.float32)
    xs = m.sample(y, 0.8, y, [0.0], [0.0], [100], [2], eps=eps)
    _close_elem(xs, o.sample(eps, 0.8, y, 100, 2), rtol=et)
    if dt == "fp32" and H * W <= 1024:      # batch-statistics mode pads the same way (its statistics passes at the padded width
        mt = NoiseFlow([H, W, 4], True, default_hps(arch=ARCH, width=width), variables=v)      # 32 hold up to 1024 pixels)
        nll_t, _ = mt._loss(x, y, [0.0], [0.0], [100], [2])
        np.testing.assert_allclose(nll_t, o.nll(x, y, 100, 2, training=True)[0], rtol=NLL_RTOL, atol=1e-4)
