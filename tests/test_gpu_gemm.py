"""GPU parity of WIDE coupling CNNs (hps.width 33 .. 512; sidd/ArgParser.py:43 defaults --width to 512) on the LDS-staged
GEMM kernel (csrc/nf_gemm.hip: v_mfma_f32_32x32x2_f32, the hidden activations of a band of 32768 / width pixels in LDS, l_2's
weights streamed from L2 by the wavefront that consumes them) against the fp64 oracle: widths that are and are not one of the
padded sizes, full and ragged patch shapes, both directions, in-kernel Philox, batches beyond the resident grid.
Tolerances as everywhere: per-patch NLL 1e-5 relative, tensors 1e-5 of their scale."""
import numpy as np
import pytest

from conftest import make_inputs, trained_like_variables

pytestmark = pytest.mark.gpu

NLL_RTOL = 1e-5
ELEM_RTOL = 1e-5
ARCH = "sdn5|unc|gain4|unc"


def _variables(arch, width, seed):
    v = trained_like_variables(arch, width, seed=seed)
    for k in v:   # activations of O(1) at every width (the helper's weights are tuned for width 4)
        if k.endswith("l_2/W") or k.endswith("l_last/W"):
            v[k] = (v[k] * np.float32((4.0 / width) ** 0.5)).astype(np.float32)
    return v


def _model(arch, variables, x_shape, width):
    from noise_flow_amd import NoiseFlow, default_hps
    return NoiseFlow(list(x_shape), False, default_hps(arch=arch, width=width), variables=variables)


def _close_elem(a, ref, rtol=ELEM_RTOL):
    """Scale-relative bound + the masked per-element relative error for the session summary (tests/conftest.py::close_elem)."""
    from conftest import close_elem
    return close_elem(a, ref, rtol)


def _path(m, direction=0):
    return m._flow.lib.nf_kernel_path(m._flow.ptr, direction)


@pytest.mark.parametrize("width,hw", [(64, (32, 32)), (128, (32, 32)), (256, (32, 32)), (512, (32, 32)),
                                      (64, (20, 28)), (512, (24, 40)), (96, (32, 32)), (48, (16, 16)), (200, (7, 5)),
                                      (64, (1, 1)), (128, (33, 31)), (64, (32, 64)), (512, (45, 45)), (160, (9, 33)),
                                      # up to 64x64 (8 pixels per thread; the configs[4] geometry at the reference's default width): the
                                      # slab kernel (64), the band kernel with the bare pass-through tile (128, 512), ragged shapes
                                      (64, (64, 64)), (128, (64, 64)), (512, (64, 64)), (256, (56, 63)), (96, (64, 40))])
def test_gemm_kernel_matches_oracle(width, hw):
    from noise_flow_amd import _lib
    from oracle.nf_oracle import NoiseFlowOracle
    H, W = hw
    B = 3 if H * W <= 2048 else 2
    v = _variables(ARCH, width, seed=H * 100 + W + width)
    x, y = make_inputs(B, H, W, seed=3)
    m = _model(ARCH, v, (H, W, 4), width)
    assert _path(m, 0) == _lib.NF_PATH_GEMM and _path(m, 1) == _lib.NF_PATH_GEMM
    o = NoiseFlowOracle(ARCH, v)
    nll, sd = m._loss(x, y, [0.0], [0.0], [100], [2])
    ref_nll, ref_sd, ref_z = o.nll(x, y, 100, 2)
    np.testing.assert_allclose(nll, ref_nll, rtol=NLL_RTOL, atol=1e-4)
    assert abs(sd - ref_sd) <= 1e-5 * ref_sd
    z, obj = m.inverse(x, None, y, [0.0], [0.0], [100], [2])
    _close_elem(z, ref_z)
    eps = np.random.RandomState(4).randn(B, H, W, 4).astype(np.float32)
    xs = m.sample(y, 0.8, y, [0.0], [0.0], [100], [2], eps=eps)
    _close_elem(xs, o.sample(eps, 0.8, y, 100, 2))


def test_gemm_full_arch_batch_beyond_the_grid_and_round_trip():
    """The shipped layer sequence at width 64, more patches than resident workgroups (persistent stride loop), slotted sums,
    sample(nll(x)) = x, in-kernel Philox against numpy Philox."""
    from conftest import FULL_ARCH
    from oracle import philox
    from oracle.nf_oracle import NoiseFlowOracle
    v = _variables(FULL_ARCH, 64, seed=11)
    B = 600
    x, y = make_inputs(B, 32, 32, seed=5)
    m = _model(FULL_ARCH, v, (32, 32, 4), 64)
    nll, sd = m._loss(x, y, [0.0], [0.0], [100], [2])
    idx = np.r_[0:4, B - 4:B]
    o = NoiseFlowOracle(FULL_ARCH, v)
    ref_nll, _, ref_z = o.nll(x[idx], y[idx], 100, 2)
    np.testing.assert_allclose(nll[idx], ref_nll, rtol=NLL_RTOL, atol=1e-4)
    mean, _ = m.loss(x, y, [0.0], [0.0], [100], [2])
    assert abs(float(mean) - float(np.mean(nll.astype(np.float64)))) <= 1e-6 * abs(float(mean))
    z, _ = m.inverse(x, None, y, [0.0], [0.0], [100], [2])
    _close_elem(z[idx], ref_z)
    x2 = m.forward(z, None, y, [0.0], [0.0], [100], [2])
    assert np.abs(x2 - x).max() <= 1e-4 * np.abs(x).max()
    xs = m.sample(y[:4], 0.7, y[:4], [0.0], [0.0], [100], [2], seed=77)
    ref = o.sample(philox.sample_eps(77, 0, 4, 32, 32), 0.7, y[:4], 100, 2)
    _close_elem(xs, ref, rtol=3e-5)   # Box-Muller on the hardware transcendental unit: ~1e-6 absolute on eps


def test_gemm_width_512_full_arch():
    """Glow's default width (sidd/ArgParser.py:43) through the whole shipped layer sequence: 8 couplings of 290 kMAC/pixel."""
    from conftest import FULL_ARCH
    from oracle.nf_oracle import NoiseFlowOracle
    v = _variables(FULL_ARCH, 512, seed=3)
    x, y = make_inputs(2, 32, 32, seed=6)
    m = _model(FULL_ARCH, v, (32, 32, 4), 512)
    o = NoiseFlowOracle(FULL_ARCH, v)
    nll, sd = m._loss(x, y, [0.0], [0.0], [800], [1])
    ref_nll, ref_sd, ref_z = o.nll(x, y, 800, 1)
    np.testing.assert_allclose(nll, ref_nll, rtol=NLL_RTOL, atol=1e-4)
    z, _ = m.inverse(x, None, y, [0.0], [0.0], [800], [1])
    _close_elem(z, ref_z)


def test_gemm_width_limits_are_reported():
    from noise_flow_amd import NoiseFlow, default_hps
    for width, hw, dt in ((513, (32, 32), "fp32"), (600, (64, 96), "fp16")):      # (images beyond 64x64 run as tiles at every width)
        v = trained_like_variables(ARCH, width, seed=1)
        with pytest.raises(Exception) as ei:
            NoiseFlow([hw[0], hw[1], 4], False, default_hps(arch=ARCH, width=width), variables=v, cnn_dtype=dt)
        assert "width" in str(ei.value)


@pytest.mark.parametrize("width,hw", [(64, (32, 32)), (128, (32, 32)), (256, (32, 32)), (512, (32, 32)), (96, (20, 28)), (512, (24, 40)),
                                      (64, (45, 45)), (200, (7, 5)), (160, (33, 31)),
                                      (64, (64, 64)), (512, (64, 64)), (256, (60, 64))])     # BASELINE configs[4]'s geometry
def test_gemm_fp16_cnn_mode(width, hw):
    """NF_CFG_FP16_CNN at widths 33 .. 512 (csrc/nf_gemm16.hip: v_mfma_f32_32x32x16_f16, fp32 accumulate, fp32 log-det): against
    the oracle's emulation of the rounding points — BN and exp(3 logs) folded in fp64, THEN the folded weights and the three
    CNN inputs rounded to half once — 1e-4 relative on the NLL, 2e-3 of scale on tensors (a near-tie at a half-rounding point
    may flip an activation by one fp16 ulp), and the mode stays within 2e-4 of the all-fp32 model on the NLL
    (the tolerances of tests/test_gpu_wide.py::test_wide_fp16_cnn_mode)."""
    from noise_flow_amd import NoiseFlow, default_hps, _lib
    from oracle.nf_oracle import NoiseFlowOracle
    H, W = hw
    B = 3
    v = _variables(ARCH, width, seed=7 + width + H)
    x, y = make_inputs(B, H, W, seed=12)
    m = NoiseFlow([H, W, 4], False, default_hps(arch=ARCH, width=width), variables=v, cnn_dtype="fp16")
    assert _path(m, 0) == _lib.NF_PATH_GEMM_FP16 and _path(m, 1) == _lib.NF_PATH_GEMM_FP16
    o16 = NoiseFlowOracle(ARCH, v, cnn_dtype="fp16")
    nll, sd = m._loss(x, y, [0.0], [0.0], [100], [2])
    ref, rsd, rz = o16.nll(x, y, 100, 2)
    np.testing.assert_allclose(nll, ref, rtol=1e-4)
    assert abs(sd - rsd) <= 1e-4 * rsd
    z, _ = m.inverse(x, None, y, [0.0], [0.0], [100], [2])
    _close_elem(z, rz, rtol=2e-3)
    eps = np.random.RandomState(3).randn(B, H, W, 4).astype(np.float32)
    _close_elem(m.sample(y, 0.6, y, [0.0], [0.0], [100], [2], eps=eps), o16.sample(eps, 0.6, y, 100, 2), rtol=2e-3)
    np.testing.assert_allclose(nll, NoiseFlowOracle(ARCH, v).nll(x, y, 100, 2)[0], rtol=2e-4)
