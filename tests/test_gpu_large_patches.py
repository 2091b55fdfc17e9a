"""GPU parity for patches beyond 64x64: the overlapping-tile evaluation (csrc/nf_device.h "overlapping tiles",
`nf_tile_plan`) against the fp64 oracle on the WHOLE image.  The reference leaves --patch_height free
(sidd/ArgParser.py:72-73); a coupling is 3x3 -> 1x1 -> 3x3 (layers.py:452-498), so a pixel of the output depends on a
5x5 neighbourhood per coupling, and a tile that keeps 2 x couplings pixels of margin reproduces the image-wide result."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import FULL_ARCH, make_inputs, trained_like_variables

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _assert_ok(r):
    assert r["ok"], json.dumps(r)


@pytest.mark.parametrize("hw,B,arch", [((65, 64), 3, None), ((64, 65), 2, None), ((96, 80), 2, None), ((128, 128), 2, None),
                                       ((32, 200), 2, None), ((130, 70), 2, "sdn5|unc|unc|gain4|unc"), ((200, 9), 2, "unc|unc"),
                                       ((72, 72), 1, "sdn4|unc|gain4|unc|unc|unc|unc|unc|unc|unc|unc|unc|unc|unc")])
def test_large_patches_match_the_oracle(hw, B, arch):
    """NLL, sd_z, log-det, latent, round trip, sampling with supplied and in-kernel eps; host-fed == resident.  The last case
    has 12 couplings (halo 24: a 64-pixel tile reports a 16-pixel core)."""
    from check_large_patches import check
    _assert_ok(check(hw[0], hw[1], B, arch))


@pytest.mark.parametrize("segments", [1, 2, 3, 8])
def test_large_patches_whatever_the_segment_count(monkeypatch, segments):
    """The program cut into 1 / 2 / 3 / 8 tiled launches (nf_tile_segments; the library picks the count by estimated work,
    NF_TILE_SEGMENTS forces it): same results against the oracle, ragged last segment included (8 couplings in 3 segments)."""
    from check_large_patches import check
    monkeypatch.setenv("NF_TILE_SEGMENTS", str(segments))
    r = check(130, 150, 2, FULL_ARCH, seed=segments)
    assert r["segments"] == segments
    _assert_ok(r)


def test_large_patches_deep_stack():
    """16 couplings: one launch would need a halo of 32 — no core in a 64-pixel tile; the segment plan splits it."""
    from check_large_patches import check
    r = check(100, 90, 1, "|".join(["unc"] * 16))
    assert r["segments"] >= 2
    _assert_ok(r)


def test_large_patches_with_the_shipped_checkpoint(shipped_variables):
    """The model that ships (S6 checkpoint, 8 couplings: halo 16) on 128x96 images, every tolerance as in
    tests/test_gpu_parity.py — no conditioning allowance."""
    from check_large_patches import check
    _assert_ok(check(128, 96, 3, FULL_ARCH, variables=shipped_variables, strict=True))


def test_large_patches_on_the_scalar_weight_kernel():
    """NF_KERNEL=valu is read once per process: the same check in a child process."""
    env = dict(os.environ, NF_KERNEL="valu")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_large_patches.py"), "96", "100", "2"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["ok"] and r["kernel_path"] == 0, r


def test_tile_results_do_not_depend_on_the_batch():
    """Per-image results are bit-identical whatever else shares the launch (tile index arithmetic, per-image sums added up
    in tile order by nf_tile_combine_kernel)."""
    from noise_flow_amd import NoiseFlow, default_hps
    v = trained_like_variables(FULL_ARCH, 4, seed=5)
    x, y = make_inputs(5, 96, 64, seed=3)
    m = NoiseFlow([96, 64, 4], False, default_hps(arch=FULL_ARCH, width=4), variables=v)
    nll, _ = m._loss(x, y, [0.0], [0.0], [100], [2])
    for i in range(5):
        one, _ = m._loss(x[i:i + 1], y[i:i + 1], [0.0], [0.0], [100], [2])
        assert np.array_equal(one, nll[i:i + 1])
    z, _ = m.inverse(x, None, y, [0.0], [0.0], [100], [2])
    z3, _ = m.inverse(x[3:4], None, y[3:4], [0.0], [0.0], [100], [2])
    assert np.array_equal(z3[0], z[3])


@pytest.mark.parametrize("width,cnn_dtype,hw,arch,path", [(32, "fp32", (100, 72), "sdn5|unc|gain4|unc|unc", 3),
                                                          (16, "fp32", (70, 130), "sdn5|unc|unc|gain4|unc", 4),      # width 16: its own kernel takes tiles
                                                          (8, "fp32", (65, 65), "unc|unc", 3),
                                                          (32, "fp16", (96, 96), "sdn5|unc|gain4|unc", 5),
                                                          (4, "fp16", (128, 80), None, 2),      # full 64x64 tiles: the width-4 fp16 kernel
                                                          (4, "fp16", (256, 96), "sdn5|unc|unc|gain4|unc|unc", 2),
                                                          (4, "fp16", (150, 40), None, 5),      # 64x40 tiles: zero-padded on the width-32 kernel
                                                          (32, "fp32", (40, 150), "unc|unc|unc", 3),
                                                          # widths beyond 32: the GEMM kernels take tiles as they take patches
                                                          (64, "fp32", (80, 100), "sdn5|unc|gain4|unc", 6),      # variant B (weights in LDS)
                                                          (200, "fp32", (70, 66), "unc|unc", 6),
                                                          (96, "fp16", (130, 64), "sdn5|unc|unc|gain4", 7),
                                                          (512, "fp16", (65, 40), "unc", 7)])
def test_large_patches_on_the_width_32_kernel(width, cnn_dtype, hw, arch, path):
    """Coupling widths 8 / 16 / 32 and the fp16-CNN mode (any width up to 32) run their tiles on the width-32 matrix-core
    kernel (narrower CNNs zero-padded: exact, a padded channel is identically zero) — except width 4 in fp16-CNN mode on images
    of at least 64 pixels per side, whose full 64x64 tiles run on the width-4 fp16 kernel (v_mfma_f32_16x16x32_f16).  Widths
    33 .. 512 run their tiles on the GEMM kernels (nf_gemm.hip / nf_gemm16.hip: a tile is a patch at an offset, border masks
    from the image, results for the core window)."""
    from check_large_patches import check
    v = None
    if width > 32:      # activations of O(1) at every width (the helper's weights are tuned for width 4): the fp16 cases compare
        v = trained_like_variables(arch, width, seed=hw[0] * 1000 + hw[1])      # tensors at 2e-3 of their scale
        for k in v:
            if k.endswith("l_2/W") or k.endswith("l_last/W"):
                v[k] = (v[k] * np.float32((4.0 / width) ** 0.5)).astype(np.float32)
    r = check(hw[0], hw[1], 2, arch, width=width, cnn_dtype=cnn_dtype, variables=v)
    assert r["kernel_path"] == path, r
    _assert_ok(r)


def test_tiled_calls_from_concurrent_streams_share_one_handle(shipped_variables):
    """Scratch of a tiled call (per-tile sums, the tensor between two segments) is a buffer of the handle's workspace set that
    no other call in flight uses: 8 threads on their own streams, one handle, 256x256 images in two segments."""
    import threading
    import torch
    from noise_flow_amd import NoiseFlow, default_hps
    m = NoiseFlow([256, 256, 4], False, default_hps(arch=FULL_ARCH, width=4), variables=shipped_variables)
    x, y = make_inputs(3, 256, 256, seed=8)
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    ref, _ = m._loss(xt, yt, [0], [0], [100], [2])
    zref, _ = m.inverse(xt, None, yt, [0], [0], [100], [2])
    ref, zref = ref.cpu().numpy(), zref.cpu().numpy()
    torch.cuda.synchronize()
    errs = []

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(6):
                    nll, _ = m._loss(xt, yt, [0], [0], [100], [2])
                    z, _ = m.inverse(xt, None, yt, [0], [0], [100], [2])
                st.synchronize()
            assert np.array_equal(nll.cpu().numpy(), ref) and np.array_equal(z.cpu().numpy(), zref)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs[0]


def test_tiled_workspace_is_owned_by_the_handle_and_reservable(shipped_variables):
    """no allocation inside nf_nll / nf_sample.  Patches up to 64x64 need no scratch (nf_workspace_bytes = 0);
    beyond that nf_workspace_bytes says what a call of B images needs, nf_reserve_workspace sets it aside for n calls in
    flight, and calls after that run with the device's free memory unchanged (no hipMalloc of any kind in them)."""
    import torch
    from noise_flow_amd import NoiseFlow, default_hps
    small = NoiseFlow([32, 32, 4], False, default_hps(), variables=shipped_variables)
    assert small._flow.lib.nf_workspace_bytes(small._flow.ptr, 0, 1024) == 0
    m = NoiseFlow([192, 160, 4], False, default_hps(arch=FULL_ARCH, width=4), variables=shipped_variables)
    lib, B = m._flow.lib, 4
    need = lib.nf_workspace_bytes(m._flow.ptr, 0, B)
    assert need >= B * 192 * 160 * 16 and lib.nf_workspace_bytes(m._flow.ptr, 1, B) <= need      # >= one tensor between two segments
    from noise_flow_amd import _lib
    _lib.check(lib.nf_reserve_workspace(m._flow.ptr, B, 2))
    x, y = make_inputs(B, 192, 160, seed=4)
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    nll0, _ = m._loss(xt, yt, [0], [0], [100], [2])          # also warms torch's own caching allocator for the outputs
    z0, _ = m.inverse(xt, None, yt, [0], [0], [100], [2])
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(5):
        nll, _ = m._loss(xt, yt, [0], [0], [100], [2])
        z, _ = m.inverse(xt, None, yt, [0], [0], [100], [2])
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    assert torch.equal(nll, nll0) and torch.equal(z, z0)
    with pytest.raises(Exception):
        _lib.check(lib.nf_reserve_workspace(m._flow.ptr, B, 0))


@pytest.mark.parametrize("compat", [None, "reference"])
def test_wrapper_samples_large_patches(compat):
    """`NoiseFlowWrapper(..., patch_shape=(H, W))`: the shipped model on 96x80 clean patches, in the trained model's semantics
    and in the upstream wrapper's literal mode (sampling-order binding, is_training=True: batch statistics on tiles)."""
    from noise_flow_amd import NoiseFlowWrapper
    from noise_flow_amd.ckpt import load_checkpoint
    from oracle import philox
    from oracle.nf_oracle import NoiseFlowOracle
    from conftest import SHIPPED_CKPT, SHIPPED_DIR
    nf = NoiseFlowWrapper(SHIPPED_DIR, sampling_temperature=0.6, seed=11, compat=compat, patch_shape=(96, 80))
    _, y = make_inputs(3, 96, 80, seed=12)
    xs = nf.sample_noise_nf(y, 0.0, 0.0, 800, 2)
    assert xs.shape == (3, 96, 80, 4) and xs.dtype == np.float32
    o = NoiseFlowOracle(FULL_ARCH, load_checkpoint(SHIPPED_CKPT), "sample_first" if compat else "loss_first")
    eps = philox.sample_eps(11, 0, 3, 96, 80)
    ref = o.sample(eps, 0.6, y, 800, 2, training=bool(compat))
    assert np.abs(xs - ref).max() <= 5e-5 * np.abs(ref).max()


def test_batch_statistics_beyond_64x64_at_other_widths():
    """Batch-statistics mode beyond 64x64 used to exist on the width-4 matrix-core schedule only (NF_EINVAL elsewhere); since round
    5 every other width walks the layers on the trainer's GEMM path over the whole resident image (csrc/nf_train.hip:
    nf_bs_wide_run) — no tiles, no halo: width 8 on 80x80 against the fp64 oracle."""
    from noise_flow_amd import NoiseFlow, default_hps
    from oracle.nf_oracle import NoiseFlowOracle
    v8 = trained_like_variables("unc|unc", 8)
    m = NoiseFlow([80, 80, 4], True, default_hps(arch="unc|unc", width=8), variables=v8)
    x, y = make_inputs(2, 80, 80, seed=1)
    nll, sd = m._loss(x, y, [0.0], [0.0], [100], [2])
    ref, ref_sd, _ = NoiseFlowOracle("unc|unc", v8).nll(x, y, 100, 2, training=True)
    np.testing.assert_allclose(nll, ref, rtol=1e-5)
    assert abs(sd - ref_sd) <= 1e-5 * ref_sd


@pytest.mark.parametrize("hw,B,arch", [((96, 80), 3, None), ((65, 130), 2, "sdn5|unc|gain4|unc|unc"), ((128, 40), 4, "unc|unc|unc")])
def test_batch_statistics_mode_on_large_patches(shipped_variables, hw, B, arch):
    """`is_training=True` graphs (layers.py:386-398; what the upstream wrapper feeds, quirk Q2) beyond 64x64: every launch of the
    batch-statistics schedule tiled with one halo, the moments taken over the core windows (= every pixel once) — NLL, sd_z,
    latent, the EMA'd running statistics and sampling against the oracle in training mode on the whole images."""
    from noise_flow_amd import NoiseFlow, default_hps
    from oracle.nf_oracle import NoiseFlowOracle
    H, W = hw
    arch = arch or FULL_ARCH
    v = shipped_variables if arch == FULL_ARCH else trained_like_variables(arch, 4, seed=H + W)
    x, y = make_inputs(B, H, W, seed=31, b1=0.003696)
    o = NoiseFlowOracle(arch, v, "loss_first")
    ref_nll, ref_sd, ref_z = o.nll(x, y, 800, 2, training=True)
    m = NoiseFlow([H, W, 4], True, default_hps(arch=arch, width=4), variables=v)
    nll, sd = m._loss(x, y, [0.0], [0.0], [800], [2])
    np.testing.assert_allclose(nll, ref_nll, rtol=1e-5)
    assert abs(sd - ref_sd) <= 1e-5 * ref_sd
    names = [L["name"] for L in o.layers if L["type"] == "coupling"]
    for scope, lname in zip(m._flow.coupling_scopes, names):
        rec = o.last_batch_moments[lname]
        for bn, key in (("bn_nvp_conv_1/mean", "new_mean1"), ("bn_nvp_conv_1/var", "new_var1"),
                        ("bn_nvp_conv_2/mean", "new_mean2"), ("bn_nvp_conv_2/var", "new_var2")):
            want = rec[key]
            assert np.abs(m.variables[scope + "/" + bn] - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-3), (scope, bn)
    m2 = NoiseFlow([H, W, 4], True, default_hps(arch=arch, width=4), variables=v)
    z, _ = m2.inverse(x, None, y, [0.0], [0.0], [800], [2])
    assert np.abs(z - ref_z).max() <= 1e-5 * np.abs(ref_z).max()
    eps = np.random.RandomState(3).randn(B, H, W, 4).astype(np.float32)
    m3 = NoiseFlow([H, W, 4], True, default_hps(arch=arch, width=4), variables=v)
    xs = m3.sample(y, 0.7, y, [0.0], [0.0], [800], [2], eps=eps)
    ref_x = o.sample(eps, 0.7, y, 800, 2, training=True)
    assert np.abs(xs - ref_x).max() <= 2e-5 * np.abs(ref_x).max()
