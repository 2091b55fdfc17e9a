"""CPU tier: the C-ABI library loads, exports every symbol include/*.h declares, and
its host-side folding (PLU -> A/A^-1, BN-eval, edge table, exp(3 logs), gain,
sdn5 scalars) agrees with the numpy oracle.  No compute calls without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import FULL_ARCH, ROOT, trained_like_variables
from oracle import nf_oracle as O


def _fold(arch, variables, width, direction, hw=(32, 32), flow_permutation=1, decomp="LU"):
    from noise_flow_amd import _lib, params
    lib = _lib.load()
    layers, descs, flat = params.pack(arch, variables, width, "loss_first", flow_permutation, decomp)
    cfg = _lib.nf_config(hw[0], hw[1], 4, len(layers), -1, 0)
    ops = (C.c_int32 * 256)()
    n_ops, nf, ld = C.c_int32(), C.c_size_t(), C.c_double()
    folded = np.zeros(1 << 16, np.float32)
    rc = lib.nf_fold_params(C.byref(cfg), descs, flat.ctypes.data_as(C.POINTER(C.c_float)), flat.size, direction,
                            ops, 128, C.byref(n_ops), folded.ctypes.data_as(C.POINTER(C.c_float)), folded.size,
                            C.byref(nf), C.byref(ld))
    _lib.check(rc)
    return [(ops[2 * i], ops[2 * i + 1]) for i in range(n_ops.value)], folded[:nf.value].copy(), ld.value


def test_header_symbols_are_exported():
    from noise_flow_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "noiseflow_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nf_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.nf_abi_version() == 1
    # the product has no CPU fallback: the python surface refuses to run without a GPU
    import torch
    if not torch.cuda.is_available():
        from noise_flow_amd import NoiseFlow, default_hps
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            NoiseFlow([32, 32, 4], False, default_hps())


def test_layer_param_counts():
    from noise_flow_amd import _lib
    lib = _lib.load()
    assert lib.nf_layer_param_count(_lib.NF_LAYER_CONV1X1, 0) == 36
    assert lib.nf_layer_param_count(_lib.NF_LAYER_COUPLING, 4) == 72 + 4 + 8 + 16 + 4 + 8 + 180 + 4 + 4 + 1
    assert lib.nf_layer_param_count(_lib.NF_LAYER_SDN5, 0) == 23
    assert lib.nf_layer_param_count(_lib.NF_LAYER_GAIN4, 0) == 1
    assert lib.nf_layer_param_count(_lib.NF_LAYER_SDN4, 0) == 7
    assert lib.nf_layer_param_count(_lib.NF_LAYER_SDN, 0) == 2 and lib.nf_layer_param_count(_lib.NF_LAYER_GAIN, 0) == 2
    assert lib.nf_layer_param_count(99, 0) < 0
    # trainable parameter count of the shipped arch from the ABI's own layout:
    # 8*(36-20) + 8*(301-16) + 10 rescaling... cross-checked against hps.txt in test_ckpt_hps


@pytest.mark.parametrize("width", [4, 8, 16, 32])
def test_folding_matches_oracle(shipped_variables, width):
    from noise_flow_amd import _lib
    arch = FULL_ARCH if width == 4 else "unc|gain4|unc|sdn5"
    v = shipped_variables if width == 4 else trained_like_variables(arch, width, seed=width)
    layers = O.bind_variables(arch, v)
    ops, blk, ld = _fold(arch, v, width, 0)
    ops_r, blk_r, ld_r = _fold(arch, v, width, 1)
    # constant log-det: H*W*sum log_S - H*W*C*log(gain)
    want_ld = sum(1024 * L["log_abs_det"] for L in layers if L["type"] == "conv1x1")
    gains = [float(np.asarray(L["gain_val"]).reshape(-1)[0]) for L in layers if L["type"] == "gain4"]
    want_ld -= sum(4096 * np.log(g) for g in gains)
    assert abs(ld - want_ld) < 1e-9 and ld == ld_r
    # op sequence: gain folded into the following 1x1 matrix
    kinds = [L["type"] for L in layers if L["type"] != "gain4"]
    code = {"conv1x1": _lib.NF_OP_MIX, "coupling": _lib.NF_OP_COUPLING_FWD, "sdn5": _lib.NF_OP_SDN_DIV}
    assert [t for t, _ in ops] == [code[k] for k in kinds]
    rcode = {"conv1x1": _lib.NF_OP_MIX, "coupling": _lib.NF_OP_COUPLING_REV, "sdn5": _lib.NF_OP_SDN_MUL}
    assert [t for t, _ in ops_r] == [rcode[k] for k in kinds[::-1]]
    assert all(off % 4 == 0 for _, off in ops + ops_r)

    w = width
    kept = [L for L in layers if L["type"] != "gain4"]
    gain_before = {}
    g = 1.0
    for L in layers:
        if L["type"] == "gain4":
            g = float(np.asarray(L["gain_val"]).reshape(-1)[0])
        elif L["type"] == "conv1x1":
            gain_before[L["name"]] = g
            g = 1.0
    for (t, off), L in zip(ops, kept):
        if L["type"] == "conv1x1":
            np.testing.assert_allclose(blk[off:off + 16].reshape(4, 4), L["A"] / gain_before[L["name"]], rtol=2e-7, atol=1e-9)
        elif L["type"] == "coupling":
            p = L["p"]
            s1 = 1 / np.sqrt(p["bn1/var"] + 1e-4)
            s2 = 1 / np.sqrt(p["bn2/var"] + 1e-4)
            es = np.exp(3 * p["l_last/logs"])
            b = blk[off:]
            W3 = b[64:64 + 36 * w].reshape(9, w, 4)
            np.testing.assert_allclose(W3, p["l_last/W"].reshape(9, w + 1, 4)[:, :w] * es, rtol=2e-7, atol=1e-12)
            o1 = 64 + 36 * w
            np.testing.assert_allclose(b[o1:o1 + 18 * w].reshape(9, 2, w), p["l_1/W"].reshape(9, 2, w) * s1, rtol=2e-7, atol=1e-12)
            np.testing.assert_allclose(b[o1 + 18 * w:o1 + 19 * w], (p["l_1/b"] - p["bn1/mean"]) * s1, rtol=2e-7, atol=1e-9)
            o2 = o1 + 19 * w
            np.testing.assert_allclose(b[o2:o2 + w * w].reshape(w, w), p["l_2/W"].reshape(w, w) * s2, rtol=2e-7, atol=1e-12)
            np.testing.assert_allclose(b[o2 + w * w:o2 + w * w + w], (p["l_2/b"] - p["bn2/mean"]) * s2, rtol=2e-7, atol=1e-9)
            assert b[o2 + w * w + w] == np.float32(p["rescaling_scale"])
            # border table: run the oracle's padded conv on an all-zero hidden map -> pure bias + edge taps
            zero = np.zeros((1, 3, 3, w))
            e = O.conv2d_nhwc(O.add_edge_padding(zero), p["l_last/W"], False) + p["l_last/b"]
            e = e * es
            E = b[:64].reshape(16, 4)
            for r in range(3):
                for c in range(3):
                    mask = (1 if r == 0 else 0) | (2 if r == 2 else 0) | (4 if c == 0 else 0) | (8 if c == 2 else 0)
                    np.testing.assert_allclose(E[mask], e[0, r, c], rtol=3e-7, atol=1e-9)
            # degenerate 1x1 patch: every outer tap is outside
            e1 = (O.conv2d_nhwc(O.add_edge_padding(np.zeros((1, 1, 1, w))), p["l_last/W"], False) + p["l_last/b"]) * es
            np.testing.assert_allclose(E[15], e1[0, 0, 0], rtol=3e-7, atol=1e-9)
    for (t, off), L in zip(ops_r, kept[::-1]):
        if L["type"] == "conv1x1":
            np.testing.assert_allclose(blk_r[off:off + 16].reshape(4, 4), L["A_inv"] * gain_before[L["name"]], rtol=2e-7, atol=1e-9)


@pytest.mark.parametrize("width,padded", [(1, 4), (3, 4), (5, 8), (12, 16), (24, 32), (31, 32)])
def test_in_between_widths_fold_zero_padded_onto_the_next_kernel_width(width, padded):
    """layers.py:452-498 takes any width; the kernels exist for 4 / 8 / 16 / 32: a width in between is folded into the next
    layout up with zero weights / biases for the extra hidden channels (relu(0) = 0 in both hidden layers: exact).  The
    padded block must equal the fold of the same model with its variables zero-extended to the padded width."""
    arch = "unc|gain4|unc|sdn5"
    v = trained_like_variables(arch, width, seed=3 * width)
    ops, blk, ld = _fold(arch, v, width, 0)
    vp = {}
    for k, a in v.items():
        a = np.asarray(a)
        if k.endswith("l_1/W"):
            b = np.zeros(a.shape[:3] + (padded,), np.float32); b[..., :width] = a
        elif k.endswith("l_2/W"):
            b = np.zeros(a.shape[:2] + (padded, padded), np.float32); b[..., :width, :width] = a
        elif k.endswith("l_last/W"):
            b = np.zeros(a.shape[:2] + (padded + 1, 4), np.float32); b[:, :, :width] = a[:, :, :width]; b[:, :, padded] = a[:, :, width]
        elif (k.endswith("l_1/b") or k.endswith("l_2/b") or k.endswith("/mean")) and a.shape[-1] == width:
            b = np.zeros(padded, np.float32); b[:width] = a
        elif k.endswith("/var") and a.shape[-1] == width:
            b = np.ones(padded, np.float32); b[:width] = a
        else:
            b = a
        vp[k] = b
    ops_p, blk_p, ld_p = _fold(arch, vp, padded, 0)
    assert ops == ops_p and ld == ld_p and blk.shape == blk_p.shape
    np.testing.assert_array_equal(blk, blk_p)


def test_secondary_layers_fold_to_conditional_slots():
    from noise_flow_amd import _lib
    v = trained_like_variables("sdn4|unc|gain|sdn", 4)
    ops, blk, ld = _fold("sdn4|unc|gain|sdn", v, 4, 0)
    assert [t for t, _ in ops] == [_lib.NF_OP_SDN_DIV, _lib.NF_OP_MIX, _lib.NF_OP_COUPLING_FWD, _lib.NF_OP_SCALE_COND,
                                   _lib.NF_OP_SDN_DIV]
    assert [off for t, off in ops if t in (_lib.NF_OP_SDN_DIV, _lib.NF_OP_SCALE_COND)] == [0, 1, 2]   # slots
    ops, _, _ = _fold("sdn4|unc|gain|sdn", v, 4, 1)
    assert [t for t, _ in ops] == [_lib.NF_OP_SDN_MUL, _lib.NF_OP_SCALE_COND, _lib.NF_OP_COUPLING_REV, _lib.NF_OP_MIX,
                                   _lib.NF_OP_SDN_MUL]
    with pytest.raises(_lib.NoiseFlowLibError):
        _fold("sdn|sdn4|sdn5|gain|sdn4", v | trained_like_variables("sdn5", 4), 4, 0)          # > 4 conditional layers


def test_lone_gain_becomes_scale_op():
    from noise_flow_amd import _lib
    v = trained_like_variables("gain4|sdn5", 4)
    ops, blk, ld = _fold("gain4|sdn5", v, 4, 0)
    assert [t for t, _ in ops] == [_lib.NF_OP_SCALE, _lib.NF_OP_SDN_DIV]
    assert abs(blk[ops[0][1]] - 1 / 1.3) < 1e-7
    ops, blk, _ = _fold("gain4|sdn5", v, 4, 1)
    assert [t for t, _ in ops] == [_lib.NF_OP_SDN_MUL, _lib.NF_OP_SCALE]
    assert abs(blk[ops[1][1]] - 1.3) < 1e-7


@pytest.mark.parametrize("iso,cam", [(100, 2), (400, 0), (800, 4), (1600, 1), (3200, 3), (250, 2)])
def test_sdn5_scalars_match_oracle(shipped_variables, iso, cam):
    from noise_flow_amd import _lib
    lib = _lib.load()
    sp = np.concatenate([shipped_variables["model/sdn_gain/" + k].reshape(-1) for k in
                         ("beta1", "beta2", "gain_params", "cam_params")] + [np.ones(1, np.float32)]).astype(np.float32)
    out = (C.c_double * 2)()
    cond = _lib.nf_cond(iso, cam, 0, 0)
    _lib.check(lib.nf_sdn5_scalars(sp.ctypes.data_as(C.POINTER(C.c_float)), C.byref(cond), out))
    L = O.bind_variables(FULL_ARCH, shipped_variables)[0]
    b1, b2, gain = O.sdn_ex5_scalars(L["p"], iso, cam)
    assert abs(out[0] - b1 / gain) <= 1e-12 * abs(b1 / gain) and abs(out[1] - b2) <= 1e-12 * b2


def test_error_reporting(shipped_variables):
    from noise_flow_amd import _lib, params
    lib = _lib.load()
    sp = np.zeros(23, np.float32)
    out = (C.c_double * 2)()
    rc = lib.nf_sdn5_scalars(sp.ctypes.data_as(C.POINTER(C.c_float)), C.byref(_lib.nf_cond(100, 5, 0, 0)), out)
    assert rc == _lib.NF_ECOND and b"camera" in lib.nf_last_error()
    layers, descs, flat = params.pack(FULL_ARCH, shipped_variables, 4)
    n_ops = C.c_int32()

    def fold(cfg, n=flat.size):
        return lib.nf_fold_params(C.byref(cfg), descs, flat.ctypes.data_as(C.POINTER(C.c_float)), n, 0, None, 0,
                                  C.byref(n_ops), None, 0, None, None)
    assert fold(_lib.nf_config(32, 32, 3, len(layers), -1, 0)) == _lib.NF_EINVAL and b"channels" in lib.nf_last_error()
    assert fold(_lib.nf_config(5000, 128, 4, len(layers), -1, 0)) == _lib.NF_EINVAL and b"per side" in lib.nf_last_error()
    assert fold(_lib.nf_config(128, 128, 4, len(layers), -1, 0)) == 0          # beyond 64x64: overlapping tiles
    assert fold(_lib.nf_config(128, 128, 4, len(layers), -1, _lib.NF_CFG_FP16_CNN)) == 0
    l64, d64, f64 = params.pack("unc", trained_like_variables("unc", 64), 64)
    assert lib.nf_fold_params(C.byref(_lib.nf_config(128, 128, 4, len(l64), -1, 0)), d64, f64.ctypes.data_as(C.POINTER(C.c_float)), f64.size,
                              0, None, 0, C.byref(n_ops), None, 0, None, None) == 0      # ... at every coupling width (GEMM kernels on tiles)
    assert fold(_lib.nf_config(32, 32, 4, len(layers), -1, 0), n=100) == _lib.NF_EINVAL
    assert fold(_lib.nf_config(32, 32, 4, len(layers), -1, 0)) == 0 and n_ops.value == 17
    with pytest.raises(NotImplementedError):
        params.parse_arch("unc|sdn7")          # not a key of noise_flow_arch (noise_flow_model.py:79-234)
    assert [L.kind for L in params.parse_arch("sdn3|gain2")] == ["sdn3", "gain2"]
    with pytest.raises(KeyError):
        params.pack("unc|unc|unc|unc|unc|unc|unc|unc|unc", shipped_variables, 4)   # no 9th template in the ckpt


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: no product module, driver script or native source may import,
    load or mention it (only tests/, __graft_entry__.smoke() and bench.py's checker legs do)."""
    import glob
    import re
    files = (glob.glob(os.path.join(ROOT, "noise_flow_amd", "*.py")) + glob.glob(os.path.join(ROOT, "noise_flow_amd", "csrc", "*"))
             + [os.path.join(ROOT, f) for f in ("train_noise_flow_amd.py", "sample_noise_flow_amd.py")]
             + glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "examples", "*.c")))
    # imports / loads only — a comment may name the oracle file a constant was cross-checked against
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|libnf_oracle|nf_oracle_c|nf_grad_oracle|import_module\([\'\"]oracle", re.M)
    for f in files:
        if f.endswith((".o", ".so")):
            continue
        txt = open(f, errors="replace").read()
        assert not pat.search(txt), f
    # bench.py may use it only inside the checker / cpu_baseline sections of the single-GPU leg, i.e. after every
    # timed region of the hot path (the N > 1 leg and the timing helpers never import it)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    marker = bench.index("parity + CPU baseline (oracle/ is the checker, never the product)")
    uses = [m.start() for m in re.finditer(r"from oracle", bench)]
    assert uses and all(u > marker for u in uses)
    assert "oracle" not in bench[bench.index("def sharded_leg"):bench.index("def single_gpu_leg")]


def test_missing_extension_fails_loudly(tmp_path):
    """No .so -> ImportError naming the build command, in a fresh interpreter (there is no fallback)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from noise_flow_amd import _lib\n"
            "_lib.LIB_PATH = %r\n"
            "try:\n"
            "    _lib.load()\n"
            "except ImportError as e:\n"
            "    assert 'HIP extension not built' in str(e) and 'no CPU fallback' in str(e), e\n"
            "    print('OK')\n") % (ROOT, os.path.join(str(tmp_path), "libnoiseflow_hip.so"))
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0 and out.stdout.decode().strip().endswith("OK"), out.stderr.decode()[-1500:]


@pytest.mark.parametrize("flow_permutation,decomp", [(1, "NONE"), (1, "LU2"), (0, "LU"), (2, "LU")])
def test_other_permutation_settings_fold_like_the_oracle(flow_permutation, decomp):
    """hps.flow_permutation / hps.decomp beyond the shipped (1, 'LU'): noise_flow_model.py:80-92, matrix_param.py:23-29,143-193."""
    from noise_flow_amd import _lib, params
    arch = "unc|unc|gain4|unc"
    v = O.fresh_variables(arch, 4, 4, seed=11, flow_permutation=flow_permutation, decomp=decomp)
    rng = np.random.RandomState(5)
    v["model/sdn_gain/gain_val"] = np.asarray([1.7], np.float32)
    for k in list(v):
        if "Conv2d_1x1" in k and not ("/P_" in k or "sign_S" in k):
            v[k] = (np.asarray(v[k]) + 0.2 * rng.randn(*np.shape(v[k]))).astype(np.float32)   # off the orthogonal start
    assert set(v) == set(params.init_variables(arch, 4, 4, 11, flow_permutation, decomp))
    layers = O.bind_variables(arch, v, flow_permutation=flow_permutation, decomp=decomp)
    mixes = [L for L in layers if L["type"] == "conv1x1"]
    assert len(mixes) == (3 if flow_permutation in (0, 1) else 0)
    for direction in (0, 1):
        ops, blk, ld = _fold(arch, v, 4, direction, flow_permutation=flow_permutation, decomp=decomp)
        got = [blk[off:off + 16].reshape(4, 4).astype(np.float64) for t, off in ops if t == _lib.NF_OP_MIX]
        assert len(got) == len(mixes)
        want_ld = sum(1024 * float(L["log_abs_det"]) for L in mixes) - 4096 * np.log(float(v["model/sdn_gain/gain_val"][0]))
        assert abs(ld - want_ld) < 1e-6 * max(1.0, abs(want_ld))
        g = float(v["model/sdn_gain/gain_val"][0])
        for k, (M, L) in enumerate(zip(got if direction == 0 else got[::-1], mixes)):
            want = L["A"] if direction == 0 else L["A_inv"]
            if k == 2:          # the gain layer in front of the third mix is folded into it
                want = want / g if direction == 0 else want * g
            np.testing.assert_allclose(M, want, rtol=3e-6, atol=3e-7)
    if flow_permutation == 0:
        assert [L.name for L in params.parse_arch(arch, 0)] == ["permute", "unc_0", "permute", "unc_1", "gain_2", "permute", "unc_3"]
    with pytest.raises(ValueError):
        params.parse_arch(arch, 1, "QR")


# ---------------------------------------------------------------------------------------------------------
# GEMM layout (widths 33 .. 512, csrc/nf_gemm.hip): numpy emulation of the kernel's dataflow from the uploaded block
# ---------------------------------------------------------------------------------------------------------
def _fold_layout(arch, variables, width, direction, path, hw=(32, 32), flags=0):
    from noise_flow_amd import _lib, params
    lib = _lib.load()
    layers, descs, flat = params.pack(arch, variables, width, "loss_first", 1, "LU")
    cfg = _lib.nf_config(hw[0], hw[1], 4, len(layers), -1, flags)
    ops = (C.c_int32 * 256)()
    n_ops, lw, nf = C.c_int32(), C.c_int32(), C.c_size_t()
    args = (C.byref(cfg), descs, flat.ctypes.data_as(C.POINTER(C.c_float)), flat.size, direction, path, ops, 128, C.byref(n_ops),
            C.byref(lw))
    _lib.check(lib.nf_fold_layout(*args, None, 0, C.byref(nf)))
    folded = np.zeros(nf.value, np.float32)
    _lib.check(lib.nf_fold_layout(*args, folded.ctypes.data_as(C.POINTER(C.c_float)), folded.size, C.byref(nf)))
    return [(ops[2 * i], ops[2 * i + 1]) for i in range(n_ops.value)], folded, lw.value


def _chan(v, g):
    return 8 * (v >> 2) + 4 * g + (v & 3)


def _mfma_32x32x2(A, B, D):
    """v_mfma_f32_32x32x2_f32 as the kernels use it: A[lane] = A-matrix[i = lane & 31][k = lane >> 5], B[lane] =
    B-matrix[k = lane >> 5][n = lane & 31]; D[v][lane] holds row c(v, lane >> 5), column lane & 31."""
    Am = np.zeros((32, 2)); Bm = np.zeros((2, 32))
    for l in range(64):
        Am[l & 31, l >> 5] = A[l]
        Bm[l >> 5, l & 31] = B[l]
    Cm = Am @ Bm
    out = D.copy()
    for v in range(16):
        for l in range(64):
            out[v, l] += Cm[_chan(v, l >> 5), l & 31]
    return out


@pytest.mark.parametrize("width,variant", [(64, "a"), (96, "a"), (64, "b"), (96, "b")])
def test_gemm_layout_emulated_lane_by_lane_matches_the_oracle_cnn(width, variant, monkeypatch):
    """The NF7 block nf_create uploads at widths > 32, consumed exactly as csrc/nf_gemm.hip consumes it (tile / lane / register
    indices, K-step order, the P rows of the transposed l_last, the 9-tap gather) in a numpy model of v_mfma_f32_32x32x2_f32 —
    against the oracle's coupling CNN on a small patch.  Catches layout / indexing mistakes without a GPU."""
    from noise_flow_amd import _lib
    arch = "unc"
    H, W = 5, 9            # 45 pixels: two pixel tiles, the second one ragged
    v = trained_like_variables(arch, width, seed=width)
    if variant == "a":     # widths <= 128 default to variant B (weights resident in LDS, one slab per channel tile)
        monkeypatch.setenv("NF_GEMM", "a")
    ops, blk, wp = _fold_layout(arch, v, width, 0, _lib.NF_PATH_GEMM, (H, W))
    assert wp == (64 if width <= 64 else 128)
    (t0, off0), (t1, off) = ops
    assert (t0, t1) == (1, 2)
    MT, KC = wp // 32, wp // 8
    img = blk[off + 68:]
    A1 = img[0:MT * 768].reshape(MT, 3, 64, 4)
    B1 = img[MT * 768:MT * 800].reshape(MT, 2, 16)
    if variant == "a":     # NF7_*: one section per operand kind
        B2 = img[MT * 800:MT * 832].reshape(MT, 2, 16)
        A2 = img[MT * 832:MT * 832 + wp * wp].reshape(MT, KC, 64, 4)
        A3 = img[MT * 832 + wp * wp:MT * 832 + wp * wp + MT * 1024].reshape(MT, 4, 64, 4)
        A3C = img[MT * 832 + wp * wp + MT * 1024:MT * 832 + wp * wp + MT * 1152].reshape(MT, 4, 8, 4)
    else:                  # NF10_*: one contiguous slab per channel tile
        slab = 32 * wp + 1024 + 128 + 32
        sl = img[MT * 832:MT * 832 + MT * slab].reshape(MT, slab)
        A2 = sl[:, :32 * wp].reshape(MT, KC, 64, 4)
        A3 = sl[:, 32 * wp:32 * wp + 1024].reshape(MT, 4, 64, 4)
        A3C = sl[:, 32 * wp + 1024:32 * wp + 1152].reshape(MT, 4, 8, 4)
        B2 = sl[:, 32 * wp + 1152:].reshape(MT, 2, 16)
    E = blk[off:off + 64].reshape(16, 4)
    rng = np.random.RandomState(1)
    z0 = rng.randn(H, W, 2)
    z0p = np.zeros((H + 2, W + 2, 2)); z0p[1:-1, 1:-1] = z0
    HW = H * W
    NTILES = (HW + 31) // 32
    P = np.zeros((NTILES * 32, 36))
    for nt in range(NTILES):
        pix = [min(nt * 32 + n, HW - 1) for n in range(32)]
        # l_1
        h1 = np.zeros((MT, 16, 64))
        for m in range(MT):
            d = np.zeros((16, 64))
            for l in range(64):
                d[:, l] = B1[m, l >> 5]
            for tap in range(9):
                Bv = np.array([z0p[pix[l & 31] // W + tap // 3, pix[l & 31] % W + tap % 3, l >> 5] for l in range(64)])
                d = _mfma_32x32x2(A1[m, tap >> 2, :, tap & 3], Bv, d)
            h1[m] = np.maximum(d, 0)
        # l_2: K step kk consumes register kk % 16 of input tile kk // 16
        h2 = np.zeros((MT, 16, 64))
        for m in range(MT):
            d = np.zeros((16, 64))
            for l in range(64):
                d[:, l] = B2[m, l >> 5]
            for kk in range(wp // 2):
                d = _mfma_32x32x2(A2[m, kk >> 2, :, kk & 3], h1[kk // 16, kk % 16], d)
            h2[m] = np.maximum(d, 0)
        # P: taps 0 .. 7 as one 32-row tile; tap 8 on v_mfma_f32_4x4x1 (lane l: D[v] += A[lane 4 (l // 4) + v] * B[l])
        d = np.zeros((16, 64))
        p8 = np.zeros((4, 64))
        for mi in range(MT):
            for vv in range(16):
                d = _mfma_32x32x2(A3[mi, vv >> 2, :, vv & 3], h2[mi, vv], d)
                Av = np.array([A3C[mi, vv >> 2, (l >> 5) * 4 + (l & 3), vv & 3] for l in range(64)])
                for l in range(64):
                    for v4 in range(4):
                        p8[v4, l] += Av[4 * (l // 4) + v4] * h2[mi, vv, l]
        for vv in range(16):
            for l in range(64):
                P[nt * 32 + (l & 31), _chan(vv, l >> 5)] = d[vv, l]
        for l in range(64):
            P[nt * 32 + (l & 31), 32:36] += p8[:, l]          # the two lane halves' partial sums of tap 8
    o = np.zeros((H, W, 4))
    for r in range(H):
        for c in range(W):
            for di in range(3):
                for dj in range(3):
                    rr, cc = r + di - 1, c + dj - 1
                    if 0 <= rr < H and 0 <= cc < W:
                        o[r, c] += P[rr * W + cc, (di * 3 + dj) * 4:(di * 3 + dj) * 4 + 4]
            bm = (r == 0) | (r == H - 1) << 1 | (c == 0) << 2 | (c == W - 1) << 3
            o[r, c] += E[bm]
    o[..., 2:] /= 2.0 * 1.4426950408889634          # raw columns are pre-scaled by 2 log2 e
    cp = [L["p"] for L in O.bind_variables(arch, v) if L["type"] == "coupling"][0]
    shift, raw = O.coupling_cnn(z0[None], cp)
    ref = np.concatenate([shift[0], raw[0]], -1)
    assert np.abs(o - ref).max() <= 2e-5 * np.abs(ref).max(), np.abs(o - ref).max() / np.abs(ref).max()


def _mfma_16x16x32(A, B, D):
    """v_mfma_f32_16x16x32_f16: A[lane][e] = A-matrix[m = lane & 15][k = 8 (lane >> 4) + e], B[lane][e] = B-matrix[k][n = lane & 15];
    D[v][lane] holds row 4 (lane >> 4) + v, column lane & 15."""
    Am = np.zeros((16, 32)); Bm = np.zeros((32, 16))
    for l in range(64):
        Am[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = A[l]
        Bm[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = B[l]
    Cm = Am @ Bm
    out = D.copy()
    for v in range(4):
        for l in range(64):
            out[v, l] += Cm[4 * (l >> 4) + v, l & 15]
    return out


@pytest.mark.parametrize("side", [32, 64])
def test_fp16_big_mfma_layout_emulated_lane_by_lane_matches_the_oracle_cnn(side):
    """The NF11 block nf_create uploads for NF_CFG_FP16_CNN at width 4, consumed exactly as nf_flow_kernel<.., PREC = 2> consumes
    it — pixel ownership of the lanes, the K slots' window rows / column pairs read from the two LDS tiles, the A operands in
    lane order, the 2x2 output block on M — in a numpy model of v_mfma_f32_16x16x32_f16, against the oracle's fp16-CNN
    emulation on a whole patch.  Catches layout / indexing mistakes without a GPU."""
    from noise_flow_amd import _lib
    arch = "unc"
    v = trained_like_variables(arch, 4, seed=11)
    ops, blk, _ = _fold_layout(arch, v, 4, 0, _lib.NF_PATH_FP16, (side, side), flags=_lib.NF_CFG_FP16_CNN)
    (t0, _), (t1, off) = ops
    assert (t0, t1) == (1, 2)
    words = blk.view(np.uint32)
    halves = lambda a, n: words[a:a + n].view(np.float16).astype(np.float64)
    E = blk[off:off + 64].reshape(16, 4).astype(np.float64)
    B1, B2 = blk[off + 64:off + 68].astype(np.float64), blk[off + 68:off + 72].astype(np.float64)
    W2h = halves(off + 76, 8).reshape(4, 4)                 # [j][i]
    A1 = halves(off + 84, 256).reshape(64, 8)
    A3 = halves(off + 340, 512).reshape(2, 64, 8)
    A2 = halves(off + 852, 512).reshape(2, 64, 8) if side == 64 else None   # NF11_CPL_A2: l_2 on the same instruction (64x64 layouts)
    K2 = 2.0 * 1.4426950408889634   # the raw output channels come out pre-scaled by 2 log2(e) (inside the rounded weights / the table)
    Wp = 80 if side == 64 else 48
    l1_row = lambda g: 2 * (g & 1) + (g >> 1)
    l3_row = lambda g, m3: 2 * (g & 1) + m3
    rng = np.random.RandomState(3)
    z0 = rng.randn(side, side, 2)
    h16 = lambda a: np.asarray(a).astype(np.float16).astype(np.float64)
    t0h = np.zeros(((side + 2) * Wp, 2)); thh = np.zeros(((side + 2) * Wp, 4))
    for r in range(side):
        t0h[(r + 1) * Wp + 1:(r + 1) * Wp + 1 + side] = h16(z0[r])
    n_waves = 16 if side == 64 else 4
    o = np.zeros((side, side, 4))
    own = {}
    for phase in (0, 1):
        for wv in range(n_waves):
            q, band = (wv & 1, wv >> 1) if side == 64 else (0, wv)
            r1_of = {}
            for k in range(4):
                lanes = range(64)
                g = [l >> 4 for l in lanes]; n = [l & 15 for l in lanes]
                rr = [band * 8 + 2 * k + (g[l] >> 1) for l in lanes]
                cc = [32 * q + 2 * n[l] + (g[l] & 1) for l in lanes]
                lidx = [(rr[l] + 1) * Wp + cc[l] + 1 for l in lanes]
                if phase == 0:
                    Bop = np.zeros((64, 8))
                    for l in lanes:
                        base = (band * 8 + l1_row(g[l])) * Wp + 32 * q + 2 * n[l] + 2 * k * Wp
                        Bop[l] = t0h[base:base + 4].reshape(-1)
                    d = _mfma_16x16x32(A1, Bop, np.repeat(B1[:, None], 64, 1))
                    r1 = h16(np.maximum(d, 0))
                    r1_of[k] = r1
                    if A2 is None:
                        h2 = B2[:, None] + W2h @ r1                 # 4x4x4: D[j][pixel] = sum_i W2h[j][i] r1[i][pixel]
                    elif k & 1:
                        # units k - 1 and k share ONE B operand: the lane's two pixels, 8 halves; the A operand picks the unit
                        Bop = np.concatenate([r1_of[k - 1].T, r1_of[k].T], axis=1)
                        pair = [_mfma_16x16x32(A2[half], Bop, np.repeat(B2[:, None], 64, 1)) for half in (0, 1)]
                        # unit k - 1's rows were visited one trip ago: store them now
                        rr0 = [band * 8 + 2 * (k - 1) + (g[l] >> 1) for l in lanes]
                        r2 = h16(np.maximum(pair[0], 0))
                        for l in lanes:
                            thh[(rr0[l] + 1) * Wp + cc[l] + 1] = r2[:, l]
                        h2 = pair[1]
                    else:
                        h2 = None
                    if h2 is not None:
                        r2 = h16(np.maximum(h2, 0))
                        for l in lanes:
                            thh[lidx[l]] = r2[:, l]
                    for l in lanes:
                        assert (rr[l], cc[l]) not in own
                        own[(rr[l], cc[l])] = 1
                else:
                    d = np.zeros((4, 64))
                    for l in lanes:
                        bm = (rr[l] == 0) | (rr[l] == side - 1) << 1 | (cc[l] == 0) << 2 | (cc[l] == side - 1) << 3
                        d[:, l] = E[bm]
                    for m3 in (0, 1):
                        Bop = np.zeros((64, 8))
                        for l in lanes:
                            base = (band * 8 + l3_row(g[l], 0)) * Wp + 32 * q + 2 * n[l] + 2 * (g[l] >> 1) + (2 * k + m3) * Wp
                            Bop[l] = thh[base:base + 2].reshape(-1)
                        d = _mfma_16x16x32(A3[m3], Bop, d)
                    for l in lanes:
                        o[rr[l], cc[l]] = d[:, l]
    assert len(own) == side * side
    cp = [L["p"] for L in O.bind_variables(arch, v) if L["type"] == "coupling"][0]
    shift, raw = O.coupling_cnn_fp16(z0[None], cp)
    ref = np.concatenate([shift[0], raw[0] * K2], -1)
    assert np.abs(o - ref).max() <= 1e-5 * np.abs(ref).max(), np.abs(o - ref).max() / np.abs(ref).max()


def _tile_plan(lib, size, tile, halo):
    o, a, b = (C.c_int32 * 512)(), (C.c_int32 * 512)(), (C.c_int32 * 512)()
    n = lib.nf_tile_plan(size, tile, halo, o, a, b, 512)
    return n, list(o[:max(n, 0)]), list(a[:max(n, 0)]), list(b[:max(n, 0)])


def test_tile_plan_partitions_the_axis_and_keeps_the_halo():
    """Patches beyond 64x64 (include/noiseflow_hip.h "Patch sizes"): along each axis the reported windows partition
    [0, size), every tile lies inside the image, and a window stays `halo` pixels away from every tile border that is not
    the image border — the distance beyond which the zero padding at a tile border cannot be seen (2 pixels per coupling:
    3x3 -> 1x1 -> 3x3, layers.py:452-498)."""
    from noise_flow_amd import _lib
    lib = _lib.load()
    assert _tile_plan(lib, 64, 64, 16) == (1, [0], [0], [64])
    assert _tile_plan(lib, 128, 64, 16) == (3, [0, 32, 64], [0, 48, 80], [48, 80, 128])
    assert _tile_plan(lib, 65, 64, 16) == (2, [0, 1], [0, 17], [17, 65])
    for size in list(range(1, 400)) + [1000, 4096]:
        for halo in (0, 2, 8, 16, 28):
            tile = min(size, 64)
            n, o, a, b = _tile_plan(lib, size, tile, halo)
            assert n >= 1 and a[0] == 0 and b[-1] == size
            assert n == 1 or n == -(-(size - tile) // (tile - 2 * halo)) + 1
            for i in range(n):
                assert 0 <= o[i] and o[i] + tile <= size and a[i] < b[i]
                if i > 0:
                    assert a[i] == b[i - 1] and a[i] >= o[i] + halo and o[i] > o[i - 1]
                if i < n - 1:
                    assert b[i] <= o[i] + tile - halo
    assert lib.nf_tile_plan(100, 64, 32, None, None, None, 0) == _lib.NF_EINVAL and b"no core" in lib.nf_last_error()
    assert lib.nf_tile_plan(10, 64, 2, None, None, None, 0) == _lib.NF_EINVAL
    assert lib.nf_tile_plan(200, 64, 16, None, None, None, 0) == 6          # counting only


def test_tiled_evaluation_equals_whole_image_evaluation_in_the_oracle():
    """The claim the tile kernels rest on, checked on the CPU with the fp64 oracle itself: evaluating a 64-pixel tile with
    zero padding at ITS border and keeping the plan's window gives the whole-image latent and per-pixel log-det terms."""
    from noise_flow_amd import _lib
    from oracle.nf_oracle import NoiseFlowOracle
    from conftest import make_inputs, trained_like_variables
    lib = _lib.load()
    arch = "sdn5|unc|unc|gain4|unc"
    v = trained_like_variables(arch, 4, seed=9)
    o = NoiseFlowOracle(arch, v, "loss_first")
    H, W, halo = 80, 70, 6
    x, y = make_inputs(1, H, W, seed=2)
    z_ref, ld_ref = o.inverse(x, y, 100, 2)
    ny, oy, ay, by = _tile_plan(lib, H, 64, halo)
    nx, ox, ax, bx = _tile_plan(lib, W, 64, halo)
    z = np.full_like(z_ref, np.nan)
    for i in range(ny):
        for j in range(nx):
            ys, xs = slice(oy[i], oy[i] + 64), slice(ox[j], ox[j] + 64)
            zt, _ = o.inverse(x[:, ys, xs], y[:, ys, xs], 100, 2)
            # interior tile borders are NOT image borders: the oracle pads them like one, which is what the halo absorbs —
            # except where the border IS the image's (first / last tile), where the padding is the image's own
            z[:, ay[i]:by[i], ax[j]:bx[j]] = zt[:, ay[i] - oy[i]:by[i] - oy[i], ax[j] - ox[j]:bx[j] - ox[j]]
    np.testing.assert_allclose(z, z_ref, rtol=0, atol=1e-12 * np.abs(z_ref).max())


def test_tile_segment_plan(monkeypatch):
    """nf_tile_segments: segments partition the op list, end right after a coupling, carry a halo of 2 x their couplings and
    the tile counts of nf_tile_plan; deeper stacks / larger images get more segments; NF_TILE_SEGMENTS forces the count."""
    from noise_flow_amd import _lib, params
    from conftest import trained_like_variables
    lib = _lib.load()

    def plan(arch, H, W, direction=0):
        v = trained_like_variables(arch, 4)
        layers, descs, flat = params.pack(arch, v, 4)
        cfg = _lib.nf_config(H, W, 4, len(layers), -1, 0)
        fp = flat.ctypes.data_as(C.POINTER(C.c_float))
        out = (C.c_int32 * 80)()
        n = lib.nf_tile_segments(C.byref(cfg), descs, fp, flat.size, direction, out, 16)
        ops = (C.c_int32 * 128)()
        n_ops = C.c_int32()
        assert lib.nf_fold_params(C.byref(cfg), descs, fp, flat.size, direction, ops, 64, C.byref(n_ops), None, 0, None, None) == 0
        types = [ops[2 * i] for i in range(n_ops.value)]
        return n, [tuple(out[5 * i:5 * i + 5]) for i in range(max(n, 0))], types

    monkeypatch.delenv("NF_TILE_SEGMENTS", raising=False)
    assert plan(FULL_ARCH, 64, 64)[0] == 0 and plan(FULL_ARCH, 32, 32)[0] == 0
    counts = {}
    for H, W in ((65, 64), (128, 128), (256, 256), (1024, 1024), (32, 200)):
        for direction in (0, 1):
            n, segs, types = plan(FULL_ARCH, H, W, direction)
            counts[(H, W)] = n
            assert n >= 1 and segs[0][0] == 0 and segs[-1][1] == len(types)
            for i, (op0, op1, halo, ny, nx) in enumerate(segs):
                ncpl = sum(1 for t in types[op0:op1] if t in (2, 3))          # NF_OP_COUPLING_FWD / _REV
                assert halo == 2 * ncpl and 64 - 2 * halo >= 8
                assert i == 0 or op0 == segs[i - 1][1]
                assert i == n - 1 or types[op1 - 1] in (2, 3)
                assert ny == lib.nf_tile_plan(H, min(H, 64), halo, None, None, None, 0)
                assert nx == lib.nf_tile_plan(W, min(W, 64), halo, None, None, None, 0)
            assert sum(s[2] for s in segs) == 16                               # 8 couplings in all
    assert counts[(65, 64)] == 1 and counts[(1024, 1024)] > counts[(128, 128)]
    assert plan("|".join(["unc"] * 16), 100, 100)[0] >= 2                      # halo 32 would leave no core
    monkeypatch.setenv("NF_TILE_SEGMENTS", "3")
    n, segs, _ = plan(FULL_ARCH, 128, 128)
    assert n == 3 and [s[2] for s in segs] == [6, 6, 4]
