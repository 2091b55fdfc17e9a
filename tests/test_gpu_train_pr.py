"""GPU parity of the patch-resident width-32 training stages (csrc/nf_train_pr.h: 32x32 patches, the paper's coupling width,
job_noise_flow.sh:18) against the fp64 autograd oracle and against the stage kernels they replace (csrc/nf_train_wide.h,
NF_TRAIN_PR=0).  Tolerances as tests/test_gpu_train.py: loss 1e-5 relative, gradients 2e-4 of each tensor's scale or the
oracle's round-off allowance, the other ReLU branch only at activations the oracle reports on their kink."""
import numpy as np
import pytest

from conftest import make_inputs, trained_like_variables
from test_gpu_train import _oracle_check_next_to_kinks, _trainer

pytestmark = pytest.mark.gpu

ARCH = "sdn5|unc|unc|gain4|unc"


def _run(monkeypatch, pr, arch, v, x, y, grid=None, max_batch=64):
    monkeypatch.setenv("NF_TRAIN_PR", pr)
    if grid is None:
        monkeypatch.delenv("NF_TRAIN_PR_GRID", raising=False)
    else:
        monkeypatch.setenv("NF_TRAIN_PR_GRID", str(grid))
    tr = _trainer(arch, v, (32, 32, 4), 32, max_batch=max_batch)
    g, loss = tr.forward_backward(x, y, [0.0], [0.0], [400], [1])
    out = (g.cpu().numpy().copy(), loss.cpu().numpy().copy(), tr.raw_params())
    return tr, out


@pytest.mark.parametrize("pr,grid,B,seed", [("1", None, 5, 4),      # 8 wavefronts per patch, one patch per workgroup
                                            ("2", None, 5, 6),      # 4 wavefronts per patch
                                            ("1", 3, 8, 2),         # 3 workgroups walk 8 patches: the persistent loops
                                            ("2", 2, 5, 5)])
def test_patch_resident_stages_against_the_oracle(pr, grid, B, seed, monkeypatch):
    v = trained_like_variables(ARCH, 32, seed=seed)
    x, y = make_inputs(B, 32, 32, seed=seed + 20)
    tr, _ = _run(monkeypatch, pr, ARCH, v, x, y, grid)
    _oracle_check_next_to_kinks(tr, ARCH, v, x, y, 400, 1, 32, rtol=1e-3)
    tr.close()


def test_patch_resident_stages_against_the_stage_kernels(monkeypatch):
    """Same sums in another order: loss to 1e-6, gradients to a few pixels' worth (an activation within round-off of its ReLU kink
    may take the other branch in one of the two paths), BN running moments to 1e-5; and the result does not depend on how the
    patches are dealt to the workgroups beyond the summation order."""
    arch, B = "unc|unc", 40
    v = trained_like_variables(arch, 32, seed=2)
    x, y = make_inputs(B, 32, 32, seed=22)
    res = {}
    # NF_TRAIN_PR bit 2 / bit 3: stage 0 of the coupling above / stage A of the coupling below in launches of their own; bit 4: d l_last/W
    # inside stage A instead of on the side stream (40 patches would not take the side stream: the second test below does)
    for key, pr, grid in (("wide", "0", None), ("pr8", "1", None), ("pr4", "2", None), ("pr8_grid7", "1", 7), ("pr8_unfused", "29", None)):
        tr, res[key] = _run(monkeypatch, pr, arch, v, x, y, grid)
        tr.close()
    g0, l0, p0 = res["wide"]
    tol = 8.0 / (B * 1024)
    for key in ("pr8", "pr4", "pr8_grid7", "pr8_unfused"):
        g, l, p = res[key]
        assert np.allclose(l, l0, rtol=1e-6, atol=0), (key, l, l0)
        assert np.abs(g - g0).max() <= tol * np.abs(g0).max(), (key, np.abs(g - g0).max(), np.abs(g0).max())
        assert np.allclose(p, p0, rtol=1e-5, atol=1e-5), key
    assert not np.array_equal(res["pr8"][0], res["wide"][0])      # it really is another code path
    # fused or not, a stage does the same arithmetic in the same order
    assert np.array_equal(res["pr8"][0], res["pr8_unfused"][0]) and np.array_equal(res["pr8"][2], res["pr8_unfused"][2])


def test_patch_resident_filter_gradient_on_the_side_stream(monkeypatch):
    """Between a quarter and three quarters of the CUs busy (here 70 patches), d l_last/W runs as a launch of its own on the side stream,
    in the idle CUs, from the gu its stage left in HBM: the same products, each patch's in the same order — the other gradients are
    bit-identical to the in-stage variant's, d l_last/W differs by the order its per-workgroup partials are added in at most."""
    arch, B = "unc|unc|unc|unc", 70
    v = trained_like_variables(arch, 32, seed=7)
    x, y = make_inputs(B, 32, 32, seed=27)
    res, grads = {}, {}
    for key, pr in (("side", "1"), ("inside", "17")):
        tr, res[key] = _run(monkeypatch, pr, arch, v, x, y, None, max_batch=B)
        grads[key] = tr.raw_to_variables(res[key][0])
        tr.close()
    gs, gi = grads["side"], grads["inside"]
    assert np.allclose(res["side"][1], res["inside"][1], rtol=0, atol=0)
    for nm in gs:
        a, b = np.asarray(gs[nm], np.float64), np.asarray(gi[nm], np.float64)
        if nm.endswith("l_last/W"):
            assert np.abs(a - b).max() <= 1e-6 * max(np.abs(b).max(), 1e-30), nm
        else:
            assert np.array_equal(a, b), nm
    assert np.array_equal(res["side"][2], res["inside"][2])


def test_patch_resident_training_is_bit_reproducible(monkeypatch):
    """No atomics: slotted partial sums added up in a fixed order — two runs of three optimizer steps end in the same bits."""
    arch = "unc|gain4|unc"
    v = trained_like_variables(arch, 32, seed=3)
    x, y = make_inputs(9, 32, 32, seed=33)
    monkeypatch.setenv("NF_TRAIN_PR", "1")
    outs = []
    for _ in range(2):
        tr = _trainer(arch, v, (32, 32, 4), 32, max_batch=16)
        for _ in range(3):
            tr.step(x, y, [0.0], [0.0], [400], [1], lr=1e-3)
        outs.append(tr.raw_params().copy())
        tr.close()
    assert np.array_equal(outs[0], outs[1])
