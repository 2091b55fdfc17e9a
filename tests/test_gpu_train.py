"""GPU parity of the training step (C ABI ``nf_trainer_*``) against the fp64 autograd oracle
(``oracle/nf_grad_oracle.py``; itself pinned to the forward oracle and to finite differences in
``tests/test_oracle.py``).

Tolerances: loss / sd_z 1e-5 relative; every gradient tensor within 2e-4 of its own max |entry|
(fp32 activations, fp64 accumulation of the parameter sums; the gradients of l_1/b and l_2/b are
analytically ZERO — BN subtracts the batch mean — so those are compared on an absolute scale);
BN running statistics 1e-5 of their scale; optimizer updates bit-level vs a float32 numpy
restatement of the same update rule.
"""
import os

import numpy as np
import pytest

from conftest import FULL_ARCH, SHIPPED_DIR, make_inputs, trained_like_variables

pytestmark = pytest.mark.gpu

GRAD_RTOL = 2e-4


def _trainer(arch, variables, x_shape=(32, 32, 4), width=4, optim="adam", max_batch=64):
    from noise_flow_amd import default_hps
    from noise_flow_amd.train import Trainer
    return Trainer(list(x_shape), default_hps(arch=arch, width=width), variables=variables, optim=optim, max_batch=max_batch)


def _grad_oracle(arch, variables):
    from oracle.nf_grad_oracle import GradOracle
    return GradOracle(arch, variables)


def _check_grads(tr, grads_dev, ref_grads, loss_scale=1.0, rtol=GRAD_RTOL, oracle=None):
    """`oracle` (optional): the GradOracle whose evaluation `ref_grads` is; an entry may then also be within the round-off
    allowance of the sum it is (conftest.py::grad_noise_allowance — a bound the fp64 oracle computes, not a second GPU run)."""
    from conftest import grad_noise_allowance
    got = tr.raw_to_variables(grads_dev.cpu().numpy())
    from oracle.nf_grad_oracle import is_trainable
    from noise_flow_amd import params as P
    names = [nm for L in tr.layers for nm in P.layer_variable_names(L, tr._tmpl) if nm is not None]
    gmax = max(np.abs(ref_grads[nm]).max() for nm in names if is_trainable(nm))
    checked = 0
    for nm in names:
        if not is_trainable(nm):
            assert np.all(np.asarray(got[nm]) == 0), nm      # masked out
            continue
        ref = np.asarray(ref_grads[nm], np.float64)
        g = np.asarray(got[nm], np.float64).reshape(ref.shape)
        scale = np.abs(ref).max()
        if nm.endswith("l_1/b") or nm.endswith("l_2/b"):
            assert scale < 1e-9 * gmax                       # analytically zero
            tol0 = 1e-5 * gmax                               # ... and round-off of a sum whose terms cancel exactly
            if oracle is not None:
                tol0 = np.maximum(tol0, grad_noise_allowance(oracle, nm).reshape(ref.shape))
            assert (np.abs(g) <= tol0).all(), (nm, np.abs(g).max(), gmax)
        else:
            floor = 1e-6 * gmax
            tol = rtol * max(scale, floor)
            if oracle is not None:
                tol = np.maximum(tol, grad_noise_allowance(oracle, nm).reshape(ref.shape))
            assert (np.abs(g - ref) <= tol).all(), (nm, np.abs(g - ref).max(), scale, np.max(tol))
        checked += ref.size
    return checked


def test_gradients_full_arch_shipped(shipped_variables):
    x, y = make_inputs(6, seed=41, b1=0.003696)
    tr = _trainer(FULL_ARCH, shipped_variables)
    grads, loss = tr.forward_backward(x, y, [0.0], [0.0], [800], [2])
    o = _grad_oracle(FULL_ARCH, shipped_variables)
    ref_loss, ref_sd, ref_grads, new_running = o.loss_and_grads(x, y, 800, 2)
    lv = loss.cpu().numpy()
    assert abs(lv[0] - ref_loss) <= 1e-5 * abs(ref_loss)
    assert abs(lv[1] - ref_sd) <= 1e-5 * ref_sd
    n = _check_grads(tr, grads, ref_grads)
    assert n == 2431          # 2433 trainables minus the two rescaling_scale variables of sdn_0 / gain_5 (unused)
    # BN running statistics moved by the EMA of the batch moments (layers.py:392-393)
    v = tr.variables
    for k, want in new_running.items():
        scale = max(np.abs(want).max(), 1e-3)
        assert np.abs(v[k] - want).max() <= 1e-5 * scale, k
    # trainables untouched by forward_backward
    for k in shipped_variables:
        if "bn_nvp_conv" not in k:
            assert np.array_equal(np.asarray(v[k], np.float32).reshape(-1), np.asarray(shipped_variables[k], np.float32).reshape(-1)), k


@pytest.mark.parametrize("arch,width,hw,B,iso,cam", [("unc|unc", 8, (24, 40), 5, 400, 1),
                                                     ("sdn5|unc|gain4|unc", 16, (16, 16), 7, 1600, 3),
                                                     ("sdn5|unc|gain4|unc", 4, (20, 12), 3, 250, 0),     # unknown ISO
                                                     ("unc", 32, (8, 8), 9, 100, 2),
                                                     ("sdn4|gain4", 4, (32, 32), 4, 800, 9),            # job_noise_flow.sh "S-G"
                                                     ("sdn4|unc|gain4", 4, (16, 16), 4, 123, 2),
                                                     ("sdn5|gain4", 4, (32, 32), 4, 3200, 4),           # "S-G-CAM"
                                                     ("unc|unc|unc|unc", 4, (32, 32), 3, 100, 2),       # "Ax4"
                                                     ("sdn5|unc|gain4|unc", 4, (72, 80), 2, 400, 2)])   # beyond 64x64: layer kernels
def test_gradients_other_widths_and_shapes(arch, width, hw, B, iso, cam):
    v = trained_like_variables(arch, width, seed=6)
    x, y = make_inputs(B, hw[0], hw[1], seed=17)
    tr = _trainer(arch, v, (hw[0], hw[1], 4), width)
    _oracle_check_next_to_kinks(tr, arch, v, x, y, iso, cam, width)


def _oracle_check_next_to_kinks(tr, arch, v, x, y, iso, cam, width, rtol=GRAD_RTOL):
    """Loss, sd_z and every gradient tensor against the fp64 oracle — ONE evaluation, no re-draws.  The loss is piecewise
    smooth: an activation within float32 round-off of its ReLU kink may take one branch in the fp64 oracle and the other on
    the GPU.  The oracle names exactly those activations (margin < 32 units of the round-off of the sum that produced them)
    and `grads_match_up_to_kinks` accepts the other branch at those and nowhere else; the number it had to excuse is
    returned (0 on almost every input)."""
    from conftest import grads_match_up_to_kinks
    grads, loss = tr.forward_backward(x, y, [0.0], [0.0], [iso], [cam])
    lv = loss.cpu().numpy()
    o = _grad_oracle(arch, v)

    def compare(ref_loss, ref_sd, ref_grads):
        assert abs(lv[0] - ref_loss) <= 1e-5 * abs(ref_loss)
        assert abs(lv[1] - ref_sd) <= 1e-5 * ref_sd
        _check_grads(tr, grads, ref_grads, rtol=rtol, oracle=o)     # o.grad_abs_terms belongs to the evaluation being compared
    # (an input with MANY on-kink candidates: the branch subset is solved for from the gradients and re-evaluated exactly)
    excused = grads_match_up_to_kinks(o, x, y, iso, cam, compare, max_kinks=64, got=tr.raw_to_variables(grads.cpu().numpy().copy()))
    assert excused <= 2, excused
    return excused


def test_optimizer_kernels_match_float32_restatement(shipped_variables):
    """Adam / momentum on supplied gradients: the update rule alone, elementwise."""
    import torch
    rng = np.random.RandomState(3)
    for optim in ("adam", "sgd"):
        tr = _trainer(FULL_ARCH, shipped_variables, optim=optim)
        p = tr.raw_params().astype(np.float32)
        m = np.zeros_like(p)
        v = np.zeros_like(p)
        lr = np.float32(1e-3)
        # which raw entries are trainable: forward_backward writes exact zeros elsewhere; use a probe gradient
        x, y = make_inputs(2, seed=1)
        gprobe, _ = tr.forward_backward(x, y, [0.0], [0.0], [100], [2])
        bn_before = tr.raw_params()
        trainable = np.ones(tr.n_params, bool)
        from noise_flow_amd import params as P
        from oracle.nf_grad_oracle import is_trainable
        pos = 0
        for L in tr.layers:
            for nm in P.layer_variable_names(L, tr._tmpl):
                n = 1 if nm is None else int(np.asarray(shipped_variables[nm]).size)
                if nm is None or not is_trainable(nm):
                    trainable[pos:pos + n] = False
                pos += n
        p = bn_before.copy()
        for t in range(1, 4):
            g = (rng.randn(tr.n_params) * 10.0 ** rng.uniform(-6, 2, tr.n_params)).astype(np.float32)
            tr.apply(float(lr), torch.from_numpy(g).cuda())
            if optim == "adam":
                lr_t = np.float32(np.float64(lr) * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t))
                m = m + (g - m) * (np.float32(1) - np.float32(0.9))        # float32 arithmetic, as TF's kernel
                v = v + (g * g - v) * (np.float32(1) - np.float32(0.999))
                upd = lr_t * m / (np.sqrt(v) + np.float32(1e-8))
            else:
                m = np.float32(0.9) * m + g
                upd = lr * m
            p = np.where(trainable, p - upd, p).astype(np.float32)
            got = tr.raw_params()
            assert tr.steps == t
            np.testing.assert_allclose(got, p, rtol=2e-6, atol=1e-5 * float(lr))   # 1e-5 of one step
            p = got.copy()   # do not let 1-ulp differences accumulate


def test_training_steps_follow_the_oracle_and_reduce_the_loss():
    """Three Adam steps from a fresh initialisation on changing minibatches: the loss sequence
    follows the fp64 oracle's, and the variables with a well-conditioned gradient move with it."""
    from oracle.nf_grad_oracle import train_step, is_trainable
    from noise_flow_amd import params as P
    arch = "sdn5|unc|unc|gain4|unc"
    v0 = trained_like_variables(arch, 4, seed=9)
    tr = _trainer(arch, v0)
    ref_vars, state = dict(v0), {}
    lr = 1e-3
    for k in range(3):
        x, y = make_inputs(8, seed=100 + k, b1=0.003696)
        loss, sd = tr.step(x, y, [0.0], [0.0], [800], [2], lr=lr)
        ref_vars, ref_loss, ref_sd = train_step(arch, ref_vars, x, y, 800, 2, state, lr)
        assert abs(loss - ref_loss) <= 2e-4 * abs(ref_loss), (k, loss, ref_loss)
        assert abs(sd - ref_sd) <= 2e-4 * ref_sd
    got = tr.variables
    for nm in got:
        if not is_trainable(nm) or nm.endswith("/b"):
            continue
        a, b, s = np.asarray(got[nm], np.float64), np.asarray(ref_vars[nm], np.float64), np.asarray(v0[nm], np.float64)
        if a.shape != b.shape:
            a = a.reshape(b.shape)
        moved = np.abs(b - s.reshape(b.shape)).max()
        if moved > 0:
            # Adam normalises the step: entries whose gradient sits at the fp32 noise floor may step
            # the other way; the bulk must agree
            frac_bad = np.mean(np.abs(a - b) > 0.05 * 3 * lr)
            assert frac_bad <= 0.05, (nm, frac_bad)

    # a longer run on a fixed minibatch must reduce its loss
    tr2 = _trainer(arch, v0)
    x, y = make_inputs(16, seed=7, b1=0.003696)
    first = tr2.step(x, y, [0.0], [0.0], [800], [2], lr=2e-3)[0]
    for _ in range(40):
        last = tr2.step(x, y, [0.0], [0.0], [800], [2], lr=2e-3)[0]
    assert last < first - 1.0, (first, last)


def test_trained_parameters_round_trip_into_the_eval_path(tmp_path):
    """Checkpoint written by the trainer → restored by NoiseFlow → NLL equals the oracle's on the
    trained variables (the epoch loop of train_noise_flow.py alternates exactly these)."""
    from noise_flow_amd import NoiseFlow, default_hps
    from oracle.nf_oracle import NoiseFlowOracle
    arch = "sdn5|unc|gain4|unc"
    v0 = trained_like_variables(arch, 4, seed=2)
    tr = _trainer(arch, v0)
    x, y = make_inputs(8, seed=5, b1=0.003696)
    for _ in range(5):
        tr.step(x, y, [0.0], [0.0], [800], [2], lr=1e-3, sync=False)
    prefix = os.path.join(str(tmp_path), "model.ckpt")
    tr.save(prefix)
    m = NoiseFlow([32, 32, 4], False, default_hps(arch=arch))
    m.restore(prefix)
    nll, _ = m._loss(x, y, [0.0], [0.0], [800], [2])
    ref = NoiseFlowOracle(arch, tr.variables).nll(x, y, 800, 2)[0]
    np.testing.assert_allclose(nll, ref, rtol=1e-5)


def test_forward_only_equals_the_batch_statistics_eval_path(shipped_variables):
    """Trainer.forward (condSDN branch) = NoiseFlow(is_training=True).loss: same loss / sd_z, same EMA,
    parameters untouched."""
    from noise_flow_amd import NoiseFlow, default_hps
    x, y = make_inputs(12, seed=77, b1=0.003696)
    tr = _trainer(FULL_ARCH, shipped_variables)
    before = tr.raw_params()
    loss, sd = tr.forward(x, y, [0.0], [0.0], [800], [2])
    m = NoiseFlow([32, 32, 4], True, default_hps(), variables=shipped_variables)
    ref_loss, ref_sd = m.loss(x, y, [0.0], [0.0], [800], [2])
    assert abs(loss - ref_loss) <= 1e-5 * abs(ref_loss) and abs(sd - ref_sd) <= 1e-5 * ref_sd
    after = tr.variables
    for k, v in m.variables.items():
        a = np.asarray(after[k], np.float32).reshape(-1)
        if "bn_nvp_conv" in k:
            assert np.abs(a - np.asarray(v, np.float32).reshape(-1)).max() <= 1e-5 * max(np.abs(v).max(), 1e-3), k
        else:
            assert np.array_equal(a, np.asarray(shipped_variables[k], np.float32).reshape(-1)), k
    assert tr.steps == 0 and not np.array_equal(before, tr.raw_params())     # only the BN statistics moved


def test_trainer_c_abi_errors(shipped_variables):
    import ctypes as C
    import torch
    from noise_flow_amd import _lib, params as P
    lib = _lib.load()
    tr = _trainer(FULL_ARCH, shipped_variables, max_batch=4)
    x = torch.zeros(8, 32, 32, 4, device="cuda")
    cond = _lib.nf_cond(100.0, 2.0, 0.0, 0.0)
    rc = lib.nf_trainer_forward_backward(tr._h, x.data_ptr(), x.data_ptr(), 8, C.byref(cond), None, None, None)
    assert rc == _lib.NF_EINVAL and b"max_batch" in lib.nf_last_error()
    rc = lib.nf_trainer_forward_backward(tr._h, x.data_ptr(), None, 2, C.byref(cond), None, None, None)
    assert rc == _lib.NF_EINVAL
    bad = _lib.nf_cond(100.0, 7.0, 0.0, 0.0)
    rc = lib.nf_trainer_forward_backward(tr._h, x.data_ptr(), x.data_ptr(), 2, C.byref(bad), None, None, None)
    assert rc == _lib.NF_ECOND
    # an unknown layer type
    layers, descs, flat = P.pack("sdn|unc", trained_like_variables("sdn|unc", 4, seed=1), 4)
    descs[0].type = 99
    cfg = _lib.nf_config(32, 32, 4, len(layers), -1, 0)
    h = C.c_void_p()
    rc = lib.nf_trainer_create(C.byref(cfg), descs, flat.ctypes.data_as(C.POINTER(C.c_float)), flat.size, 4, 0, C.byref(h))
    assert rc == _lib.NF_EINVAL
    # coupling widths beyond the trainer's (it takes 1 .. 512)
    for w in (520,):
        layers, descs, flat = P.pack("sdn|unc", trained_like_variables("sdn|unc", w, seed=1), w)
        cfg = _lib.nf_config(32, 32, 4, len(layers), -1, 0)
        rc = lib.nf_trainer_create(C.byref(cfg), descs, flat.ctypes.data_as(C.POINTER(C.c_float)), flat.size, 4, 0, C.byref(h))
        assert rc == _lib.NF_EINVAL and b"coupling widths" in lib.nf_last_error(), (w, lib.nf_last_error())


def test_fit_epoch_loop_logs_checkpoints_and_learns(tmp_path):
    """harness.fit = the epoch loop of train_noise_flow.py:379-511 on synthetic NLF noise."""
    from noise_flow_amd import NoiseFlow, default_hps, patches
    from noise_flow_amd.harness import fit, TRAIN_COLUMNS, TEST_COLUMNS, SAMPLE_COLUMNS
    from noise_flow_amd.train import Trainer
    arch = "sdn5|unc|gain4|unc"
    hps = default_hps(arch=arch, seed=1)
    tr = Trainer([32, 32, 4], hps, max_batch=32)
    nf = NoiseFlow([32, 32, 4], False, hps, variables=tr.variables)
    nlf = (0.003696, 2e-6)

    def mbs(first, n):
        out = []
        for k in range(n):
            x, y = patches.synth_patches(3, first + 32 * k, 32, nlf=nlf)
            out.append({"_x": x, "_y": y, "nlf0": [nlf[0]], "nlf1": [nlf[1]], "iso": [800.0], "cam": [2.0]})
        return out

    logdir = str(tmp_path)
    res = fit(tr, nf, mbs(0, 6), mbs(1000, 2), logdir, epochs=12, lr=2e-3, epochs_full_valid=10)
    assert len(res["train"]) == 12 and len(res["test"]) == 10      # epochs 1..9 and 10
    assert res["test"][-1] < res["test"][0] - 10.0                  # it learns
    for name, cols, rows in (("train.txt", TRAIN_COLUMNS, 12), ("test.txt", TEST_COLUMNS, 10), ("sample.txt", SAMPLE_COLUMNS, 10)):
        lines = open(os.path.join(logdir, name)).read().split("\n")
        assert lines[0].split("\t") == cols and len(lines) == rows + 1, name
    # the best checkpoint restores into the eval path and reproduces the best test loss
    from noise_flow_amd.harness import test_epoch
    m = NoiseFlow([32, 32, 4], False, hps)
    m.restore(os.path.join(logdir, "ckpt", "model.ckpt.best"))
    nll, _, _ = test_epoch(m, mbs(1000, 2))
    assert abs(nll - min(res["test"])) <= 1e-4 * abs(nll)
    assert os.path.exists(os.path.join(logdir, "ckpt", "model.ckpt-10.index"))


def test_data_parallel_step_on_a_one_rank_group(shipped_variables):
    """step(group=...) = forward_backward -> all_reduce(SUM)/world -> apply; on a 1-rank RCCL group it
    must leave bit-identical parameters (the collective itself is exercised)."""
    import torch
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        x, y = make_inputs(8, seed=12, b1=0.003696)
        a = _trainer(FULL_ARCH, shipped_variables)
        b = _trainer(FULL_ARCH, shipped_variables)
        for _ in range(3):
            la = a.step(x, y, [0.0], [0.0], [800], [2], lr=1e-3)
            lb = b.step(x, y, [0.0], [0.0], [800], [2], lr=1e-3, group=True)
            # the reported loss sums per-patch float atomics (order-dependent in the last bits) ...
            assert abs(la[0] - lb[0]) <= 1e-6 * abs(la[0]) and abs(la[1] - lb[1]) <= 1e-6 * la[1]
        # ... the gradients and the update are slot-reduced and deterministic
        assert np.array_equal(a.raw_params(), b.raw_params())
    finally:
        if created:
            dist.destroy_process_group()


DP_ARCH = "sdn5|unc|gain4|unc"


def _dp_worker(rank, world, port, outdir):
    """One data-parallel rank (both ranks share the single GPU of the test box; gloo carries the
    CUDA gradient).  Each rank trains on its own shard with group=True."""
    import sys
    import torch.distributed as dist
    from conftest import ROOT, make_inputs, trained_like_variables
    sys.path.insert(0, ROOT)
    from noise_flow_amd import default_hps
    from noise_flow_amd.train import Trainer
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    v = trained_like_variables(DP_ARCH, 4, seed=9)
    tr = Trainer([32, 32, 4], default_hps(arch=DP_ARCH), variables=v, max_batch=8)
    losses = []
    for k in range(3):
        x, y = make_inputs(8, seed=300 + 10 * k + rank, b1=0.003696)
        losses.append(tr.step(x, y, [0.0], [0.0], [800], [2], lr=1e-3, group=True))
    np.save(os.path.join(outdir, "params_%d.npy" % rank), tr.raw_params())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_training_matches_manual_gradient_averaging(tmp_path):
    """world_size 2: forward_backward on each rank's shard → all-reduce(SUM)/2 → apply.  Both ranks
    must end with identical trainable parameters, equal (bit for bit) to two single-process
    trainers whose gradients are averaged by hand; BN running statistics stay per rank."""
    import socket
    import torch
    import torch.multiprocessing as mp
    from noise_flow_amd import params as P
    from oracle.nf_grad_oracle import is_trainable
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    [p.start() for p in procs]
    [p.join(timeout=300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    p0, p1 = (np.load(os.path.join(str(tmp_path), "params_%d.npy" % r)) for r in range(2))

    v = trained_like_variables(DP_ARCH, 4, seed=9)
    a, b = _trainer(DP_ARCH, v, max_batch=8), _trainer(DP_ARCH, v, max_batch=8)
    for k in range(3):
        gs = []
        for rank, tr in enumerate((a, b)):
            x, y = make_inputs(8, seed=300 + 10 * k + rank, b1=0.003696)
            g, _ = tr.forward_backward(x, y, [0.0], [0.0], [800], [2])
            gs.append(g.clone())
        mean = (gs[0] + gs[1]) / 2
        a.apply(1e-3, mean)
        b.apply(1e-3, mean)
    # which raw entries are trainable
    mask = np.zeros(a.n_params, bool)
    pos = 0
    for L in a.layers:
        for nm in P.layer_variable_names(L, a._tmpl):
            n = 1 if nm is None else int(np.asarray(v[nm]).size)
            mask[pos:pos + n] = nm is not None and is_trainable(nm)
            pos += n
    assert np.array_equal(p0[mask], p1[mask])
    assert np.array_equal(p0[mask], a.raw_params()[mask])
    assert np.array_equal(p0, a.raw_params()) and np.array_equal(p1, b.raw_params())   # incl. per-rank BN statistics
    assert not np.array_equal(p0[~mask], p1[~mask])


def _syncbn_worker(rank, world, port, outdir, width=4, hw=32, per=8):
    """One rank of a synchronised-BN data-parallel run: its shard of the global minibatch, group + sync_bn."""
    import sys
    import torch.distributed as dist
    from conftest import ROOT, make_inputs, trained_like_variables
    sys.path.insert(0, ROOT)
    from noise_flow_amd import default_hps
    from noise_flow_amd.train import Trainer
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    v = trained_like_variables(DP_ARCH, width, seed=9)
    tr = Trainer([hw, hw, 4], default_hps(arch=DP_ARCH, width=width), variables=v, max_batch=per)
    x, y = make_inputs(per * world, hw, hw, seed=500, b1=0.003696)
    sl = slice(per * rank, per * rank + per)
    # the two halves of step(group=True, sync_bn=True), with the averaged gradient kept for the comparison
    tr.set_sync_bn(True)
    g, loss = tr.forward_backward(x[sl], y[sl], [0.0], [0.0], [800], [2])
    dist.all_reduce(g)
    g.div_(world)
    np.save(os.path.join(outdir, "sync_grad_%d.npy" % rank), g.cpu().numpy())
    np.save(os.path.join(outdir, "sync_loss_%d.npy" % rank), loss.cpu().numpy())
    tr.apply(1e-3, g)
    np.save(os.path.join(outdir, "sync_params1_%d.npy" % rank), tr.raw_params())
    x2, y2 = make_inputs(per * world, hw, hw, seed=501, b1=0.003696)
    tr.step(x2[sl], y2[sl], [0.0], [0.0], [800], [2], lr=1e-3, group=True, sync_bn=True)
    np.save(os.path.join(outdir, "sync_params_%d.npy" % rank), tr.raw_params())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sync_bn_step_equals_one_rank_on_the_concatenated_batch(tmp_path):
    """SURVEY §8 f-3 'training-mode BN with cross-GPU moment all-reduce': with sync_bn the batch moments (and the two
    batch means of the BN gradient) are taken over the union of the ranks' shards, so the rank-averaged gradient of
    2 ranks x 8 patches IS the gradient of one rank on the 16 concatenated patches (layers.py:386-398; up to the order
    of the fp32 partial sums), the running statistics move identically, and both ranks stay bit-identical over further
    steps.  Without the hook the same shards give a different gradient."""
    import socket
    import torch.multiprocessing as mp
    from noise_flow_amd import params as P
    from oracle.nf_grad_oracle import is_trainable
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    [p.start() for p in procs]
    [p.join(timeout=300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    load = lambda n, r: np.load(os.path.join(str(tmp_path), "sync_%s_%d.npy" % (n, r)))   # noqa: E731
    g0, g1, p0, p1 = load("grad", 0), load("grad", 1), load("params", 0), load("params", 1)
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)      # same model on both ranks, running statistics included

    v = trained_like_variables(DP_ARCH, 4, seed=9)
    x, y = make_inputs(16, seed=500, b1=0.003696)
    one = _trainer(DP_ARCH, v, max_batch=16)
    g_one, loss_one = one.forward_backward(x, y, [0.0], [0.0], [800], [2])
    g_one = g_one.cpu().numpy().copy()
    ref = one.raw_to_variables(g_one)
    got = one.raw_to_variables(g0)
    gmax = max(np.abs(ref[n]).max() for n in ref if is_trainable(n))
    for name in ref:
        if not is_trainable(name):
            continue
        if name.endswith("l_1/b") or name.endswith("l_2/b"):     # analytically zero (a bias in front of a batch norm)
            assert np.abs(got[name]).max() <= 1e-5 * gmax and np.abs(ref[name]).max() <= 1e-5 * gmax
        else:
            assert np.abs(got[name] - ref[name]).max() <= 2e-4 * max(np.abs(ref[name]).max(), 1e-6 * gmax), name
    # the global-batch loss is the mean of the two ranks' losses
    l2 = 0.5 * (load("loss", 0)[0] + load("loss", 1)[0])
    assert abs(l2 - float(loss_one.cpu().numpy()[0])) <= 1e-5 * abs(l2)
    # per-rank statistics (no hook) on the same shards: a measurably different gradient
    a, b = _trainer(DP_ARCH, v, max_batch=8), _trainer(DP_ARCH, v, max_batch=8)
    ga, _ = a.forward_backward(x[:8], y[:8], [0.0], [0.0], [800], [2])
    gb, _ = b.forward_backward(x[8:], y[8:], [0.0], [0.0], [800], [2])
    g_local = one.raw_to_variables(((ga + gb) / 2).cpu().numpy())
    worst = max(np.abs(g_local[n] - ref[n]).max() / max(np.abs(ref[n]).max(), 1e-6 * gmax) for n in ref
                if is_trainable(n) and not n.endswith(("l_1/b", "l_2/b")))
    assert worst > 2e-3, worst        # ten times the tolerance the synchronised gradient meets
    # BN running statistics after the first step: moved by the GLOBAL moments (later steps inherit Adam's sign
    # sensitivity on the analytically-zero bias gradients, in one process as in two)
    r1 = one.raw_params()
    q0 = load("params1", 0)
    mask = np.zeros(one.n_params, bool)
    pos = 0
    for L in one.layers:
        for nm in P.layer_variable_names(L, one._tmpl):
            n = 1 if nm is None else int(np.asarray(v[nm]).size)
            mask[pos:pos + n] = nm is not None and ("bn_nvp_conv" in nm)
            pos += n
    assert mask.any() and np.abs(q0[mask] - r1[mask]).max() <= 1e-5 * np.abs(r1[mask]).max()
    assert np.abs(q0[mask] - np.asarray(P.pack_layers(one.layers, v, 4, one._tmpl)[2])[mask]).max() > 1e-3   # they did move


def _unequal_worker(rank, world, port, outdir):
    import sys
    import torch.distributed as dist
    from conftest import ROOT, make_inputs, trained_like_variables
    sys.path.insert(0, ROOT)
    from noise_flow_amd import default_hps
    from noise_flow_amd.train import Trainer
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    tr = Trainer([16, 16, 4], default_hps(arch=DP_ARCH), variables=trained_like_variables(DP_ARCH, 4, seed=9), max_batch=8)
    n = 8 - rank                                        # 8 patches on rank 0, 7 on rank 1
    x, y = make_inputs(n, 16, 16, seed=3 + rank, b1=0.003696)
    msg = ""
    try:
        tr.step(x, y, [0.0], [0.0], [800], [2], lr=1e-3, group=True, sync_bn=True, sync=True)
    except ValueError as e:
        msg = str(e)
    open(os.path.join(outdir, "unequal_%d.txt" % rank), "w").write(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_sync_bn_rejects_unequal_shards(tmp_path):
    """Synchronised BN forms the global moments with world x the LOCAL pixel count: ranks that feed different numbers of
    patches would silently get wrong moments and gradients — the step reports it instead (one 16-byte MAX all-reduce)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_unequal_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    [p.start() for p in procs]
    [p.join(timeout=300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for r in range(2):
        msg = open(os.path.join(str(tmp_path), "unequal_%d.txt" % r)).read()
        assert "same number of patches on every rank" in msg and "between 7 and 8" in msg, msg


@pytest.mark.parametrize("width,hw,per", [(32, 16, 6), (96, 16, 6), (32, 32, 3)])
def test_two_rank_sync_bn_at_width_32(tmp_path, width, hw, per):
    """The same property on the matrix-core stage kernels (width 32) and on the library-GEMM path of the widths beyond (96;
    csrc/nf_train_gemm.h): behind a cross-rank all-reduce the statistics are finalised by k_bn_fin / k_bnb_fin from the scattered
    totals; 2 ranks x 6 patches of 16x16 = 1 rank on the 12.  (32, 32, 3): the patch-resident stages of csrc/nf_train_pr.h, whose
    consumers finalise the moments themselves from the two slots the global totals come back in."""
    import socket
    import torch.multiprocessing as mp
    from oracle.nf_grad_oracle import is_trainable
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, str(tmp_path), width, hw, per)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(timeout=300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    load = lambda n, r: np.load(os.path.join(str(tmp_path), "sync_%s_%d.npy" % (n, r)))   # noqa: E731
    g0, g1 = load("grad", 0), load("grad", 1)
    assert np.array_equal(g0, g1) and np.array_equal(load("params", 0), load("params", 1))
    v = trained_like_variables(DP_ARCH, width, seed=9)
    x, y = make_inputs(2 * per, hw, hw, seed=500, b1=0.003696)
    one = _trainer(DP_ARCH, v, (hw, hw, 4), width, max_batch=2 * per)
    g_one, loss_one = one.forward_backward(x, y, [0.0], [0.0], [800], [2])
    ref, got = one.raw_to_variables(g_one.cpu().numpy().copy()), one.raw_to_variables(g0)
    gmax = max(np.abs(ref[n]).max() for n in ref if is_trainable(n))
    tol = 8.0 / (2 * per * hw * hw)   # a few pixels' worth: ReLU kinks under a different summation order (wide couplings)
    for name in ref:
        if is_trainable(name) and not name.endswith(("l_1/b", "l_2/b")):
            assert np.abs(got[name] - ref[name]).max() <= tol * max(np.abs(ref[name]).max(), 1e-6 * gmax), name
    l2 = 0.5 * (load("loss", 0)[0] + load("loss", 1)[0])
    assert abs(l2 - float(loss_one.cpu().numpy()[0])) <= 1e-5 * abs(l2)
    # per-rank statistics on the same shards: a different gradient
    a, b = _trainer(DP_ARCH, v, (hw, hw, 4), width, max_batch=per), _trainer(DP_ARCH, v, (hw, hw, 4), width, max_batch=per)
    ga, _ = a.forward_backward(x[:per], y[:per], [0.0], [0.0], [800], [2])
    gb, _ = b.forward_backward(x[per:], y[per:], [0.0], [0.0], [800], [2])
    g_local = one.raw_to_variables(((ga + gb) / 2).cpu().numpy())
    worst = max(np.abs(g_local[n] - ref[n]).max() / max(np.abs(ref[n]).max(), 1e-6 * gmax) for n in ref
                if is_trainable(n) and not n.endswith(("l_1/b", "l_2/b")))
    assert worst > 4 * tol, worst


def test_gradients_with_the_reversed_template_binding():
    """binding='sample_first' (quirk Q1): the coupling CNN variables are bound to the layers in reversed
    order; trainer and oracle must agree on which template each gradient belongs to."""
    from noise_flow_amd import default_hps
    from noise_flow_amd.train import Trainer
    from oracle.nf_grad_oracle import GradOracle
    arch = "sdn5|unc|unc|gain4|unc"
    v = trained_like_variables(arch, 4, seed=21)
    x, y = make_inputs(5, seed=23, b1=0.003696)
    tr = Trainer([32, 32, 4], default_hps(arch=arch), variables=v, binding="sample_first", max_batch=8)
    grads, loss = tr.forward_backward(x, y, [0.0], [0.0], [800], [2])
    ref_loss, ref_sd, ref_grads, _ = GradOracle(arch, v, binding="sample_first").loss_and_grads(x, y, 800, 2)
    lv = loss.cpu().numpy()
    assert abs(lv[0] - ref_loss) <= 1e-5 * abs(ref_loss)
    _check_grads(tr, grads, ref_grads)
    # and the two bindings really differ
    other = GradOracle(arch, v, binding="loss_first").loss_and_grads(x, y, 800, 2)[0]
    assert abs(other - ref_loss) > 1e-3 * abs(ref_loss)


def test_training_golden_fixture(shipped_variables):
    """tests/golden/train_step_shipped.npz (made by tools/make_golden_train.py from the fp64 autograd
    oracle): loss, every gradient tensor, the BN EMA and one Adam step of the shipped model."""
    from conftest import ROOT
    from oracle.nf_grad_oracle import is_trainable
    g = np.load(os.path.join(ROOT, "tests", "golden", "train_step_shipped.npz"))
    x, y, iso, cam, lr = g["x"], g["y"], float(g["iso"]), float(g["cam"]), float(g["lr"])
    tr = _trainer(FULL_ARCH, shipped_variables)
    grads, loss = tr.forward_backward(x, y, [0.0], [0.0], [iso], [cam])
    lv = loss.cpu().numpy()
    assert abs(lv[0] - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert abs(lv[1] - float(g["sd_z"])) <= 1e-5 * float(g["sd_z"])
    ref_grads = {k[len("grad/"):]: g[k] for k in g.files if k.startswith("grad/")}
    assert _check_grads(tr, grads, ref_grads) == 2431
    v = tr.variables
    for k in g.files:
        if k.startswith("bn/"):
            want = g[k]
            assert np.abs(v[k[3:]] - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-3), k
    tr.apply(lr)
    after = tr.variables
    for k, gr in ref_grads.items():
        if not is_trainable(k) or k.endswith("/b") or ("adam/" + k) not in g.files or k not in after:
            continue
        want = g["adam/" + k].reshape(-1)
        got = np.asarray(after[k], np.float32).reshape(-1)
        if got.shape != want.shape:
            continue                                         # unused rescaling_scale of sdn_0 / gain_5
        solid = np.abs(gr.reshape(-1)) > 1e-3 * max(np.abs(gr).max(), 1e-30)   # well above the fp32 noise floor
        if solid.any():
            # first Adam step = lr * g / (|g| + eps'): the bulk must move exactly like the oracle's
            assert np.abs(got[solid] - want[solid]).max() <= 0.02 * lr + 1e-6 * np.abs(want[solid]).max(), k


def test_wide_training_golden_fixture():
    """tests/golden/train_step_width48.npz (tools/make_golden_train.py wide, from the fp64 autograd oracle; the model is part of
    the fixture): loss, sd_z, every gradient tensor (2e-4 of its scale or the fixture's round-off allowance of the entry) and the
    BN EMA of a step at width 48 — the trainer's library-GEMM path (csrc/nf_train_gemm.h)."""
    from conftest import ROOT, GRAD_NOISE_C
    from oracle.nf_grad_oracle import is_trainable
    from noise_flow_amd import params as P
    g = np.load(os.path.join(ROOT, "tests", "golden", "train_step_width48.npz"))
    arch, width = str(g["arch"]), int(g["width"])
    v = {k[4:]: g[k] for k in g.files if k.startswith("var/")}
    x, y = g["x"], g["y"]
    tr = _trainer(arch, v, (x.shape[1], x.shape[2], 4), width)
    grads, loss = tr.forward_backward(x, y, [0.0], [0.0], [float(g["iso"])], [float(g["cam"])])
    lv = loss.cpu().numpy()
    assert abs(lv[0] - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert abs(lv[1] - float(g["sd_z"])) <= 1e-5 * float(g["sd_z"])
    got = tr.raw_to_variables(grads.cpu().numpy())
    names = [nm for L in tr.layers for nm in P.layer_variable_names(L, tr._tmpl) if nm is not None and is_trainable(nm)]
    gmax = max(np.abs(g["grad/" + nm]).max() for nm in names)
    n = 0
    for nm in names:
        ref = np.asarray(g["grad/" + nm], np.float64)
        allow = GRAD_NOISE_C * 2.0 ** -24 * np.asarray(g["abs/" + nm], np.float64).reshape(ref.shape)
        a = np.asarray(got[nm], np.float64).reshape(ref.shape)
        if nm.endswith("l_1/b") or nm.endswith("l_2/b"):
            assert (np.abs(a) <= np.maximum(1e-5 * gmax, allow)).all(), nm
        else:
            assert (np.abs(a - ref) <= np.maximum(GRAD_RTOL * max(np.abs(ref).max(), 1e-6 * gmax), allow)).all(), (nm, np.abs(a - ref).max())
        n += ref.size
    assert n > 10000
    after = tr.variables
    for k in g.files:
        if k.startswith("bn/"):
            want = g[k]
            assert np.abs(np.asarray(after[k[3:]]).reshape(want.shape) - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-3), k


def test_variables_shared_between_layers_receive_the_summed_gradient():
    """The reference creates the sdn / gain parameters under an AUTO_REUSE scope: arch ``gain4|unc|gain4`` has ONE gain_val
    and ``sdn4|unc|sdn4`` one beta1 / beta2 / gain_params.  The raw layout holds a slot per layer; the trainer sums the
    slots' gradients (found by tests/test_gpu_random_sweep.py) and the copies stay identical through the update."""
    arch = "sdn4|unc|gain4|sdn4|gain4"
    v = trained_like_variables(arch, 4, seed=12)
    x, y = make_inputs(5, 16, 16, seed=9)
    tr = _trainer(arch, v, (16, 16, 4), 4)
    grads, loss = tr.forward_backward(x, y, [0.0], [0.0], [800], [2])
    ref_loss, _, ref_grads, _ = _grad_oracle(arch, v).loss_and_grads(x, y, 800, 2)
    assert abs(loss.cpu().numpy()[0] - ref_loss) <= 1e-5 * abs(ref_loss)
    _check_grads(tr, grads, ref_grads)
    for _ in range(3):
        tr.step(x, y, [0.0], [0.0], [800], [2], lr=1e-3)
    raw = tr.raw_params()
    from noise_flow_amd import params as P
    pos, seen = 0, {}
    for L in tr.layers:
        for nm in P.layer_variable_names(L, tr._tmpl):
            n = 1 if nm is None else int(np.asarray(v[nm]).size)
            if nm is not None:
                if nm in seen:
                    np.testing.assert_array_equal(raw[pos:pos + n], seen[nm])
                seen[nm] = raw[pos:pos + n].copy()
            pos += n
    assert not np.array_equal(seen["model/sdn_gain/gain_val"], np.asarray(v["model/sdn_gain/gain_val"]).reshape(-1))


@pytest.mark.parametrize("width", [4, 64])
def test_a_training_run_is_reproducible_bit_for_bit(shipped_variables, width):
    """Gradients, BN moments and updates come from slot reductions in a fixed order; on patches of a multiple of 64 pixels
    the reported loss / sd_z do too (one partial per wavefront, summed in order).  Two trainers fed the same minibatches
    end in identical parameters and identical logged values, to the bit — also on the matrix-core GEMM path (width 64, csrc/nf_train_mm.h:
    batch sums as slotted partials, filter gradients as pixel-chunk partial products added up in a fixed order, no atomics)."""
    import torch
    outs = []
    for rep in range(2):
        tr = _trainer(FULL_ARCH, shipped_variables if width == 4 else trained_like_variables(FULL_ARCH, width, seed=3), width=width, max_batch=12)
        log = []
        for step in range(4):
            x, y = make_inputs(12, seed=60 + step, b1=0.003696)
            loss = tr.step(x, y, [0.0], [0.0], [800], [2], lr=1e-3, sync=False)
            log.append(loss.clone())
        torch.cuda.synchronize()
        outs.append((tr.raw_params(), torch.stack(log).cpu().numpy()))
        tr.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32))


@pytest.mark.parametrize("arch,width,hw,B", [(FULL_ARCH, 4, (32, 32), 9),          # 1024 threads per patch
                                             ("sdn5|unc|unc|gain4|unc", 8, (16, 24), 6),   # 512, width 8
                                             ("unc|unc|unc", 4, (10, 12), 11),      # 256, threads beyond the patch idle
                                             ("unc|sdn5|unc|unc", 8, (32, 32), 3),  # a coupling chain cut by another layer
                                             ("unc|gain4|unc|unc|gain4", 4, (8, 8), 40)])
def test_tiled_stages_and_layer_kernels_agree(arch, width, hw, B, monkeypatch):
    """The trainer runs couplings of width <= 8 on patches of <= 1024 pixels through one-workgroup-per-patch tiled stages
    (two launches per coupling and pass) and everything else through one kernel per layer stage; NF_TRAIN_TILED selects
    (bit 0 backward, bit 1 forward).  Both must give the oracle's loss and gradients; between themselves they only differ
    in how the fp32 partial sums are grouped, so the logged loss agrees to 1e-6 and gradients to 1e-4 of their scale."""
    v = trained_like_variables(arch, width, seed=8)
    x, y = make_inputs(B, hw[0], hw[1], seed=23)
    res = {}
    for mode in ("0", "1", "2", "3"):
        monkeypatch.setenv("NF_TRAIN_TILED", mode)
        tr = _trainer(arch, v, (hw[0], hw[1], 4), width)
        grads, loss = tr.forward_backward(x, y, [0.0], [0.0], [800], [2])
        res[mode] = (grads.cpu().numpy().copy(), loss.cpu().numpy().copy(), tr.raw_params())
        if mode == "3":
            ref_loss, ref_sd, ref_grads, _ = _grad_oracle(arch, v).loss_and_grads(x, y, 800, 2)
            assert abs(res[mode][1][0] - ref_loss) <= 1e-5 * abs(ref_loss)
            _check_grads(tr, grads, ref_grads)
        tr.close()
    g0, l0, p0 = res["0"]
    for mode in ("1", "2", "3"):
        g, l, p = res[mode]
        assert np.allclose(l, l0, rtol=1e-6, atol=0), (mode, l, l0)
        assert np.abs(g - g0).max() <= 1e-4 * np.abs(g0).max(), (mode, np.abs(g - g0).max(), np.abs(g0).max())
        assert np.allclose(p, p0, rtol=1e-5, atol=1e-7), mode      # the BN running moments moved by the forward pass


@pytest.mark.parametrize("arch,width,hw,B", [("sdn5|unc|unc|gain4|unc", 32, (32, 32), 5),
                                             ("unc|unc", 32, (20, 12), 7),          # ragged: tiles of 32 pixels straddle rows
                                             ("unc", 32, (5, 7), 3),                # a patch smaller than one 64-pixel step
                                             ("unc|gain4|unc", 32, (64, 64), 2),    # per-patch operand tiles too large for LDS
                                             ("sdn5|unc|unc", 16, (16, 24), 6),     # width 16: the same templates
                                             ("unc|unc", 32, (8, 8), 1100)])        # more patches than slots: the persistent loops
def test_wide_matrix_core_stages_and_layer_kernels_agree(arch, width, hw, B, monkeypatch):
    """At width 32 the l_2 forward / backward stages and the three filter gradients run as GEMMs on v_mfma_f32_32x32x2_f32
    (exact fp32), the slot sums are added up by one-pass finaliser kernels and the BN1 backward walks the tensors flat;
    NF_TRAIN_WIDE_MFMA=0 keeps the one-kernel-per-layer-stage path.  Both give the oracle's gradients and differ from
    each other only by summation order."""
    v = trained_like_variables(arch, width, seed=9)
    x, y = make_inputs(B, hw[0], hw[1], seed=29)
    res = {}
    for mode in ("0", "4095"):
        monkeypatch.setenv("NF_TRAIN_WIDE_MFMA", mode)
        tr = _trainer(arch, v, (hw[0], hw[1], 4), width, max_batch=max(64, B))
        grads, loss = tr.forward_backward(x, y, [0.0], [0.0], [400], [1])
        res[mode] = (grads.cpu().numpy().copy(), loss.cpu().numpy().copy(), tr.raw_params())
        if mode == "4095":
            _oracle_check_next_to_kinks(tr, arch, v, x, y, 400, 1, width, rtol=1e-3)
        tr.close()
    (g0, l0, p0), (g1, l1, p1) = res["0"], res["4095"]
    assert np.allclose(l1, l0, rtol=1e-6, atol=0), (l1, l0)
    # other summation order only — unless an activation within round-off of its ReLU kink takes the other branch in one of
    # the two paths: a few pixels' worth of gradient (the bound of tests/test_gpu_random_sweep.py for wide couplings)
    tol = max(1e-5, min(8.0 / (B * hw[0] * hw[1]), 2e-2))
    assert np.abs(g1 - g0).max() <= tol * np.abs(g0).max(), (np.abs(g1 - g0).max(), np.abs(g0).max())
    assert np.allclose(p1, p0, rtol=1e-5, atol=1e-5)      # the BN running moments moved by the forward pass


def test_wide_filter_gradients_fused_into_the_stage_kernels(monkeypatch):
    """From ~400 patches of 32x32 up (stage kernels that fill the GPU) the width-32 trainer accumulates d l_2/W and d l_last/W
    inside the backward stage kernels that already hold both operands in LDS (NF_TRAIN_WIDE_MFMA bit 7) instead of in their
    own kernels on the side stream.  Same sums, other grouping: the three paths agree to 1e-5 of the gradient scale."""
    arch, width, B = "unc|unc", 32, 416
    v = trained_like_variables(arch, width, seed=10)
    x, y = make_inputs(B, 32, 32, seed=31)
    res = {}
    monkeypatch.setenv("NF_TRAIN_PR", "0")   # the stage kernels of nf_train_wide.h (32x32 patches run on nf_train_pr.h by default)
    for mode in ("0", "127", "511", "4095"):
        monkeypatch.setenv("NF_TRAIN_WIDE_MFMA", mode)
        tr = _trainer(arch, v, (32, 32, 4), width, max_batch=B)
        grads, loss = tr.forward_backward(x, y, [0.0], [0.0], [800], [2])
        res[mode] = (grads.cpu().numpy().copy(), loss.cpu().numpy().copy())
        tr.close()
    g0, l0 = res["0"]
    for mode in ("127", "511", "4095"):
        g, l = res[mode]
        assert np.allclose(l, l0, rtol=1e-6, atol=0), (mode, l, l0)
        assert np.abs(g - g0).max() <= 1e-5 * np.abs(g0).max(), (mode, np.abs(g - g0).max(), np.abs(g0).max())
    assert not np.array_equal(res["127"][0], res["4095"][0])      # the fused path really is another code path


def test_round_off_allowance_where_a_gradient_cancels():
    """The one place the gradient comparison needs more than a relative tolerance, and what grants it: a gain layer AT its
    optimum (gain_val solved for on the fp64 oracle), where d loss / d gain = 0.016 is the remainder of per-element terms that
    add up to 6.8e5 in magnitude.  No fp32 evaluation resolves that to 2e-4 of itself; it is resolved to a few 2^-24 of the
    terms.  The allowance (conftest.py::grad_noise_allowance) is computed by the fp64 oracle alone.  Checked here: both kernel
    paths of the trainer are within it, so is their distance from each other (two summation orders of the same arithmetic), and
    it is not vacuous — within three orders of magnitude of the distances actually observed."""
    from conftest import grad_noise_allowance, other_kernel_path_gradients
    arch, key = "sdn5|gain4|unc", "model/sdn_gain/gain_val"
    v = trained_like_variables(arch, 4, seed=2)
    v[key] = np.asarray([0.016486794], np.float32)
    x, y = make_inputs(8, 32, 32, seed=5)
    o = _grad_oracle(arch, v)
    ref_loss, _, ref_grads, _ = o.loss_and_grads(x, y, 800, 2)
    ref = float(np.asarray(ref_grads[key]).reshape(-1)[0])
    terms = float(np.asarray(o.grad_abs_terms[key]).reshape(-1)[0])
    allow = float(grad_noise_allowance(o, key).reshape(-1)[0])
    assert abs(ref) < 1e-6 * terms and allow < 2e-6 * terms           # the case is what it is meant to be
    tr = _trainer(arch, v, (32, 32, 4), 4)
    grads, loss = tr.forward_backward(x, y, [0.0], [0.0], [800], [2])
    a = float(np.asarray(tr.raw_to_variables(grads.cpu().numpy())[key]).reshape(-1)[0])
    b = float(np.asarray(other_kernel_path_gradients(lambda: _trainer(arch, v, (32, 32, 4), 4), x, y, 800, 2)[key]).reshape(-1)[0])
    assert abs(loss.cpu().numpy()[0] - ref_loss) <= 1e-5 * abs(ref_loss)
    seen = max(abs(a - ref), abs(b - ref), abs(a - b))
    assert abs(a - ref) <= allow and abs(b - ref) <= allow and abs(a - b) <= allow, (a, b, ref, allow)
    assert seen > 0 and allow <= 1000.0 * seen, (a, b, ref, allow, seen)
    _check_grads(tr, grads, ref_grads, oracle=o)                      # and every other tensor of the step at the usual tolerance


@pytest.mark.parametrize("arch,width,hw,B,iso,cam", [("sdn5|unc|gain4|unc", 128, (8, 8), 3, 800, 2),
                                                     ("unc|unc", 64, (12, 10), 4, 100, 1),
                                                     ("sdn5|unc|unc|gain4|unc", 50, (9, 7), 5, 400, 0),      # not a multiple of 4
                                                     ("unc", 512, (6, 6), 2, 1600, 3),                        # the reference's default width
                                                     ("unc|unc", 96, (32, 32), 6, 800, 2),
                                                     ("sdn5|unc|unc|gain4", 24, (10, 12), 4, 800, 2),        # between two kernel widths
                                                     ("unc|unc", 5, (8, 8), 3, 100, 0),
                                                     ("|".join(["unc"] * 17), 12, (6, 6), 3, 400, 1),       # more couplings than one store / reduce launch holds
                                                     ("unc|unc", 36, (64, 64), 2, 800, 2),                  # 64x64 patches; a ragged 64-channel tile
                                                     ("unc", 200, (16, 24), 1, 100, 0),                     # one patch; two channel tiles, the second ragged
                                                     ("sdn5|unc|gain4", 132, (7, 5), 3, 1600, 3),           # 35 pixels per patch; 128 + 4 channels
                                                     ("unc", 2, (8, 6), 3, 3200, 4)])    # (width 1: torch's CPU conv backward refuses the oracle's graph)
def test_gradients_at_coupling_widths_beyond_32(arch, width, hw, B, iso, cam):
    """sidd/ArgParser.py:43 defaults --width to 512 and train_noise_flow.py:50-77 trains at whatever width is set: beyond 32 the
    step's dense products are this repo's fp32 matrix-core GEMMs (csrc/nf_train_mm.h: pixels on M resp. on K, BN + ReLU and the BN
    backward formed while the operands are staged, batch sums in the epilogues) between kernels over [pixel][w] tensors of run-time
    width (csrc/nf_train_gemm.h) — and so are the widths below 32 that have no kernels of their own (1 .. 31 except 4, 8, 16):
    the trainer refuses nothing the reference's flag accepts up to 512.  Loss, sd_z, every gradient tensor and the BN running statistics against the fp64 autograd
    oracle; one Adam step against the float32 restatement of TF's update rule."""
    from oracle.nf_grad_oracle import adam_step
    v = trained_like_variables(arch, width, seed=width)
    for k in v:     # activations of O(1) at every width (the helper's weights are tuned for width 4)
        if k.endswith("l_2/W") or k.endswith("l_last/W"):
            v[k] = (v[k] * np.float32((4.0 / width) ** 0.5)).astype(np.float32)
    x, y = make_inputs(B, hw[0], hw[1], seed=19)
    tr = _trainer(arch, v, (hw[0], hw[1], 4), width)
    _oracle_check_next_to_kinks(tr, arch, v, x, y, iso, cam, width, rtol=1e-3)
    o = _grad_oracle(arch, v)
    _, _, ref_grads, new_running = o.loss_and_grads(x, y, iso, cam)
    got = tr.variables
    for k, want in new_running.items():          # BN running statistics moved by the EMA of the batch moments (layers.py:392-393)
        assert np.abs(np.asarray(got[k], np.float64).reshape(want.shape) - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-3), k
    tr.step(x, y, [0.0], [0.0], [iso], [cam], lr=1e-3)
    after = tr.variables
    want = adam_step({k: np.asarray(a, np.float32) for k, a in v.items()}, ref_grads, {}, 1e-3, dtype=np.float32)
    for k in ref_grads:
        if k.endswith("l_1/b") or k.endswith("l_2/b"):
            continue                             # analytically zero gradients: Adam normalises round-off to +-lr
        a, b = np.asarray(after[k], np.float64).reshape(-1), np.asarray(want[k], np.float64).reshape(-1)
        moved = np.abs(b - np.asarray(v[k], np.float64).reshape(-1)) > 0
        assert (np.abs(a - b)[moved] <= 2.5e-3).all(), k       # a step is +-lr wherever the gradient is resolved; sign flips only at its noise floor
