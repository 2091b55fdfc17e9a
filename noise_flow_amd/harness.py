"""Evaluation / sampling epoch loops and loggers — the callers either side of the
hot path, with the reference's call pattern and on-disk formats.

* ``ResultLogger``  — tab-separated ``train.txt / test.txt / sample.txt``
  (reference ``borealisflows/utils.py:90-107``): header row without a trailing
  newline, every record written as ``"\\n" + "\\t".join(values)``.
* ``test_epoch``    — ``test_multithread`` / ``test_thread``
  (reference ``train_noise_flow.py:80-117, 386-407``): one ``loss`` call per
  minibatch dict, epoch NLL = mean over minibatches of the per-minibatch means
  (quirk Q12), sd_z likewise.
* ``sample_epoch``  — ``sample_thread`` (reference ``train_noise_flow.py:139-184``):
  fixed ISO 100 / camera S6, sample, NLL of the sample, and the reference's marginal-KL
  recipe ``calc_kldiv_mb`` (``sidd/sidd_utils.py:995-1058``) → KLD_G / KLD_NLF / KLD_NF / KLD_R.

* ``train_epoch`` / ``fit`` — ``train_multithread`` / ``train_thread`` and the epoch loop of
  ``main`` (reference ``train_noise_flow.py:27-77, 379-511``): one training step per minibatch
  dict, epoch loss = mean of the per-minibatch losses, test / sampling on the reference's epoch
  schedule, ``model.ckpt-<epoch>`` + ``model.ckpt.best`` checkpoints, the three TSV logs.
  The reference runs its 16 training threads Hogwild-style on one session; here the steps of an
  epoch are enqueued in order on one stream (deterministic).

Minibatch dicts follow ``noise_flow_amd.patches.make_minibatch``.  Threads are
optional (``n_threads``): the HIP handle is re-entrant like the shared tf.Session.
"""
from __future__ import annotations

import queue
import threading
import time
from typing import Iterable, List, Sequence

import numpy as np



def _np(a) -> np.ndarray:
    """numpy view of a minibatch field (numpy array or torch tensor on any device)."""
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


class ResultLogger(object):
    def __init__(self, path, columns, append=False):
        self.columns = list(columns)
        mode = "a" if append else "w"
        self.f_log = open(path, mode)
        if mode == "w":
            self.f_log.write("\t".join(self.columns))

    def log(self, run_info):
        self.f_log.write("\n")
        self.f_log.write("\t".join("{0}".format(run_info[c]) for c in self.columns))
        self.f_log.flush()

    def close(self):
        if not self.f_log.closed:
            self.f_log.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


TRAIN_COLUMNS = ["epoch", "NLL", "NLL_G", "NLL_SDN", "sdz", "train_time"]                # train_noise_flow.py:346
TEST_COLUMNS = ["epoch", "NLL", "NLL_G", "NLL_SDN", "sdz", "msg"]                       # train_noise_flow.py:340
SAMPLE_COLUMNS = ["epoch", "NLL", "NLL_G", "NLL_SDN", "sdz", "sample_time",             # train_noise_flow.py:344-348
                  "KLD_G", "KLD_NLF", "KLD_NF", "KLD_R"]


def _run_threads(worker, items: Sequence, n_threads: int) -> List:
    if n_threads <= 1:
        return [worker(it) for it in items]
    q: "queue.Queue" = queue.Queue()
    for i, it in enumerate(items):
        q.put((i, it))
    out = [None] * len(items)
    errs = []

    def loop():
        while True:
            try:
                i, it = q.get_nowait()
            except queue.Empty:
                return
            try:
                out[i] = worker(it)
            except Exception as e:  # pragma: no cover
                errs.append(e)

    ths = [threading.Thread(target=loop) for _ in range(n_threads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    if errs:
        raise errs[0]
    return out


def test_epoch(nf, minibatches: Iterable[dict], n_threads: int = 1):
    """→ (mean over minibatches of batch-mean NLL, mean sd_z, per-batch losses)."""
    mbs = list(minibatches)

    def one(mb):
        loss, sd_z = nf.loss(mb["_x"], mb["_y"], mb["nlf0"], mb["nlf1"], mb["iso"], mb["cam"])
        return float(loss), float(sd_z)

    res = _run_threads(one, mbs, n_threads)
    losses = [r[0] for r in res]
    return float(np.mean(losses)), float(np.mean([r[1] for r in res])), losses


S6_NLF = {100: (0.000479, 0.000002), 400: (0.001774, 0.000002), 800: (0.003696, 0.000002),
          1600: (0.008211, 0.000002), 3200: (0.019930, 0.000002)}    # train_noise_flow.py:146-147


def sample_epoch(nf, minibatches: Iterable[dict], temp: float = 1.0, fix_iso: float = 100.0, fix_cam: float = 2.0,
                 n_threads: int = 1, sc_sd: float = 1.0, vis_dir=None, seed=None):
    """``sample_multithread`` / ``sample_thread`` (train_noise_flow.py:119-184) → dict(NLL, sdz, KLD_G, KLD_NLF, KLD_NF,
    KLD_R, sample_time).  Per minibatch: sample with the camera / ISO FIXED to S6 / 100 (``is_fix``; the fed ``nlf1`` is the
    reference's ``nlf_s6[..][0]``, quirk Q4 — inert for sdn5), NLL of the sample under the model, and the marginal-KL
    recipe ``calc_kldiv_mb`` (every 5th patch, four noise models against the minibatch's real noise; ``sc_sd`` = the
    training set's noise standard deviation, ``pat_stats['sc_in_sd']``).  The KL draws use the global numpy RNG, like the
    reference; they are taken on the calling thread in minibatch order after the device work, so a seeded epoch is
    reproducible for any ``n_threads`` (``seed``: seed that RNG first)."""
    from .metrics import calc_kldiv_mb
    mbs = list(minibatches)
    nlf0 = S6_NLF.get(int(fix_iso), S6_NLF[100])[0]
    nlf1 = nlf0                                                  # train_noise_flow.py:158-159 (index 0 twice)
    t0 = time.time()

    def one(mb):
        y = mb["_y"]
        xs = nf.sample(y, temp, y, [nlf0], [nlf1], [fix_iso], [fix_cam])
        loss, sd_z = nf.loss(xs, y, [nlf0], [nlf1], [fix_iso], [fix_cam])
        return _np(xs), float(loss), float(sd_z)

    res = _run_threads(one, mbs, n_threads)
    if seed is not None:
        np.random.seed(seed)
    kld = np.zeros(4)
    for mb, r in zip(mbs, res):
        host = dict(mb, _x=_np(mb["_x"]), _y=_np(mb["_y"]), pid=mb.get("pid", np.arange(len(r[0]))), fn=mb.get("fn", ""))
        kld += calc_kldiv_mb(host, r[0], vis_dir, sc_sd)
    kld /= max(len(mbs), 1)
    return {"NLL": float(np.mean([r[1] for r in res])), "sdz": float(np.mean([r[2] for r in res])),
            "KLD_G": float(kld[0]), "KLD_NLF": float(kld[1]), "KLD_NF": float(kld[2]), "KLD_R": float(kld[3]),
            "sample_time": time.time() - t0}


def train_epoch(trainer, minibatches: Iterable[dict], lr: float, group=None, sync_bn: bool = False):
    """``train_multithread`` (train_noise_flow.py:27-77, 484-504) → (mean over minibatches of the
    training loss, mean sd_z, per-batch losses).  Steps are enqueued back to back; the losses are
    read once at the end of the epoch (one synchronisation per epoch, not per step)."""
    outs = []
    for mb in minibatches:
        out = trainer.step(mb["_x"], mb["_y"], mb["nlf0"], mb["nlf1"], mb["iso"], mb["cam"], lr=lr, group=group,
                           sync=False, sync_bn=sync_bn)
        outs.append(out.clone())
    if not outs:
        return float("nan"), float("nan"), []
    vals = np.stack([o.cpu().numpy() for o in outs]).astype(np.float64)
    return float(vals[:, 0].mean()), float(vals[:, 1].mean()), [float(v) for v in vals[:, 0]]


def _is_eval_epoch(epoch: int, epochs_full_valid: int) -> bool:
    """train_noise_flow.py:386-387."""
    return epoch < 10 or (epoch < 100 and epoch % 10 == 0) or epoch % epochs_full_valid == 0


def fit(trainer, nf_eval, train_mbs: Sequence[dict], test_mbs: Sequence[dict], logdir: str, epochs: int, lr: float,
        epochs_full_valid: int = 10, nll_gauss: float = 0.0, nll_sdn: float = 0.0, do_sampling: bool = True,
        start_epoch: int = 1, group=None, log=None, sync_bn: bool = False, sc_sd: float = 1.0):
    """The epoch loop of ``train_noise_flow.py:379-511``: per epoch test (on the reference's
    schedule; saves ``ckpt/model.ckpt-<epoch>`` and ``ckpt/model.ckpt.best``), sampling, training;
    appends to ``train.txt`` / ``test.txt`` / ``sample.txt`` under ``logdir``.

    ``trainer``: :class:`noise_flow_amd.train.Trainer`; ``nf_eval``: an eval-mode
    :class:`NoiseFlow` of the same architecture (refreshed from the trainer before each test /
    sampling epoch).  → dict of the per-epoch result lists."""
    import os
    os.makedirs(os.path.join(logdir, "ckpt"), exist_ok=True)
    ckpt_path = os.path.join(logdir, "ckpt", "model.ckpt")
    append = start_epoch > 1
    train_logger = ResultLogger(os.path.join(logdir, "train.txt"), TRAIN_COLUMNS, append)
    test_logger = ResultLogger(os.path.join(logdir, "test.txt"), TEST_COLUMNS, append)
    sample_logger = ResultLogger(os.path.join(logdir, "sample.txt"), SAMPLE_COLUMNS, append)
    res = {"train": [], "test": [], "sample": []}
    best = float("inf")
    train_time = 0.0
    for epoch in range(start_epoch, epochs + 1):
        evaluate = _is_eval_epoch(epoch, epochs_full_valid)
        if evaluate:
            nf_eval.load_variables(trainer.variables)
            nll, sdz, _ = test_epoch(nf_eval, test_mbs)
            res["test"].append(nll)
            trainer.save("%s-%d" % (ckpt_path, epoch))
            is_best = int(nll < best)
            if is_best:
                best = nll
                trainer.save(ckpt_path + ".best")
            test_logger.log({"epoch": epoch, "NLL": nll, "NLL_G": nll_gauss, "NLL_SDN": nll_sdn, "sdz": sdz, "msg": is_best})
            if do_sampling:
                sr = sample_epoch(nf_eval, test_mbs, temp=1.0, sc_sd=sc_sd)
                res["sample"].append(sr["NLL"])
                sample_logger.log({"epoch": epoch, "NLL": sr["NLL"], "NLL_G": nll_gauss, "NLL_SDN": nll_sdn, "sdz": sr["sdz"],
                                   "sample_time": sr["sample_time"], "KLD_G": sr["KLD_G"], "KLD_NLF": sr["KLD_NLF"],
                                   "KLD_NF": sr["KLD_NF"], "KLD_R": sr["KLD_R"]})
        t = time.time()
        nll_tr, sdz_tr, _ = train_epoch(trainer, train_mbs, lr, group, sync_bn)
        train_time += time.time() - t
        res["train"].append(nll_tr)
        train_logger.log({"epoch": epoch, "train_time": int(train_time), "NLL": nll_tr, "NLL_G": nll_gauss, "NLL_SDN": nll_sdn,
                          "sdz": sdz_tr})
        if log is not None and evaluate:
            log("epoch %d  train %.4f  test %.4f  best %.4f" % (epoch, nll_tr, res["test"][-1], best))
    for lg in (train_logger, test_logger, sample_logger):
        lg.close()
    return res
