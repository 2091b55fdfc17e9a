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
  fixed ISO 100 / camera S6, sample, marginal KL vs the real noise, NLL of the
  sample.

Minibatch dicts follow ``noise_flow_amd.patches.make_minibatch``.  Threads are
optional (``n_threads``): the HIP handle is re-entrant like the shared tf.Session.
"""
from __future__ import annotations

import queue
import threading
import time
from typing import Iterable, List, Sequence

import numpy as np

from .metrics import kl_div_3_data, noise_bin_edges


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
                 n_threads: int = 1, kl_edges=None, seed=None):
    """→ dict(NLL, sdz, KLD_NF, KLD_NLF, sample_time): sample with fixed ISO / camera,
    marginal KL of the synthesised noise (and of a camera-NLF draw) against the real
    noise of the minibatch, NLL of the sample under the model."""
    mbs = list(minibatches)
    edges = noise_bin_edges() if kl_edges is None else kl_edges
    nlf0, nlf1 = S6_NLF.get(int(fix_iso), S6_NLF[100])
    rng = np.random.RandomState(0 if seed is None else seed)
    t0 = time.time()

    def one(mb):
        y = mb["_y"]
        xs = nf.sample(y, temp, y, [nlf0], [nlf1], [fix_iso], [fix_cam])
        kl_nf = kl_div_3_data(np.asarray(mb["_x"]).ravel(), np.asarray(xs).ravel(), edges)[0]
        loss, sd_z = nf.loss(xs, y, [nlf0], [nlf1], [fix_iso], [fix_cam])
        return kl_nf, float(loss), float(sd_z)

    res = _run_threads(one, mbs, n_threads)
    kl_nlf = []
    for mb in mbs:   # the camera-NLF baseline draw (host side, like kldiv_patch_set)
        y = np.asarray(mb["_y"], np.float64)
        nl = np.sqrt(nlf0 * y + nlf1) * rng.standard_normal(y.shape)
        kl_nlf.append(kl_div_3_data(np.asarray(mb["_x"]).ravel(), nl.ravel(), edges)[0])
    return {"NLL": float(np.mean([r[1] for r in res])), "sdz": float(np.mean([r[2] for r in res])),
            "KLD_NF": float(np.mean([r[0] for r in res])), "KLD_NLF": float(np.mean(kl_nlf)),
            "sample_time": time.time() - t0}
