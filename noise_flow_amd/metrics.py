"""Host-side metrics the reference's drivers compute around the hot path.

* marginal (histogram) KL divergences — ``kl_div_3_data`` / ``get_histogram``
  (reference ``sidd/sidd_utils.py:1247-1274``), used by ``sample_noise_flow.py:96``
  and ``train_noise_flow.py:171``;
* closed-form baseline NLLs — Gaussian and camera-NLF (signal-dependent) models
  (reference ``sidd/PatchStatsCalculator.py:92-123``; the ``NLL_G`` / ``NLL_SDN``
  columns of the reference's logs, ``hps.txt:71,118``).

Pure numpy; restated, not copied.
"""
from __future__ import annotations

import numpy as np


def get_histogram(data, bin_edges=None, left_edge=0.0, right_edge=1.0, n_bins=1000):
    """Normalised histogram (fractions of ALL samples, like the reference: values
    outside the edges are dropped from the counts but not from the divisor)."""
    width = (right_edge - left_edge) / n_bins
    if bin_edges is None:
        bin_edges = np.arange(left_edge, right_edge + width, width)
    data = np.asarray(data)
    counts, _ = np.histogram(data, bin_edges)
    return counts / data.size, bin_edges[:-1] + width / 2.0


def kl_div_3_data(p_data, q_data, bin_edges=None, left_edge=0.0, right_edge=1.0, n_bins=1000):
    """(forward, inverse, symmetric) KL divergence between two sample sets on a common
    histogram; bins empty in either set are skipped (sidd_utils.py:1255-1262)."""
    if bin_edges is None:
        width = (right_edge - left_edge) / n_bins
        bin_edges = np.arange(left_edge, right_edge + width, width)
    p, _ = get_histogram(p_data, bin_edges, left_edge, right_edge, n_bins)
    q, _ = get_histogram(q_data, bin_edges, left_edge, right_edge, n_bins)
    both = (p > 0) & (q > 0)
    p, q = p[both], q[both]
    lp, lq = np.log(p), np.log(q)
    fwd = float(np.sum(p * (lp - lq)))
    inv = float(np.sum(q * (lq - lp)))
    return fwd, inv, (fwd + inv) / 2.0


def kl_div_forward(p, q):
    """sum p log(p / q) over the bins where both histograms are finite and positive (sidd_utils.py:1202-1209)."""
    p, q = np.asarray(p), np.asarray(q)
    ok = ~(np.isnan(p) | np.isinf(p) | np.isnan(q) | np.isinf(q))
    p, q = p[ok], q[ok]
    ok = (p > 0) & (q > 0)
    p, q = p[ok], q[ok]
    return np.sum(p * np.log(p / q))


def kldiv_patch_set(i, mb, x_samples, sc_sd, subdir=None):
    """The four marginal KL divergences of ONE patch of a minibatch against its real noise — the sampling-epoch metric of
    the reference's training driver (``sidd_utils.py:1011-1058``): the noise of patch ``i`` under (0) an i.i.d. Gaussian of
    standard deviation ``sc_sd``, (1) the camera NLF ``sqrt(nlf0 y + nlf1)``, (2) the flow's sample ``x_samples[i]``, (3) the
    real noise itself (= 0), each unpacked to the Bayer mosaic and binned on ``[-1000, -0.1 : 0.2/64 : 0.1, 1000]``.
    The two random draws use the GLOBAL numpy RNG in the reference's order (Gaussian first, then the NLF draw).
    ``subdir``: also write the reference's per-patch ``.mat`` dumps there (scipy)."""
    from .patches import unpack_raw
    y = unpack_raw(np.asarray(mb['_y'])[i, :, :, :])
    nlf_sd = np.sqrt(mb['nlf0'] * y + mb['nlf1'])          # Camera NLF
    ng = np.random.normal(0, sc_sd, y.shape)               # Gaussian
    ns = unpack_raw(np.asarray(x_samples)[i, :, :, :])     # NF-sampled
    nl = nlf_sd * np.random.normal(0, 1, y.shape)          # Camera NLF
    n = unpack_raw(np.asarray(mb['_x'])[i, :, :, :])       # Real
    noise_pats_raw = (ng, nl, ns, n)
    bw = 0.2 / 64
    bin_edges = np.concatenate(([-1000.0], np.arange(-0.1, 0.1 + 1e-9, bw), [1000.0]), axis=0)
    hists = [get_histogram(pat, bin_edges=bin_edges)[0] for pat in noise_pats_raw]
    klds = np.asarray([kl_div_forward(hists[-1], h) for h in hists], np.float64)
    if subdir is not None:
        from scipy.io import savemat
        import os
        pid = mb['pid'][i]
        xs, xg, xl, x = (np.clip(y + v, 0.0, 1.0) for v in (ns, ng, nl, n))
        for name, val in (('y', y), ('ng', ng), ('nl', nl), ('ns', ns), ('n', n), ('xg', xg), ('xl', xl), ('xs', xs), ('x', x),
                          ('kl_ng', klds[0]), ('kl_nl', klds[1]), ('kl_ns', klds[2])):
            savemat(os.path.join(subdir, '%s_%04d.mat' % (name, pid)), {'x': val})
    return klds


def calc_kldiv_mb(mb, x_samples, vis_dir=None, sc_sd=1.0):
    """``sidd_utils.py:995-1008``: :func:`kldiv_patch_set` on every 5th patch of the minibatch, averaged →
    ``[KLD_G, KLD_NLF, KLD_NF, KLD_R]``.  ``vis_dir``: write the reference's ``.mat`` dumps under ``vis_dir/<fn>``."""
    subdir = None
    if vis_dir is not None:
        import os
        subdir = os.path.join(vis_dir, str(mb['fn']).split('|')[0])
        os.makedirs(subdir, exist_ok=True)
    step = 5
    klds_avg = np.zeros(4)
    cnt = 0
    for i in range(0, np.asarray(mb['_x']).shape[0], step):
        klds_avg += kldiv_patch_set(i, mb, x_samples, sc_sd, subdir)
        cnt += 1
    return klds_avg / cnt


def noise_bin_edges(n_bins=1000, lo=-0.25, hi=0.25):
    """Symmetric edges for NOISE values (the reference's default [0,1] range suits
    clipped images; raw noise is centred at 0)."""
    return np.linspace(lo, hi, n_bins + 1)


def nll_gauss(x, sd):
    """Per-patch NLL of noise ``x`` under i.i.d. N(0, sd²) — the NLL_G baseline."""
    x = np.asarray(x, np.float64)
    n = x[0].size
    return 0.5 * n * np.log(2 * np.pi * sd * sd) + 0.5 * (x * x).sum(axis=tuple(range(1, x.ndim))) / (sd * sd)


def nll_sdn(x, y, beta1, beta2):
    """Per-patch NLL of noise ``x`` under the camera NLF N(0, β1·y + β2) — NLL_SDN."""
    x = np.asarray(x, np.float64)
    var = beta1 * np.asarray(y, np.float64) + beta2
    return 0.5 * (np.log(2 * np.pi * var) + x * x / var).sum(axis=tuple(range(1, x.ndim)))
