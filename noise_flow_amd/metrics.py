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
