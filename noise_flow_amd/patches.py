"""Patch indexing and minibatch contract either side of the hot path.

Restates (does not copy) the host-side data contract of the reference:

* patch origins: ``sample_indices_uniform`` (``sidd/sidd_utils.py:830-846``) —
  row-major grid of non-overlapping ph x pw tiles, truncated at
  ``n_pat_per_im``;
* Bayer packing order: ``pack_raw`` / ``unpack_raw``
  (``sidd/sidd_utils.py:732-764``) — channels (0,0), (0,1), (1,1), (1,0);
* minibatch dict: ``MiniBatchSampler.sample_minibatch_thread``
  (``sidd/MiniBatchSampler.py:42-70``) — keys ``_x`` (noise = noisy − clean),
  ``_y`` (clean), ``pid``, and length-1 lists ``nlf0/nlf1/iso/cam``.

plus the synthetic, sharding-invariant patch source used by the benchmark
(SURVEY.md §8d): patch ``k`` is a pure function of ``(seed, k)``, generated on
the GPU by ``nf_synth_patches``.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from . import _lib

S6_ISO100_NLF = (0.000479, 0.000002)   # cam_iso_nlf.txt:8, train_noise_flow.py:146
CAM_NAMES = ("IP", "GP", "S6", "N6", "G4")   # sidd_utils.py:262


def patch_origins(h: int, w: int, ph: int, pw: int, n_pat_per_im: Optional[int] = None,
                  shuffle_seed: Optional[int] = None) -> Tuple[List[int], List[int], int]:
    """(ii, jj, n) — top-left corners, row-major; identical to the reference's
    ``sample_indices_uniform`` (without its sklearn shuffle unless a seed is given)."""
    rows = range(0, h - ph + 1, ph)
    cols = range(0, w - pw + 1, pw)
    ii = [i for i in rows for _ in cols]
    jj = [j for _ in rows for j in cols]
    if n_pat_per_im is not None:
        ii, jj = ii[:n_pat_per_im], jj[:n_pat_per_im]
    if shuffle_seed is not None:
        perm = np.random.RandomState(shuffle_seed).permutation(len(ii))
        ii, jj = [ii[p] for p in perm], [jj[p] for p in perm]
    return ii, jj, len(ii)


def patch_index_to_origin(k: int, h: int, w: int, ph: int, pw: int) -> Tuple[int, int]:
    """Closed form of the k-th origin of :func:`patch_origins` — what lets any rank
    address patch ``k`` without enumerating the others."""
    n_cols = (w - pw) // pw + 1
    n_rows = (h - ph) // ph + 1
    if not 0 <= k < n_rows * n_cols:
        raise IndexError("patch index %d out of range (%d patches)" % (k, n_rows * n_cols))
    return (k // n_cols) * ph, (k % n_cols) * pw


def pack_raw(raw: np.ndarray) -> np.ndarray:
    """Bayer (h, w) → (h/2, w/2, 4) in the reference's channel order."""
    return np.stack([raw[0::2, 0::2], raw[0::2, 1::2], raw[1::2, 1::2], raw[1::2, 0::2]], axis=2)


def unpack_raw(raw4: np.ndarray) -> np.ndarray:
    h, w = raw4.shape[:2]
    out = np.zeros((2 * h, 2 * w), dtype=np.float32)
    out[0::2, 0::2] = raw4[:, :, 0]
    out[0::2, 1::2] = raw4[:, :, 1]
    out[1::2, 1::2] = raw4[:, :, 2]
    out[1::2, 0::2] = raw4[:, :, 3]
    return out


def extract_patches(img: np.ndarray, ph: int, pw: int, n_pat_per_im: Optional[int] = None) -> np.ndarray:
    """[h, w, C] → [n, ph, pw, C] in :func:`patch_origins` order."""
    ii, jj, n = patch_origins(img.shape[0], img.shape[1], ph, pw, n_pat_per_im)
    return np.stack([img[i:i + ph, j:j + pw] for i, j in zip(ii, jj)], axis=0) if n else \
        np.zeros((0, ph, pw) + img.shape[2:], img.dtype)


def make_minibatch(noisy: np.ndarray, clean: np.ndarray, pid, nlf0: float, nlf1: float, iso: float, cam: float,
                   fn: str = "", metadata=None) -> dict:
    """The dict the reference's queues carry (MiniBatchSampler.py:66-69); arrays are
    float64 like the reference's (quirk Q9) and cast to fp32 at the device boundary."""
    return {"_x": np.asarray(noisy, np.float64) - np.asarray(clean, np.float64), "_y": np.asarray(clean, np.float64),
            "pid": np.asarray(pid, np.float64), "nlf0": [nlf0], "nlf1": [nlf1], "iso": [iso], "cam": [cam],
            "fn": fn, "metadata": metadata}


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition of the patch index range [0, n_total)."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def synth_patches(seed: int, first_patch: int, count: int, height: int = 32, width: int = 32,
                  nlf=S6_ISO100_NLF, device=None, want_x: bool = True, out=None):
    """Device-resident synthetic patches ``k = first_patch … first_patch+count-1``:
    ``y_k ~ U[0,1)``, ``x_k = ε·sqrt(β1·y_k + β2)`` → (x, y) float32 CUDA tensors
    [count, H, W, 4].  Identical for any sharding of the index range.  ``out=(x, y)`` writes into
    caller-owned contiguous tensors of that shape (e.g. slices of a resident shard) instead."""
    import torch
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
    if out is not None:
        x, y = out
        for t in (x, y):
            if t is not None and (tuple(t.shape) != (count, height, width, 4) or t.dtype != torch.float32
                                  or not t.is_contiguous() or t.device != dev):
                raise ValueError("out tensors must be contiguous float32 [count,H,W,4] on %s" % dev)
        if y is None:
            raise ValueError("out[1] (y) is required")
    else:
        y = torch.empty((count, height, width, 4), dtype=torch.float32, device=dev)
        x = torch.empty_like(y) if want_x else None
    with torch.cuda.device(dev):
        _lib.check(lib.nf_synth_patches(int(seed) & ((1 << 64) - 1), int(first_patch), int(count), height, width,
                                        float(nlf[0]), float(nlf[1]), y.data_ptr(),
                                        x.data_ptr() if x is not None else None,
                                        int(torch.cuda.current_stream(dev).cuda_stream)))
    return x, y
