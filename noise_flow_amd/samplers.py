"""The host pipeline in front of the hot path: image tuples → patch dicts → minibatch dicts.

Mirrors the reference's queue stages for IN-MEMORY image tuples (what ``ImageLoader`` hands on after it has
read a SIDD pair — the h5py / .mat reading itself needs files this project does not have and stays out):

* ``sample_indices_random``   — ``sidd/sidd_utils.py:849-858``: ``n_p`` origins drawn with ``np.random.randint``
  (row first, then column, per patch — the draw order is part of the contract: a seeded run of the reference gives
  the same origins);
* ``PatchSampler``            — ``sidd/PatchSampler.py:20-79``: same constructor, same patch dict
  (``in / gt / vr / nlf0 / nlf1 / iso / cam / fn / metadata / pid``), 'uniform' = every grid patch (optionally shuffled
  like ``sklearn.utils.shuffle`` on the global numpy RNG), anything else = ``n_pat_per_im`` random origins;
* ``MiniBatchSampler``        — ``sidd/MiniBatchSampler.py:19-78``: same constructor, same minibatch dict (float64 ``_x``
  / ``_y`` / ``pid``; the conditioning of the LAST patch as length-1 lists — "only one value for the whole mini-batch").

Differences, all on the outside: the worker threads are daemons and ``close()`` stops them (the reference's never
end); a patch count that differs from ``n_pat_per_im`` raises instead of dropping into ``pdb``.
"""
from __future__ import annotations

import queue
from threading import Event, Thread
from typing import List, Tuple

import numpy as np

from .patches import patch_origins


def sample_indices_uniform(h, w, ph, pw, shuf=False, n_pat_per_im=None):
    """``sidd_utils.py:830-846``: the row-major grid of non-overlapping patches (truncated at ``n_pat_per_im``); ``shuf``
    permutes it the way ``sklearn.utils.shuffle(ii, jj)`` does — one in-place shuffle of ``arange(n)`` on the GLOBAL numpy
    RNG, applied to both lists."""
    ii, jj, n_p = patch_origins(h, w, ph, pw, n_pat_per_im)
    if shuf:
        perm = np.arange(n_p)
        np.random.shuffle(perm)
        ii, jj = [ii[k] for k in perm], [jj[k] for k in perm]
    return ii, jj, n_p


def sample_indices_random(h, w, ph, pw, n_p) -> Tuple[List[int], List[int]]:
    """``n_p`` random patch origins in an ``h`` x ``w`` image (global numpy RNG, row then column per patch)."""
    ii, jj = [], []
    for _ in range(int(n_p)):
        ii.append(np.random.randint(0, h - ph + 1))
        jj.append(np.random.randint(0, w - pw + 1))
    return ii, jj


class _Stage:
    """A queue stage with daemon workers that can be stopped."""

    def __init__(self, max_queue_size, n_threads, target):
        self.total_wait_time_get = 0
        self.total_wait_time_put = 0
        self.max_queue_size = max_queue_size
        self.queue = queue.Queue(maxsize=self.max_queue_size)
        self._stop = Event()
        self.n_threads = n_threads
        self.threads = []
        for t in range(self.n_threads):
            th = Thread(target=target, args=(t,), daemon=True)
            self.threads.append(th)
            th.start()

    def _get(self, q):
        while not self._stop.is_set():
            try:
                return q.get(timeout=0.05)
            except queue.Empty:
                continue
        raise _Stopped()

    def _put(self, item):
        while not self._stop.is_set():
            try:
                self.queue.put(item, timeout=0.05)
                return
            except queue.Full:
                continue
        raise _Stopped()

    def get_queue(self):
        return self.queue

    def get_total_wait_time(self):
        return self.total_wait_time_get, self.total_wait_time_put

    def close(self):
        self._stop.set()
        for th in self.threads:
            th.join(timeout=2.0)


class _Stopped(Exception):
    pass


class PatchSampler(_Stage):
    def __init__(self, im_tuple_queue, patch_height=256, sampling='uniform', max_queue_size=256, n_threads=4,
                 n_reuse_image=0, n_pat_per_im=1, shuffle=True):
        self.im_tuple_queue = im_tuple_queue
        self.patch_height = patch_height
        self.sampling = sampling
        self.n_reuse_image = n_reuse_image
        self.n_pat_per_im = n_pat_per_im
        self.shuffle = shuffle
        super().__init__(max_queue_size, n_threads, self.sample_patches_thread)

    def patches_of(self, im_tuple) -> List[dict]:
        """The patch dicts of one image tuple (``in`` / ``gt``: [1, H, W, C]), in queue order."""
        H, W = im_tuple['in'].shape[1], im_tuple['in'].shape[2]
        ph = self.patch_height
        if self.sampling == 'uniform':   # use all patches in image
            ii, jj, n_p = sample_indices_uniform(H, W, ph, ph, shuf=self.shuffle, n_pat_per_im=self.n_pat_per_im)
            if n_p != self.n_pat_per_im:
                raise ValueError('# patches/image = %d != %d (fn = %s)' % (n_p, self.n_pat_per_im, str(im_tuple['fn'])))
        else:                            # use self.n_pat_per_im patches
            ii, jj = sample_indices_random(H, W, ph, ph, self.n_pat_per_im)
        out = []
        for pid, (i, j) in enumerate(zip(ii, jj)):
            out.append({'in': im_tuple['in'][:, i:i + ph, j:j + ph, :], 'gt': im_tuple['gt'][:, i:i + ph, j:j + ph, :], 'vr': [],
                        'nlf0': im_tuple['nlf0'], 'nlf1': im_tuple['nlf1'], 'iso': im_tuple['iso'], 'cam': im_tuple['cam'],
                        'fn': im_tuple['fn'], 'metadata': im_tuple['metadata'], 'pid': pid})
        return out

    def sample_patches_thread(self, thread_id, n_reuse_image=0):
        try:
            while True:
                for pat_dict in self.patches_of(self._get(self.im_tuple_queue)):
                    self._put(pat_dict)
        except _Stopped:
            return


class MiniBatchSampler(_Stage):
    def __init__(self, patch_tuple_queue, minibatch_size=24, max_queue_size=16, n_threads=4, pat_stats=None):
        self.patch_tuple_queue = patch_tuple_queue
        self.mini_batch_size = minibatch_size
        self.pat_stats = pat_stats
        super().__init__(max_queue_size, n_threads, self.sample_minibatch_thread)

    @staticmethod
    def collate(pat_dicts) -> dict:
        """One minibatch dict from ``mini_batch_size`` patch dicts (arrays float64, like ``np.zeros`` in the reference)."""
        n = len(pat_dicts)
        p_shape = pat_dicts[0]['in'].shape
        x = np.zeros((n, p_shape[1], p_shape[2], p_shape[3]))
        y = np.zeros((n, p_shape[1], p_shape[2], p_shape[3]))
        pid = np.zeros(n)
        for p, d in enumerate(pat_dicts):
            x[p, :, :, :] = d['in']
            y[p, :, :, :] = d['gt']
            pid[p] = d['pid']   # patch index in image
        last = pat_dicts[-1]    # only one value for the whole mini-batch
        return {'_x': x, '_y': y, 'pid': pid, 'nlf0': [last['nlf0']], 'nlf1': [last['nlf1']], 'iso': [last['iso']],
                'cam': [last['cam']], 'fn': last['fn'], 'metadata': last['metadata']}

    def sample_minibatch_thread(self, thread_id, pat_stats=None):
        try:
            while True:
                self._put(self.collate([self._get(self.patch_tuple_queue) for _ in range(self.mini_batch_size)]))
        except _Stopped:
            return


class ImageTupleFeeder(_Stage):
    """In-memory stand-in for the reference's ``ImageLoader`` (``sidd/ImageLoader.py:40-72`` reads SIDD files with h5py —
    out of scope): cycles over image tuples that are already in memory and puts them on a queue, forever (``requeue=True``,
    the reference's ``n_reuse_image`` / epoch behaviour) or once."""

    def __init__(self, im_tuples, max_queue_size=16, requeue=True):
        self.im_tuples = list(im_tuples)
        self.requeue = requeue
        super().__init__(max_queue_size, 1, self._feed)

    def _feed(self, thread_id):
        try:
            while True:
                for im in self.im_tuples:
                    self._put(im)
                if not self.requeue:
                    return
        except _Stopped:
            return


class QueueEpoch:
    """``n_its`` minibatch dicts per epoch from a minibatch queue — what ``train_multithread`` / ``test_multithread`` pull
    (``train_noise_flow.py:27-117``: ``divide_parts(n_its, nthr)`` blocking ``get`` calls per epoch)."""

    def __init__(self, mb_queue, n_its, timeout=120.0):
        self.queue, self.n_its, self.timeout = mb_queue, int(n_its), timeout

    def __len__(self):
        return self.n_its

    def __iter__(self):
        for _ in range(self.n_its):
            yield self.queue.get(timeout=self.timeout)
