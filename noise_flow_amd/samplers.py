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


# ---------------------------------------------------------------------------------------------------------
# the stage in front of PatchSampler: filename tuples -> image tuples (sidd/ImageLoader.py, sidd_utils.py:224-283)
# ---------------------------------------------------------------------------------------------------------
SIDD_CAMERAS = ['IP', 'GP', 'S6', 'N6', 'G4']   # sidd_utils.py:262


def _read_raw_mat(path):
    """The first (only) dataset of a MATLAB v7.3 file — what ``load_one_tuple_images`` reads with h5py.  h5py is not part of
    this project's environment: without it the caller must hand a reader to :func:`load_one_tuple_images`."""
    try:
        import h5py
    except ImportError as e:   # pragma: no cover
        raise ImportError("reading SIDD .MAT files needs h5py (pass read_raw= to load_one_tuple_images otherwise)") from e
    with h5py.File(path, 'r') as f:
        return np.asarray(f[list(f.keys())[0]])


def _read_metadata_mat(path):
    """``load_metadata`` (sidd_utils.py:718-723)."""
    from scipy.io import loadmat
    return loadmat(path)['metadata'][0, 0]


def get_nlf(metadata):
    """``get_nlf`` (sidd_utils.py:726-729): the camera's two NLF parameters out of the DNG tags."""
    return metadata['UnknownTags'][7, 0][2][0][0:2]


def load_one_tuple_images(filepath_tuple, read_raw=None, read_metadata=None):
    """``load_one_tuple_images`` (sidd_utils.py:224-283): (noisy path, clean path, variance path, metadata path) →
    ``(noise, gt, var, nlf0, nlf1, iso, cam, metadata)``.  Both images are Bayer-packed to [1, h/2, w/2, 4], NaNs zeroed,
    clipped to [0, 1]; the first output is the NOISE LAYER (noisy − clean, "crucial step"); non-positive NLF parameters are
    floored at 1e-6; ISO and camera come from the scene directory name (``…/0001_001_S6_00100_00060_3200_L/…`` → 100.0,
    ``SIDD_CAMERAS.index('S6')``).  ``read_raw(path)`` / ``read_metadata(path)`` default to the h5py / scipy readers."""
    from .patches import pack_raw
    in_path, gt_path, _var_path, meta_path = filepath_tuple[0], filepath_tuple[1], filepath_tuple[2], filepath_tuple[3]
    read_raw = read_raw or _read_raw_mat
    read_metadata = read_metadata or _read_metadata_mat

    def packed(path):
        im = np.expand_dims(pack_raw(np.asarray(read_raw(path))), axis=0)
        return np.clip(np.nan_to_num(im), 0.0, 1.0)
    input_image, gt_image = packed(in_path), packed(gt_path)
    metadata = read_metadata(meta_path)
    nlf0, nlf1 = get_nlf(metadata)
    fparts = in_path.split('/')
    sdir = fparts[-3]
    if len(sdir) != 30:
        sdir = fparts[-2]   # if subdirectory does not exist
    iso = float(sdir[12:17])
    cam = float(SIDD_CAMERAS.index(sdir[9:11]))
    input_image = input_image - gt_image   # the noise layer instead of the noisy image
    nlf0 = 1e-6 if nlf0 <= 0 else nlf0
    nlf1 = 1e-6 if nlf1 <= 0 else nlf1
    return input_image, gt_image, [], nlf0, nlf1, iso, cam, metadata


class ImageLoader(_Stage):
    """``sidd/ImageLoader.py:16-80``: filename tuples → image dicts ``{in, gt, vr, nlf0, nlf1, iso, cam, fn, metadata}`` with
    ``fn = <scene dir>|<file name>``; with ``requeue`` the filename tuples go round again for further epochs (the reference
    swaps two queues when the first runs empty and waits for the image queue to drain; here every consumed tuple is put back
    behind the others, which yields the same epoch order with one worker).  ``loader`` = :func:`load_one_tuple_images` or any
    callable with its result."""

    def __init__(self, filename_tuple_queue, max_queue_size=4, n_threads=4, requeue=True, loader=None):
        self.filename_tuple_queue = filename_tuple_queue
        self.requeue = requeue
        self.loader = loader or load_one_tuple_images
        super().__init__(max_queue_size, n_threads, self.load_image_tuple_thread)

    def load_image_tuple_thread(self, thread_id):
        try:
            while True:
                filename_tuple = self._get(self.filename_tuple_queue)
                parts = str.split(filename_tuple[0], '/')
                fn = parts[-3] + '|' + parts[-1]
                noise, gt, var, nlf0, nlf1, iso, cam, metadata = self.loader(filename_tuple)
                self._put({'in': noise, 'gt': gt, 'vr': var, 'nlf0': nlf0, 'nlf1': nlf1, 'iso': iso, 'cam': cam, 'fn': fn,
                           'metadata': metadata})
                if self.requeue:
                    self.filename_tuple_queue.put(filename_tuple)
                elif self.filename_tuple_queue.empty():
                    return
        except _Stopped:
            return
