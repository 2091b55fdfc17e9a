"""``NoiseFlow`` — the reference's operator surface on the MI355X HIP library.

Mirrors ``borealisflows/noise_flow_model.py::NoiseFlow`` (reference file:line in
each docstring): same constructor arguments, same method names, argument order
and conditioning convention (``iso`` / ``cam`` / ``nlf0`` / ``nlf1`` are
length-1 lists or scalars — ONE value per call, ``MiniBatchSampler.py:61-64``).
The TF graph tensors become eager arrays: numpy in → numpy out, torch (CUDA)
tensor in → torch tensor out (zero-copy).  Every method runs the fused HIP
kernels through the C ABI; there is no CPU execution path.

``x`` = real noise, ``y`` = clean image, ``z`` = latent
(``train_noise_flow.py:284-285``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np

from . import _lib, params as _params
from .ckpt import load_checkpoint, save_checkpoint

_U64 = (1 << 64) - 1


def _first(v, default=0.0) -> float:
    """Conditioning values arrive as length-1 lists (reference feed dicts) or scalars."""
    if v is None:
        return float(default)
    a = np.asarray(v if not hasattr(v, "detach") else v.detach().cpu().numpy(), dtype=np.float64).reshape(-1)
    if a.size != 1:
        raise ValueError("conditioning is per call, not per patch: expected one value, got %d "
                         "(reference feeds length-1 lists)" % a.size)
    return float(a[0])


_NARROW_POOL = None


def _to_float32(a) -> np.ndarray:
    """Contiguous float32 view/copy of a host array.  float64 minibatches (the reference's dtype, quirk Q9)
    are narrowed in parallel slices — numpy's cast releases the GIL but is single-threaded (6 GB/s), and
    torch's threaded CPU cast oversubscribes a cgroup-limited container (measured 10x slower)."""
    a = np.asarray(a)
    if a.dtype == np.float32 or a.size < (1 << 20) or a.ndim == 0:
        return np.ascontiguousarray(a, dtype=np.float32)
    global _NARROW_POOL
    if _NARROW_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        try:
            n = len(os.sched_getaffinity(0))
        except Exception:
            n = os.cpu_count() or 1
        _NARROW_POOL = ThreadPoolExecutor(max_workers=max(1, min(8, n)))
    out = np.empty(a.shape, np.float32)
    n = a.shape[0]
    parts = min(8, n)
    edges = [n * k // parts for k in range(parts + 1)]

    def cast(k):
        out[edges[k]:edges[k + 1]] = a[edges[k]:edges[k + 1]]
    list(_NARROW_POOL.map(cast, range(parts)))
    return out


def _host_out(torch, shape):
    """float32 host array for a result.  Large ones come from torch's caching PINNED host allocator: the library then copies
    D2H straight into them (no staging copy), and — unlike a fresh ``np.empty`` — a recycled block has no first-touch page
    faults (64 MiB of fresh pages cost ~6 ms, more than the whole sampling call).  The array keeps its block alive."""
    n = int(np.prod(shape))
    if n < (1 << 20):
        return np.empty(shape, np.float32)
    return torch.empty(tuple(shape), dtype=torch.float32, device="cpu", pin_memory=True).numpy()


def _host_tensor(a, shape_tail):
    """A host array as the host-fed C ABI takes it: C-contiguous float32 or float64 [B, *shape_tail] → (array, dtype code)."""
    a = np.asarray(a)
    if a.dtype != np.float32 and a.dtype != np.float64:
        a = a.astype(np.float32)
    a = np.ascontiguousarray(a)
    if shape_tail is not None and tuple(a.shape[1:]) != tuple(shape_tail):
        raise ValueError("expected tensor of shape [B,%s], got %s" % (",".join(map(str, shape_tail)), tuple(a.shape)))
    return a, (_lib.NF_HOST_F64 if a.dtype == np.float64 else _lib.NF_HOST_F32)


class _Dev:
    """Device plumbing (torch is used ONLY for HBM allocations and streams)."""

    def __init__(self, device=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("noise_flow_amd needs a ROCm GPU (MI355X): torch.cuda.is_available() is False "
                               "and there is no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))

    def to_dev(self, a, shape_tail=None):
        """→ (contiguous float32 CUDA tensor, was_numpy)."""
        torch = self.torch
        was_np = not isinstance(a, torch.Tensor)
        if was_np:
            t = torch.from_numpy(_to_float32(a)).to(self.device, non_blocking=False)
        else:
            t = a.to(device=self.device, dtype=torch.float32).contiguous()
        if shape_tail is not None and tuple(t.shape[1:]) != tuple(shape_tail):
            raise ValueError("expected tensor of shape [B,%s], got %s" % (",".join(map(str, shape_tail)), tuple(t.shape)))
        return t, was_np

    def empty(self, shape, dtype=None):
        return self.torch.empty(shape, dtype=dtype or self.torch.float32, device=self.device)

    def stream_ptr(self) -> int:
        return int(self.torch.cuda.current_stream(self.device).cuda_stream)

    def back(self, t, as_numpy: bool):
        """Results go back the way the inputs came.  numpy: through a page-locked host tensor from torch's
        caching host allocator for large tensors (64 MiB: 10 ms pageable, 1.3 ms pinned); the returned array
        owns that block until it is garbage-collected."""
        if not as_numpy:
            return t
        if t.device.type == "cpu":    # the host-fed path already delivered into host memory
            return t.numpy()
        if t.numel() < (1 << 23):     # below 32 MiB the pageable path is as fast (measured) and has no set-up cost
            return t.cpu().numpy()
        h = self.torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
        h.copy_(t)
        return h.numpy()


class FlowHandle:
    """Owns one ``nf_handle`` (a folded, device-resident model)."""

    def __init__(self, arch, variables: Dict[str, np.ndarray], x_shape, width: int,
                 binding: str = "loss_first", device: Optional[int] = None, layers=None, tmpl=None,
                 cnn_dtype: str = "fp32", flow_permutation: int = 1, decomp: str = "LU"):
        self.lib = _lib.load()
        if layers is None:
            self.layers, descs, flat = _params.pack(arch, variables, width, binding, flow_permutation, decomp)
        else:   # an explicit sub-list of bijectors (noise_flow_amd.layers)
            self.layers, descs, flat = _params.pack_layers(layers, variables, width, tmpl or {})
        H, W, Cc = (int(v) for v in x_shape)
        self.x_shape = (H, W, Cc)
        if cnn_dtype not in ("fp32", "fp16"):
            raise ValueError("cnn_dtype must be 'fp32' or 'fp16'")
        flags = _lib.NF_CFG_FP16_CNN if cnn_dtype == "fp16" else 0
        cfg = _lib.nf_config(H, W, Cc, len(self.layers), -1 if device is None else int(device), flags)
        h = C.c_void_p()
        _lib.check(self.lib.nf_create(C.byref(cfg), descs, flat.ctypes.data_as(C.POINTER(C.c_float)), flat.size,
                                      C.byref(h)))
        self._h = h
        self.has_sdn = any(L.kind.startswith("sdn") for L in self.layers)
        self.width = int(width)
        # template scope of every coupling CNN, NLL layer order (the rows of nf_*_batchstats' moments)
        tb = tmpl if layers is not None else _params.template_binding(self.layers, binding)
        self.coupling_scopes = [_params.template_scope((tb or {}).get(L.arch_index, 0))
                                for L in self.layers if L.kind == "coupling"]

    def close(self):
        if getattr(self, "_h", None):
            self.lib.nf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ptr(self):
        return self._h


class NoiseFlow(object):
    """Reference: ``borealisflows/noise_flow_model.py:44-513``.

    Parameters
    ----------
    x_shape : [H, W, C]          (noise_flow_model.py:46)
    is_training : bool           ``False`` (default): stored running BN statistics, the
                                 fused single-pass path of every NLL / sampling
                                 measurement (train_noise_flow.py:112-113,167-168).
                                 ``True``: batch-statistics BN (layers.py:386-398) —
                                 every call normalises with the moments of its own
                                 patches (2 statistics passes per coupling, then the
                                 fused pass) and moves the running statistics by the
                                 reference's EMA (decay 0.1) in :attr:`variables`.
    hps : namespace with ``arch, width, decomp, flow_permutation, squeeze_factor,
          n_levels`` (and optionally ``seed``).
    variables : optional ``{name: ndarray}`` under the reference's checkpoint
          names; default = fresh initialisation with the reference initialisers.
    cnn_dtype : 'fp32' (default) | 'fp16' — precision of the coupling-CNN convolutions.
    binding : 'loss_first' | 'sample_first' — template→layer binding (quirk Q1).
    """

    def __init__(self, x_shape, is_training=False, hps=None, variables=None, binding="loss_first", device=None,
                 cnn_dtype=None):
        if hps is None:
            raise ValueError("hps is required (arch, width, ...)")
        self.x_shape = [int(v) for v in x_shape]
        self.hps = hps
        self.depth = getattr(hps, "depth", -1)
        self.n_levels = int(getattr(hps, "n_levels", 1))
        self._is_training = is_training
        self.binding = binding
        if self.n_levels != 1:
            raise NotImplementedError("n_levels > 1 (split2d) is outside the hot path (shipped: n_levels = 1)")
        if int(getattr(hps, "squeeze_factor", 1)) != 1:
            raise NotImplementedError("squeeze_factor != 1 is outside the hot path (shipped: 1)")
        # noise_flow_model.py:80-92 / matrix_param.py:191-193: 1 = Conv2d1x1 (shipped; decomp LU | LU2 | NONE), 0 = channel-reversing
        # tfb.Permute, anything else = no mixing layer
        self.flow_permutation = int(getattr(hps, "flow_permutation", 1))
        self.decomp = str(getattr(hps, "decomp", "LU"))
        self.arch = hps.arch
        self.width = int(getattr(hps, "width", 4))
        self._dev = _Dev(device)
        self._seed = int(getattr(hps, "seed", 0) or 0)
        self._draws = 0
        self._lock = threading.Lock()
        self._variables = dict(variables) if variables is not None else _params.init_variables(
            self.arch, self.width, self.x_shape[-1], self._seed, self.flow_permutation, self.decomp,
            float(getattr(hps, "gain_init", -5.0)))
        self.model = [_params.parse_arch(self.arch, self.flow_permutation, self.decomp)]   # bijector list per level (define_flow_structure)
        # 'fp16': coupling-CNN convs in half precision on the matrix cores, everything else fp32
        # (BASELINE configs[4]); also selectable as hps.cnn_dtype.  Default: all fp32.
        self.cnn_dtype = cnn_dtype or str(getattr(hps, "cnn_dtype", "fp32"))
        self._flow = FlowHandle(self.arch, self._variables, self.x_shape, self.width, binding, self._dev.device.index,
                                cnn_dtype=self.cnn_dtype, flow_permutation=self.flow_permutation, decomp=self.decomp)

    # ------------------------------------------------------------------ variables
    @property
    def variables(self) -> Dict[str, np.ndarray]:
        return self._variables

    def num_params(self) -> int:
        return _params.count_trainable(self._variables)

    def load_variables(self, variables: Dict[str, np.ndarray], binding: Optional[str] = None) -> None:
        """Swap in a new variable set (the equivalent of ``Saver.restore``)."""
        if binding is not None:
            self.binding = binding
        self._variables = dict(variables)
        old = self._flow
        self._flow = FlowHandle(self.arch, self._variables, self.x_shape, self.width, self.binding,
                                self._dev.device.index, cnn_dtype=self.cnn_dtype, flow_permutation=self.flow_permutation,
                                decomp=self.decomp)
        old.close()
        if getattr(self, "_sync", None) is not None:
            self._install_sync()

    # ------------------------------------------------------------------ cross-rank batch statistics (is_training=True)
    def set_sync_bn(self, group=None, enabled: bool = True) -> None:
        """Batch-statistics calls (``is_training=True``) across ranks: the reference's ``batch_norm`` takes its moments over the
        whole minibatch (layers.py:386-398), which under data parallelism is the union of the ranks' shards.  Installs the
        all-reduce hook of ``nf_set_sync`` for ``group`` (``None`` / ``True`` = the default process group): every statistics
        pass (2 per coupling) all-reduces its sums, so N ranks x B patches evaluate exactly like one rank on the N*B
        concatenated patches — same moments, same running-statistics EMA on every rank.  Every rank must call with the same B."""
        import torch.distributed as dist
        grp = None if group in (None, True) else group
        self._sync = None
        if enabled and dist.is_available() and dist.is_initialized() and dist.get_world_size(grp) > 1:
            torch = self._dev.torch
            buf = torch.zeros((64,), dtype=torch.float64, device=self._dev.device)
            err = []

            def allreduce(user, ptr, count, stream):
                try:   # `ptr` is buf's storage; the library wrote this rank's sums on `stream` and reads them back there
                    with torch.cuda.stream(torch.cuda.ExternalStream(int(stream or 0), device=self._dev.device)):
                        dist.all_reduce(buf[:int(count)], op=dist.ReduceOp.SUM, group=grp)
                    return 0
                except Exception as e:   # never let an exception cross the C frame
                    err.append(e)
                    return 1
            self._sync = (_lib.ALLREDUCE_FN(allreduce), buf, dist.get_world_size(grp), err, grp)
        self._install_sync()

    def _check_equal_shards(self, B: int) -> None:
        """The library forms the synchronised moments with n = world x the LOCAL pixel count (nf_set_sync), i.e. every rank must
        feed the same number of patches to an ``is_training=True`` call; a MAX all-reduce of (B, -B) says so before the call
        (these calls synchronise anyway: nf_*_batchstats end in a stream synchronise)."""
        sync = getattr(self, "_sync", None)
        if sync is None:
            return
        import torch.distributed as dist
        torch = self._dev.torch
        t = torch.tensor([float(B), -float(B)], dtype=torch.float64, device=self._dev.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=sync[4])
        hi, neg_lo = (float(v) for v in t.cpu())
        if hi != -neg_lo:
            raise ValueError("set_sync_bn needs the same number of patches on every rank (between %d and %d here): the batch "
                             "moments are formed with world x the local pixel count" % (int(-neg_lo), int(hi)))

    def _install_sync(self):
        sync = getattr(self, "_sync", None)
        if sync is None:
            _lib.check(self._flow.lib.nf_set_sync(self._flow.ptr, _lib.ALLREDUCE_FN(0), None, None, 1))
        else:
            _lib.check(self._flow.lib.nf_set_sync(self._flow.ptr, sync[0], None, sync[1].data_ptr(), sync[2]))

    def restore(self, ckpt_prefix: str, binding: Optional[str] = None) -> None:
        """``saver.restore(sess, prefix)`` (NoiseFlowWrapper.py:77) on a TF bundle, without TF."""
        self.load_variables(load_checkpoint(ckpt_prefix), binding)

    def save(self, ckpt_prefix: str) -> None:
        save_checkpoint(ckpt_prefix, self._variables)

    def get_layer_names(self):
        """noise_flow_model.py:508-513 (matches hps.txt:1-18)."""
        return [L.name for L in self.model[0]]

    # ------------------------------------------------------------------ helpers
    def _cond(self, nlf0, nlf1, iso, cam):
        return _lib.nf_cond(_first(iso), _first(cam), _first(nlf0), _first(nlf1))

    def _check_mode(self):
        if self._is_training not in (True, False):
            raise NotImplementedError("is_training must be a Python bool (the reference's placeholder is fed per run; "
                                      "construct one NoiseFlow per mode)")

    def _moments_buffer(self):
        n = len(self._flow.coupling_scopes)
        return np.zeros((max(n, 1), 4, self._flow.width), np.float32)

    def _apply_bn_ema(self, moments: np.ndarray) -> None:
        """layers.py:392-393: ``train_m -= decay*(train_m - m)`` (same for the variance) with the
        batch moments the call just used; the handle itself never reads the running statistics
        in this mode, so only :attr:`variables` (what ``save`` writes) moves."""
        decay = np.float32(0.1)
        with self._lock:
            for row, scope in enumerate(self._flow.coupling_scopes):
                for k, name in enumerate(("bn_nvp_conv_1/mean", "bn_nvp_conv_1/var",
                                          "bn_nvp_conv_2/mean", "bn_nvp_conv_2/var")):
                    key = scope + "/" + name
                    old = np.asarray(self._variables[key], np.float32)
                    self._variables[key] = (old - decay * (old - moments[row, k].reshape(old.shape))).astype(np.float32)

    def _run_nll_host(self, x, y, cond, want_z: bool, flags: int, want_sums: bool):
        """numpy in → numpy out through ``nf_nll_host``: the float64 → float32 narrowing, H2D, the kernel and D2H of
        consecutive chunks overlap inside the library (what a ``sess.run(feed_dict=numpy)`` caller of the reference gets).
        Returns CPU torch tensors that alias the numpy results, so that the callers' arithmetic is the device path's."""
        torch = self._dev.torch
        tail = tuple(self.x_shape)
        xa, dt = _host_tensor(x, tail)
        ya = None
        if y is not None:
            ya, dty = _host_tensor(y, tail)
            if ya.shape[0] != xa.shape[0]:
                raise ValueError("x and y batch sizes differ")
            if dty != dt:                      # one dtype per call: narrow the float64 one here (rare)
                xa, ya, dt = xa.astype(np.float32, copy=False), ya.astype(np.float32, copy=False), _lib.NF_HOST_F32
        B = int(xa.shape[0])
        nll, sd, ld = (np.empty((B,), np.float32) for _ in range(3))
        z = _host_out(torch, xa.shape) if want_z else None
        sums = np.zeros(3, np.float64) if want_sums else None
        with torch.cuda.device(self._dev.device):
            _lib.check(self._flow.lib.nf_nll_host(self._flow.ptr, xa.ctypes.data, ya.ctypes.data if ya is not None else None, dt, B,
                                                  C.byref(cond), nll.ctypes.data, sd.ctypes.data, ld.ctypes.data,
                                                  z.ctypes.data if z is not None else None,
                                                  sums.ctypes.data if sums is not None else None, flags))
        t = torch.from_numpy
        return t(nll), t(sd), t(ld), (t(z) if z is not None else None), (t(sums) if sums is not None else None), True

    def _run_nll(self, x, y, cond, want_z: bool, flags: int = 0, want_sums: bool = False):
        dev = self._dev
        tail = tuple(self.x_shape)
        if not self._is_training and not isinstance(x, dev.torch.Tensor) and not isinstance(y, dev.torch.Tensor):
            return self._run_nll_host(x, y, cond, want_z, flags, want_sums)
        xt, was_np = dev.to_dev(x, tail)
        yt = None
        if y is not None:
            yt, _ = dev.to_dev(y, tail)
            if yt.shape[0] != xt.shape[0]:
                raise ValueError("x and y batch sizes differ")
        B = int(xt.shape[0])
        torch = dev.torch
        nll = dev.empty((B,))
        sd = dev.empty((B,))
        ld = dev.empty((B,))
        z = dev.empty(xt.shape) if want_z else None
        sums = None
        if want_sums:   # slotted layout: the per-workgroup atomics spread over 64 cache lines
            sums = torch.empty((_lib.NF_SUMS_SLOTS * _lib.NF_SUMS_STRIDE,), dtype=torch.float64, device=dev.device)
            flags |= _lib.NF_SUMS_WIDE
        args = (self._flow.ptr, xt.data_ptr(), yt.data_ptr() if yt is not None else None, B, C.byref(cond),
                nll.data_ptr(), sd.data_ptr(), ld.data_ptr(), z.data_ptr() if z is not None else None,
                sums.data_ptr() if sums is not None else None, flags)
        with torch.cuda.device(dev.device):
            if self._is_training:
                mom = self._moments_buffer()
                self._check_equal_shards(int(xt.shape[0]))
                _lib.check(self._flow.lib.nf_nll_batchstats(*args, mom.ctypes.data, dev.stream_ptr()))
                self._apply_bn_ema(mom)
            else:
                _lib.check(self._flow.lib.nf_nll(*args, dev.stream_ptr()))
        return nll, sd, ld, z, sums, was_np

    # ------------------------------------------------------------------ NLL direction
    def inverse(self, x, objective, yy=None, nlf0=None, nlf1=None, iso=None, cam=None):
        """noise_flow_model.py:394-428: run every bijector's
        ``_inverse_and_log_det_jacobian``; returns ``(z, objective + Σ log|det J|)``."""
        self._check_mode()
        if yy is None and self._flow.has_sdn:
            raise ValueError("this architecture has a signal-dependent layer: the clean image yy is required")
        nll, sd, ld, z, _, was_np = self._run_nll(x, yy, self._cond(nlf0, nlf1, iso, cam), True, _lib.NF_NO_PRIOR)
        if objective is None:
            obj = ld
        elif isinstance(objective, self._dev.torch.Tensor):
            obj = objective.to(ld.device, ld.dtype) + ld
        else:
            obj = ld + self._dev.torch.as_tensor(np.asarray(objective, np.float32), device=ld.device)
        return self._dev.back(z, was_np), self._dev.back(obj, was_np)

    def _loss(self, x, y, nlf0=None, nlf1=None, iso=None, cam=None, reuse=False):
        """noise_flow_model.py:458-480 → ``(nll[B], sd_z)``."""
        self._check_mode()
        cond_on = getattr(self.hps, "sidd_cond", "mix") not in (None, "uncond")
        yy = y if (cond_on or self._flow.has_sdn) else None
        nll, sd, _, _, _, was_np = self._run_nll(x, yy, self._cond(nlf0, nlf1, iso, cam), False)
        self.hps.top_shape = list(self.x_shape)
        sd_z = sd.double().mean().float()
        return self._dev.back(nll, was_np), (float(sd_z) if was_np else sd_z)

    def loss(self, x, y, nlf0=None, nlf1=None, iso=None, cam=None, reuse=False):
        """noise_flow_model.py:482-484 → ``(mean_b nll_b, sd_z)``."""
        self._check_mode()
        cond_on = getattr(self.hps, "sidd_cond", "mix") not in (None, "uncond")
        yy = y if (cond_on or self._flow.has_sdn) else None
        _, _, _, _, sums, was_np = self._run_nll(x, yy, self._cond(nlf0, nlf1, iso, cam), False, 0, True)
        sums = self.fold_sums(sums)
        mean = sums[:2] / sums[2]
        if was_np:
            m = mean.cpu().numpy()
            return np.float32(m[0]), np.float32(m[1])
        return mean[0].float(), mean[1].float()

    def nll_sums(self, x, y, nlf0=None, nlf1=None, iso=None, cam=None, sums=None):
        """Device-resident ``(Σ nll, Σ sd, count)`` accumulators for one shard of a data-parallel
        evaluation; pass ``sums`` back in to keep accumulating.  ``sums`` is the C ABI's slotted
        layout (``NF_SUMS_WIDE``: 64 slots, 128 B apart, so the per-workgroup atomics do not
        serialise on one cache line) — a float64 tensor of ``NF_SUMS_SLOTS*NF_SUMS_STRIDE``
        elements; :meth:`fold_sums` turns it into the plain ``float64[3]``, after which the
        caller finishes the mean with one RCCL all-reduce (``noise_flow_amd.dist``).  A plain
        3-element tensor is accepted too (all atomics on one line: ~10 % slower at batch 1024)."""
        self._check_mode()
        dev = self._dev
        tail = tuple(self.x_shape)
        xt, _ = dev.to_dev(x, tail)
        yt = dev.to_dev(y, tail)[0] if y is not None else None
        torch = dev.torch
        if sums is None:
            sums = self.new_sums()
        flags = _lib.NF_ACCUMULATE | (_lib.NF_SUMS_WIDE if sums.numel() != 3 else 0)
        cond = self._cond(nlf0, nlf1, iso, cam)
        args = (self._flow.ptr, xt.data_ptr(), yt.data_ptr() if yt is not None else None, int(xt.shape[0]),
                C.byref(cond), None, None, None, None, sums.data_ptr(), flags)
        with torch.cuda.device(dev.device):
            if self._is_training:
                mom = self._moments_buffer()
                self._check_equal_shards(int(xt.shape[0]))
                _lib.check(self._flow.lib.nf_nll_batchstats(*args, mom.ctypes.data, dev.stream_ptr()))
                self._apply_bn_ema(mom)
            else:
                _lib.check(self._flow.lib.nf_nll(*args, dev.stream_ptr()))
        return sums

    def new_sums(self):
        """A zeroed slotted accumulator for :meth:`nll_sums`."""
        torch = self._dev.torch
        return torch.zeros((_lib.NF_SUMS_SLOTS * _lib.NF_SUMS_STRIDE,), dtype=torch.float64, device=self._dev.device)

    def fold_sums(self, sums, out=None):
        """Slotted accumulator → ``float64[3] = (Σ nll, Σ sd, count)`` on the device (``nf_sums_reduce``);
        ``out`` (optional, 3 elements) is ADDED to.  A 3-element ``sums`` is returned unchanged."""
        if sums.numel() == 3:
            if out is None:
                return sums
            out.add_(sums)
            return out
        dev = self._dev
        torch = dev.torch
        acc = out is not None
        if out is None:
            out = torch.empty((3,), dtype=torch.float64, device=dev.device)
        with torch.cuda.device(dev.device):
            _lib.check(self._flow.lib.nf_sums_reduce(sums.data_ptr(), out.data_ptr(), _lib.NF_ACCUMULATE if acc else 0,
                                                     dev.stream_ptr()))
        return out

    # ------------------------------------------------------------------ sampling direction
    def forward(self, z, eps_std=None, yy=None, nlf0=None, nlf1=None, iso=None, cam=None):
        """noise_flow_model.py:430-447: bijectors in reverse order, ``_forward`` each.
        ``eps_std`` only matters for multi-level split priors (unused at n_levels = 1)."""
        self._check_mode()
        return self._run_sample(z, 1.0, yy, self._cond(nlf0, nlf1, iso, cam), z_is_eps=True)

    def sample(self, y, eps_std=None, yy=None, nlf0=None, nlf1=None, iso=None, cam=None, eps=None, seed=None):
        """noise_flow_model.py:449-456: ``z = ε·eps_std`` (prior.sample, :499-504), then
        :meth:`forward`.  ``y`` only supplies the batch shape in the reference.

        ``eps`` (optional, [B,H,W,C]) supplies the N(0,1) draw — the parity path,
        since TF's ``random_normal`` stream cannot be reproduced (quirk Q11).
        Without it ε is generated in-kernel (Philox4x32-10 keyed by ``seed``, a
        running patch counter and the pixel index)."""
        self._check_mode()
        cond = self._cond(nlf0, nlf1, iso, cam)
        tv = None if eps_std is None else np.asarray(eps_std.detach().cpu() if hasattr(eps_std, "detach") else eps_std,
                                                     np.float32).reshape(-1)
        if tv is not None and tv.size > 1 and not np.all(tv == tv[0]):
            # one temperature PER PATCH (the reference reshapes eps_std to [-1,1,1,1], noise_flow_model.py:501): the draw is
            # scaled patch by patch here and the kernel runs at temperature 1
            torch = self._dev.torch
            B = int(np.shape(y)[0])
            if tv.size != B:
                raise ValueError("eps_std holds %d temperatures for %d patches" % (tv.size, B))
            as_np = eps is not None and not isinstance(eps, torch.Tensor) or eps is None and not isinstance(y, torch.Tensor)
            if eps is None:
                # the draw the kernel would make for these patches (same Philox key: seed, running patch counter, pixel), so
                # that a seed gives the same noise whether the temperature is one number or one per patch
                with self._lock:
                    base = self._draws
                    self._draws += B
                e = torch.empty((B,) + tuple(self.x_shape), device=self._dev.device, dtype=torch.float32)
                with torch.cuda.device(self._dev.device):
                    _lib.check(self._flow.lib.nf_sample_eps(int(self._seed if seed is None else seed) & _U64, base, B, self.x_shape[0],
                                                            self.x_shape[1], e.data_ptr(), self._dev.stream_ptr()))
            else:
                e = self._dev.to_dev(eps, tuple(self.x_shape))[0]
            e = e * torch.as_tensor(tv, device=self._dev.device).reshape(-1, 1, 1, 1)
            out = self._run_sample(e, 1.0, yy, cond, z_is_eps=True)
            return self._dev.back(out, True) if as_np and isinstance(out, torch.Tensor) else out
        temp = 1.0 if tv is None else float(tv[0])
        if eps is not None:
            return self._run_sample(eps, temp, yy, cond, z_is_eps=True)
        return self._run_sample(y, temp, yy, cond, z_is_eps=False, seed=seed)

    def _run_sample(self, z_or_y, temp, yy, cond, z_is_eps, seed=None):
        dev = self._dev
        tail = tuple(self.x_shape)
        if yy is None and self._flow.has_sdn:
            raise ValueError("this architecture has a signal-dependent layer: the clean image yy is required")
        if not self._is_training and not isinstance(z_or_y, dev.torch.Tensor) and not isinstance(yy, dev.torch.Tensor):
            # numpy in → numpy out through nf_sample_host (chunked, full duplex: y goes down while x comes up)
            za, zdt = _host_tensor(z_or_y, tail)
            B = int(za.shape[0])
            ya, ydt = _host_tensor(yy, tail) if yy is not None else (None, _lib.NF_HOST_F32)
            if ya is not None and ya.shape[0] != B:
                raise ValueError("batch sizes differ")
            eps = np.ascontiguousarray(za, dtype=np.float32) if z_is_eps else None
            if z_is_eps:
                base, sd = 0, 0
            else:
                sd = self._seed if seed is None else int(seed)
                with self._lock:
                    base = self._draws
                    self._draws += B
            out = _host_out(dev.torch, za.shape)
            with dev.torch.cuda.device(dev.device):
                _lib.check(self._flow.lib.nf_sample_host(self._flow.ptr, ya.ctypes.data if ya is not None else None, ydt,
                                                         eps.ctypes.data if eps is not None else None, sd & _U64, base, float(temp), B,
                                                         C.byref(cond), out.ctypes.data))
            return out
        zt, was_np = dev.to_dev(z_or_y, tail)
        yt = dev.to_dev(yy, tail)[0] if yy is not None else None
        B = int(zt.shape[0])
        out = dev.empty(zt.shape)
        if z_is_eps:
            base, sd = 0, 0
        else:
            sd = self._seed if seed is None else int(seed)
            with self._lock:
                base = self._draws
                self._draws += B
        args = (self._flow.ptr, yt.data_ptr() if yt is not None else None, zt.data_ptr() if z_is_eps else None,
                sd & _U64, base, float(temp), B, C.byref(cond), out.data_ptr())
        with dev.torch.cuda.device(dev.device):
            if self._is_training:
                mom = self._moments_buffer()
                self._check_equal_shards(B)
                _lib.check(self._flow.lib.nf_sample_batchstats(*args, mom.ctypes.data, dev.stream_ptr()))
                self._apply_bn_ema(mom)
            else:
                _lib.check(self._flow.lib.nf_sample(*args, dev.stream_ptr()))
        return dev.back(out, was_np)


def default_hps(**kw) -> SimpleNamespace:
    """The hyper-parameters the hot path reads, with the shipped values
    (models/NoiseFlow/hps.txt)."""
    d = dict(arch="sdn5|unc|unc|unc|unc|gain4|unc|unc|unc|unc", width=4, decomp="LU", flow_permutation=1,
             squeeze_factor=1, squeeze_type="chessboard", n_levels=1, depth=-1, sidd_cond="mix", gain_init=-5.0,
             x_shape=[None, 32, 32, 4], seed=0, temp=1.0)
    d.update(kw)
    return SimpleNamespace(**d)
