"""Training step of the Noise Flow stack — host mirror of the reference's
``sess.run([train_op, loss, sd_z], {..., is_training: True})`` (``train_noise_flow.py:50-77``)
with ``train_op = get_optimizer(hps, lr, loss)`` (``train_noise_flow.py:187-198``).

``Trainer`` owns one ``nf_trainer`` (C ABI, ``include/noiseflow_hip.h``): the raw parameters, the
optimizer slots and the activation workspace live on the GPU; ``step`` only enqueues kernels on
torch's current stream.  Data-parallel training = ``forward_backward`` → one RCCL all-reduce of
the 2 433-float gradient → ``apply`` (``step(..., group=...)`` does exactly that).  With
``sync_bn=True`` the batch-normalisation sums are all-reduced as well (2·width doubles at each of
the 4 points per coupling where the reference's ``batch_norm`` / its gradient reduce over the
minibatch, ``layers.py:386-398``), so that N ranks on N shards take the SAME step as one rank on
the concatenated minibatch; without it the moments stay per rank (replicas with local statistics).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np

from . import _lib
from . import params as _params
from .noise_flow_model import _Dev, _first


class Trainer:
    """Parameters
    ----------
    x_shape : [H, W, C]
    hps : namespace with ``arch``, ``width`` and optionally ``optim`` ('adam' | 'sgd',
          train_noise_flow.py:190-196) and ``seed``
    variables : ``{name: ndarray}`` under the reference's checkpoint names; default = fresh
          initialisation with the reference's initialisers
    max_batch : largest minibatch a step will see (sizes the activation workspace);
          default ``hps.n_batch_train`` or 138 (job_noise_flow.sh)
    """

    def __init__(self, x_shape, hps, variables: Optional[Dict[str, np.ndarray]] = None, binding: str = "loss_first",
                 device=None, max_batch: Optional[int] = None, optim: Optional[str] = None):
        self.lib = _lib.load()
        self.x_shape = [int(v) for v in x_shape]
        self.hps = hps
        self.arch = hps.arch
        self.width = int(getattr(hps, "width", 4))
        self.binding = binding
        self._dev = _Dev(device)
        self.flow_permutation = int(getattr(hps, "flow_permutation", 1))
        self.decomp = str(getattr(hps, "decomp", "LU"))
        seed = int(getattr(hps, "seed", 0) or 0)
        self._variables = dict(variables) if variables is not None else _params.init_variables(
            self.arch, self.width, self.x_shape[-1], seed, self.flow_permutation, self.decomp,
            float(getattr(hps, "gain_init", -5.0)))
        self.layers = _params.parse_arch(self.arch, self.flow_permutation, self.decomp)
        self._tmpl = _params.template_binding(self.layers, binding)
        self.layers, descs, flat = _params.pack_layers(self.layers, self._variables, self.width, self._tmpl)
        self.n_params = int(flat.size)
        optim = optim or str(getattr(hps, "optim", "adam"))
        if optim not in ("adam", "sgd"):
            raise ValueError("optim must be 'adam' or 'sgd' (train_noise_flow.py:190-196)")
        self.optim = optim
        self.max_batch = int(max_batch or getattr(hps, "n_batch_train", 138) or 138)
        H, W, Cc = self.x_shape
        cfg = _lib.nf_config(H, W, Cc, len(self.layers), self._dev.device.index, 0)
        h = C.c_void_p()
        _lib.check(self.lib.nf_trainer_create(C.byref(cfg), descs, flat.ctypes.data_as(C.POINTER(C.c_float)), flat.size,
                                              self.max_batch, _lib.NF_OPT_ADAM if optim == "adam" else _lib.NF_OPT_MOMENTUM,
                                              C.byref(h)))
        self._h = h
        torch = self._dev.torch
        self._grads = torch.zeros((self.n_params,), dtype=torch.float32, device=self._dev.device)
        self._loss = torch.zeros((2,), dtype=torch.float32, device=self._dev.device)
        self.has_sdn = any(L.kind.startswith("sdn") for L in self.layers)
        # Variables shared by several layers (the reference's AUTO_REUSE scope 'sdn_gain': arch "gain4|...|gain4" has ONE
        # gain_val, "sdn5|...|sdn5" one beta1 / beta2 / gain_params / cam_params) hold one slot PER LAYER in the raw layout.
        # Their gradient is the sum over the slots; every slot gets that sum, so the copies stay identical under the update.
        slots: Dict[str, list] = {}
        pos = 0
        for L in self.layers:
            for nm in _params.layer_variable_names(L, self._tmpl):
                n = 1 if nm is None else int(np.asarray(self._variables[nm]).size)
                if nm is not None:
                    slots.setdefault(nm, []).append((pos, n))
                pos += n
        self._tied = [torch.tensor([list(range(p0, p0 + n)) for p0, n in lst], dtype=torch.long, device=self._dev.device)
                      for lst in slots.values() if len(lst) > 1]
        self._sync = None          # (group key, callback object, buffer): keeps the ctypes thunk alive

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None):
            self.lib.nf_trainer_destroy(self._h)
            self._h = None
        self._sync = None

    # ------------------------------------------------------------------ cross-rank batch normalisation
    def set_sync_bn(self, group=None, enabled: bool = True) -> None:
        """Install (or remove) the all-reduce hook of ``nf_trainer_set_sync`` for ``group`` (``None`` / ``True`` = the
        default process group).  Every rank must then call :meth:`step` with the same batch size."""
        import torch.distributed as dist
        grp = None if group in (None, True) else group
        if not enabled or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(grp) < 2:
            if self._sync is not None:
                _lib.check(self.lib.nf_trainer_set_sync(self._h, _lib.ALLREDUCE_FN(0), None, None, 1))
                self._sync = None
            return
        key = ("default" if grp is None else id(grp))
        if self._sync is not None and self._sync[0] == key:
            return
        torch = self._dev.torch
        buf = torch.zeros((64,), dtype=torch.float64, device=self._dev.device)
        err = []

        def allreduce(user, ptr, count, stream):
            try:   # `ptr` is buf's storage; the library wrote this rank's sums on torch's current stream
                dist.all_reduce(buf[:int(count)], op=dist.ReduceOp.SUM, group=grp)
                return 0
            except Exception as e:   # never let an exception cross the C frame
                err.append(e)
                return 1
        cb = _lib.ALLREDUCE_FN(allreduce)
        _lib.check(self.lib.nf_trainer_set_sync(self._h, cb, None, buf.data_ptr(), dist.get_world_size(grp)))
        self._sync = (key, cb, buf, err)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ one step
    def _inputs(self, x, y):
        tail = tuple(self.x_shape)
        xt, _ = self._dev.to_dev(x, tail)
        yt = self._dev.to_dev(y, tail)[0] if y is not None else None
        if yt is not None and yt.shape[0] != xt.shape[0]:
            raise ValueError("x and y batch sizes differ")
        if yt is None and self.has_sdn:
            raise ValueError("this architecture has a signal-dependent layer: the clean image y is required")
        return xt, yt

    def forward_backward(self, x, y, nlf0=None, nlf1=None, iso=None, cam=None):
        """→ (grads, loss2): device tensors — the raw-layout gradient of ``mean_b nll_b`` (zeros at
        non-trainable positions) and ``(loss, sd_z)``.  Moves the BN running statistics."""
        xt, yt = self._inputs(x, y)
        cond = _lib.nf_cond(_first(iso), _first(cam), _first(nlf0), _first(nlf1))
        dev = self._dev
        with dev.torch.cuda.device(dev.device):
            _lib.check(self.lib.nf_trainer_forward_backward(
                self._h, xt.data_ptr(), yt.data_ptr() if yt is not None else None, int(xt.shape[0]), C.byref(cond),
                self._grads.data_ptr(), self._loss.data_ptr(), dev.stream_ptr()))
            for idx in self._tied:
                self._grads[idx] = self._grads[idx].sum(dim=0, keepdim=True).expand(idx.shape[0], -1)
        return self._grads, self._loss

    def forward(self, x, y, nlf0=None, nlf1=None, iso=None, cam=None, sync: bool = True):
        """``sess.run([loss, sd_z], {..., is_training: True})`` without ``train_op`` — the
        ``sidd_cond == 'condSDN'`` branch of ``train_thread`` (train_noise_flow.py:61-63): batch-
        statistics forward, running statistics moved, parameters untouched."""
        xt, yt = self._inputs(x, y)
        cond = _lib.nf_cond(_first(iso), _first(cam), _first(nlf0), _first(nlf1))
        dev = self._dev
        with dev.torch.cuda.device(dev.device):
            _lib.check(self.lib.nf_trainer_forward(
                self._h, xt.data_ptr(), yt.data_ptr() if yt is not None else None, int(xt.shape[0]), C.byref(cond),
                self._loss.data_ptr(), dev.stream_ptr()))
        if not sync:
            return self._loss
        v = self._loss.cpu().numpy()
        return np.float32(v[0]), np.float32(v[1])

    def apply(self, lr: float, grads=None):
        g = self._grads if grads is None else grads
        dev = self._dev
        with dev.torch.cuda.device(dev.device):
            _lib.check(self.lib.nf_trainer_apply(self._h, g.data_ptr(), float(lr), dev.stream_ptr()))

    def step(self, x, y, nlf0=None, nlf1=None, iso=None, cam=None, lr: float = 1e-4, group=None, sync: bool = True,
             sync_bn: bool = False):
        """One ``sess.run([train_op, loss, sd_z])`` → ``(train_loss, sd_z)``.

        ``group``: a ``torch.distributed`` process group (or ``True`` for the default group) —
        averages the gradient over ranks with one all-reduce before the update.
        ``sync_bn``: also all-reduce the batch-normalisation sums over ``group`` (see the module docstring): the
        step then equals the single-process step on the concatenated minibatch; the returned loss stays the
        rank's own mean.  ``sync=False`` returns the device tensor ``[loss, sd_z]`` without waiting."""
        if group is not None or self._sync is not None:
            self.set_sync_bn(group, enabled=bool(sync_bn) and group is not None)
        if self._sync is not None:
            self._check_equal_shards(int(np.shape(x)[0]), None if group is True else group)
        try:
            grads, loss = self.forward_backward(x, y, nlf0, nlf1, iso, cam)
        except _lib.NoiseFlowLibError:
            if self._sync is not None and self._sync[3]:
                raise self._sync[3].pop()
            raise
        if group is not None:
            import torch.distributed as dist
            grp = None if group is True else group
            dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=grp)
            grads.div_(dist.get_world_size(grp))
        self.apply(lr, grads)
        if not sync:
            return loss
        v = loss.cpu().numpy()
        self._drain_shard_checks(wait=True)
        return np.float32(v[0]), np.float32(v[1])

    def _drain_shard_checks(self, wait: bool):
        pend = getattr(self, "_shard_checks", None)
        if pend is None:
            pend = self._shard_checks = []
        while pend and (wait or pend[0][0].query()):
            ev, host, step_no, src = pend.pop(0)
            ev.synchronize()
            hi, neg_lo = float(host[0]), float(host[1])
            self._shard_pool.append((src, host, ev))
            if hi != -neg_lo:
                pend.clear()
                self._shard_verified_B = None
                raise ValueError("sync_bn needs the same number of patches on every rank (step %d: between %d and %d): the batch "
                                 "moments are formed with world x the local pixel count" % (step_no, int(-neg_lo), int(hi)))

    def _check_equal_shards(self, B: int, grp):
        """Synchronised batch normalisation takes its moments over world x the LOCAL pixel count (the library's ``n``), which
        is the global count only when every rank feeds the same number of patches.  One 16-byte MAX all-reduce of (B, -B).
        The FIRST step, and every step whose B differs from the last verified one, waits for the answer before any kernel of
        the step runs — no update is ever taken with wrong moments; a step whose B equals the last verified one (every step of
        a normal run) only enqueues the check and looks at it when its copy has landed, without stalling (another rank may
        have changed ITS B).  Pinned buffers and events are allocated once per trainer and recycled."""
        import torch.distributed as dist
        torch = self._dev.torch
        self._drain_shard_checks(wait=False)
        pend = self._shard_checks
        dev = self._dev.device
        pool = getattr(self, "_shard_pool", None)
        if pool is None:
            pool = self._shard_pool = []
        if pool:
            src, host, ev = pool.pop()
        else:
            src = torch.empty(2, dtype=torch.float64, pin_memory=True)
            host = torch.empty(2, dtype=torch.float64, pin_memory=True)
            ev = torch.cuda.Event()
        src[0], src[1] = float(B), -float(B)
        t = src.to(dev, non_blocking=True)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=grp)
        host.copy_(t, non_blocking=True)
        ev.record(torch.cuda.current_stream(dev))
        pend.append((ev, host, self.steps, src))
        if getattr(self, "_shard_verified_B", None) != B:
            self._drain_shard_checks(wait=True)          # raises on a mismatch, before forward_backward
            self._shard_verified_B = B

    # ------------------------------------------------------------------ parameters
    @property
    def steps(self) -> int:
        return int(self.lib.nf_trainer_steps(self._h))

    def raw_params(self) -> np.ndarray:
        out = np.empty((self.n_params,), np.float32)
        dev = self._dev
        with dev.torch.cuda.device(dev.device):
            _lib.check(self.lib.nf_trainer_get_params(self._h, out.ctypes.data, out.size, dev.stream_ptr()))
        return out

    @property
    def variables(self) -> Dict[str, np.ndarray]:
        """The current variables under the reference's checkpoint names (trained values and the
        EMA-updated BN statistics); synchronises."""
        self._drain_shard_checks(wait=True)
        self._variables = _params.unpack_layers(self.layers, self.raw_params(), self._variables, self._tmpl)
        return self._variables

    def raw_to_variables(self, flat) -> Dict[str, np.ndarray]:
        """Name view of any raw-layout vector (e.g. the gradient)."""
        return _params.unpack_layers(self.layers, np.asarray(flat, np.float32), self._variables, self._tmpl)

    def save(self, ckpt_prefix: str) -> None:
        from .ckpt import save_checkpoint
        save_checkpoint(ckpt_prefix, self.variables)
