"""Per-bijector operator surface: ``Conv2d1x1``, ``AffineCoupling``,
``AffineCouplingSdnEx5``, ``AffineCouplingGainEx4``.

Mirrors the six ``tfb.Bijector`` methods of the reference classes
(``borealisflows/layers.py:74-145, 251-375``,
``noise_flow_layers/AffineCouplingSdnEx5.py:22-132``,
``noise_flow_layers/AffineCouplingGainEx4.py:23-127``): ``_forward``,
``_inverse``, ``_forward_log_det_jacobian``, ``_inverse_log_det_jacobian``,
``_forward_and_log_det_jacobian``, ``_inverse_and_log_det_jacobian``.  The
conditional layers take the extra ``(yy, nlf0, nlf1, iso, cam)`` arguments
(dispatch list at ``noise_flow_model.py:403-409``).

Each bijector is a ONE-op program of the same fused HIP kernel that runs the
whole stack, so these are the product path too (no separate implementation).
Direction naming follows the reference: ``_inverse`` = likelihood direction,
``_forward`` = sampling direction.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import numpy as np

from . import _lib, params as _params
from .noise_flow_model import FlowHandle, _Dev, _first


class _Bijector:
    conditional = False

    def __init__(self, spec, variables, x_shape, width, tmpl, device=None):
        self.name = spec.name
        self.x_shape = tuple(int(v) for v in x_shape)
        self.i0, self.i1, self.ic = self.x_shape
        self.id = spec.arch_index
        self._dev = _Dev(device)
        self._flow = FlowHandle(None, variables, self.x_shape, width, device=self._dev.device.index,
                                layers=[spec], tmpl=tmpl)

    # -- raw launches ---------------------------------------------------------
    def _cond(self, nlf0=None, nlf1=None, iso=None, cam=None):
        return _lib.nf_cond(_first(iso, 100.0), _first(cam, 0.0), _first(nlf0), _first(nlf1))

    def _nll_dir(self, z, yy, cond, want_z=True):
        dev = self._dev
        zt, was_np = dev.to_dev(z, self.x_shape)
        yt = dev.to_dev(yy, self.x_shape)[0] if yy is not None else None
        B = int(zt.shape[0])
        out = dev.empty(zt.shape) if want_z else None
        ld = dev.empty((B,))
        with dev.torch.cuda.device(dev.device):
            _lib.check(self._flow.lib.nf_nll(self._flow.ptr, zt.data_ptr(), yt.data_ptr() if yt is not None else None,
                                             B, C.byref(cond), None, None, ld.data_ptr(),
                                             out.data_ptr() if out is not None else None, None, _lib.NF_NO_PRIOR,
                                             dev.stream_ptr()))
        return (dev.back(out, was_np) if want_z else None), dev.back(ld, was_np)

    def _sample_dir(self, x, yy, cond):
        dev = self._dev
        xt, was_np = dev.to_dev(x, self.x_shape)
        yt = dev.to_dev(yy, self.x_shape)[0] if yy is not None else None
        out = dev.empty(xt.shape)
        with dev.torch.cuda.device(dev.device):
            _lib.check(self._flow.lib.nf_sample(self._flow.ptr, yt.data_ptr() if yt is not None else None,
                                                xt.data_ptr(), 0, 0, 1.0, int(xt.shape[0]), C.byref(cond),
                                                out.data_ptr(), dev.stream_ptr()))
        return dev.back(out, was_np)


class _Unconditional(_Bijector):
    def _forward(self, x):
        return self._sample_dir(x, None, self._cond())

    def _inverse(self, y):
        return self._nll_dir(y, None, self._cond())[0]

    def _inverse_log_det_jacobian(self, y):
        return self._nll_dir(y, None, self._cond(), want_z=False)[1]

    def _forward_log_det_jacobian(self, x):
        # the log-scale depends only on the pass-through half, identical in both directions
        return -self._inverse_log_det_jacobian(x)

    def _forward_and_log_det_jacobian(self, x):
        return self._forward(x), self._forward_log_det_jacobian(x)

    def _inverse_and_log_det_jacobian(self, y):
        return self._nll_dir(y, None, self._cond())


class _Conditional(_Bijector):
    conditional = True

    def _forward(self, x, yy, nlf0=None, nlf1=None, iso=None, cam=None):
        return self._sample_dir(x, yy, self._cond(nlf0, nlf1, iso, cam))

    def _inverse(self, y, yy, nlf0=None, nlf1=None, iso=None, cam=None):
        return self._nll_dir(y, yy, self._cond(nlf0, nlf1, iso, cam))[0]

    def _inverse_log_det_jacobian(self, z, yy, nlf0=None, nlf1=None, iso=None, cam=None):
        return self._nll_dir(z, yy, self._cond(nlf0, nlf1, iso, cam), want_z=False)[1]

    def _forward_log_det_jacobian(self, x, yy, nlf0=None, nlf1=None, iso=None, cam=None):
        # scale depends on (yy, iso, cam) only: log|det| of the two directions are negatives
        return -self._inverse_log_det_jacobian(x, yy, nlf0, nlf1, iso, cam)

    def _forward_and_log_det_jacobian(self, x, yy, nlf0=None, nlf1=None, iso=None, cam=None):
        return (self._forward(x, yy, nlf0, nlf1, iso, cam),
                self._forward_log_det_jacobian(x, yy, nlf0, nlf1, iso, cam))

    def _inverse_and_log_det_jacobian(self, y, yy, nlf0=None, nlf1=None, iso=None, cam=None):
        return self._nll_dir(y, yy, self._cond(nlf0, nlf1, iso, cam))


class Conv2d1x1(_Unconditional):
    """layers.py:74-145 (decomp='LU', bias=False): ``_inverse`` = z @ A,
    ``_forward`` = x @ A⁻¹, constant log|det| = H·W·Σ log_S."""


class Conv2d1x1LU2(Conv2d1x1):
    """decomp='LU2' (matrix_param.py:143-188): full-matrix L / U variables, float64 evaluation."""


class Conv2d1x1Dense(Conv2d1x1):
    """decomp='NONE' (matrix_param.py:23-29): A is the variable; A⁻¹ and log|det A| computed from it."""


class Permute(_Unconditional):
    """``tfb.Permute(permutation=[3, 2, 1, 0])`` of flow_permutation = 0 (noise_flow_model.py:80-84): log|det| = 0."""


class AffineCoupling(_Unconditional):
    """layers.py:251-375 with ``real_nvp_conv_template`` (layers.py:452-498)."""


class AffineCouplingSdnEx5(_Conditional):
    """AffineCouplingSdnEx5.py:22-132 + cond_utils.py:205-239."""


class AffineCouplingGainEx4(_Conditional):
    """AffineCouplingGainEx4.py:23-127 + cond_utils.py:432-440."""


class AffineCouplingSdnEx4(_Conditional):
    """AffineCouplingSdnEx4.py + cond_utils.py:178-202 (sdn5 without camera parameters)."""


class AffineCouplingSdn(_Conditional):
    """AffineCouplingSdn.py + cond_utils.py:41-52: scale = sqrt(sigmoid(b1)*y + sigmoid(b2))."""


class AffineCouplingGain(_Conditional):
    """AffineCouplingGain.py + cond_utils.py:319-330: scale = sigmoid(g1)*iso + sigmoid(g2);
    log|det| = -/+ log(scale) ONCE per patch, as the reference writes it (no H*W*C factor)."""


class AffineCouplingSdnEx1(_Conditional):
    """AffineCouplingSdnEx1.py + cond_utils.py:55-98: sqrt(sig(b1)*y/r_gain + sig(b2)), r_gain = exp(1e-2*rg[iso])*iso."""


class AffineCouplingSdnEx2(_Conditional):
    """AffineCouplingSdnEx2.py + cond_utils.py:101-138: sqrt(gain*(sig(b1)*y/gain + sig(b2))), gain = exp(0.1*g[iso])*iso."""


class AffineCouplingSdnEx3(_Conditional):
    """AffineCouplingSdnEx3.py + cond_utils.py:141-175: gain*sqrt(sig(b1)*y/gain + sig(b2))."""


class AffineCouplingSdnEx6(_Conditional):
    """AffineCouplingSdnEx6.py + cond_utils.py:242-276: SdnEx5 with ONE camera parameter (on the gain exponent)."""


class AffineCouplingGainEx1(_Conditional):
    """AffineCouplingGainEx1.py + cond_utils.py:333-350: exp(1e-5*g1)*iso + exp(1e-5*g2); log|det| once per patch."""


class AffineCouplingGainEx2(_Conditional):
    """AffineCouplingGainEx2.py + cond_utils.py:353-392: exp(0.1*g[iso])*iso; log|det| = -H*W*C*log(scale)."""


class AffineCouplingGainEx3(_Conditional):
    """AffineCouplingGainEx3.py + cond_utils.py:395-429: exp(1e-5*g[iso]); log|det| once per patch."""


_CLASS = {"sdn1": AffineCouplingSdnEx1, "sdn2": AffineCouplingSdnEx2, "sdn3": AffineCouplingSdnEx3, "sdn6": AffineCouplingSdnEx6,
          "gain1": AffineCouplingGainEx1, "gain2": AffineCouplingGainEx2, "gain3": AffineCouplingGainEx3,
          "sdn4": AffineCouplingSdnEx4, "sdn": AffineCouplingSdn, "gain": AffineCouplingGain, "conv1x1": Conv2d1x1, "conv1x1_lu2": Conv2d1x1LU2, "conv1x1_none": Conv2d1x1Dense, "permute": Permute, "coupling": AffineCoupling, "sdn5": AffineCouplingSdnEx5, "gain4": AffineCouplingGainEx4}


def bijectors_from_arch(arch: str, variables: Dict[str, np.ndarray], x_shape, width: int,
                        binding: str = "loss_first", device=None, flow_permutation: int = 1, decomp: str = "LU") -> List[_Bijector]:
    """The bijector list ``NoiseFlow.noise_flow_arch`` would build
    (noise_flow_model.py:71-235), each bound to its checkpoint variables."""
    specs = _params.parse_arch(arch, flow_permutation, decomp)
    tmpl = _params.template_binding(specs, binding)
    return [_CLASS[s.kind](s, variables, x_shape, width, tmpl, device) for s in specs]
