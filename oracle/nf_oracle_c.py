"""ctypes front-end of the plain-C oracle (``oracle/nf_oracle.c``) — TEST INFRASTRUCTURE ONLY.

Builds the C layer stream from the numpy oracle's bound layers (so the PLU / name binding is
shared, while every tensor operation is re-implemented independently in C, fp32, un-folded)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import nf_oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libnf_oracle_c.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "libnf_oracle_c.so"])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        lib = C.CDLL(_LIB)
        fp = C.POINTER(C.c_float)
        lib.nfo_nll.restype = C.c_int
        lib.nfo_nll.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_long, fp, fp, fp]
        lib.nfo_sample.restype = C.c_int
        lib.nfo_sample.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, C.c_float, fp, C.c_long, fp]
        lib.nfo_threads.restype = C.c_int
        lib.nfo_set_threads.argtypes = [C.c_int]
        _lib = lib
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


class COracle:
    def __init__(self, arch, variables, binding="loss_first"):
        self.layers = O.bind_variables(arch, variables, binding, np.float64)

    def _stream(self, H, W, iso, cam):
        s = []
        for L in self.layers:
            t = L["type"]
            if t == "conv1x1":
                s += [[1.0], L["A"].ravel(), L["A_inv"].ravel(), [L["log_abs_det"]]]
            elif t == "coupling":
                p = L["p"]
                w = p["l_2/W"].shape[-1]
                s += [[2.0, float(w)], p["l_1/W"].ravel(), p["l_1/b"], p["bn1/mean"], p["bn1/var"], p["l_2/W"].ravel(),
                      p["l_2/b"], p["bn2/mean"], p["bn2/var"], p["l_last/W"].ravel(), p["l_last/b"], p["l_last/logs"],
                      [p["rescaling_scale"]]]
            elif t == "sdn5":
                b1, b2, gain = O.sdn_ex5_scalars(L["p"], iso, cam)
                s += [[3.0, b1 / gain, b2]]
            elif t == "sdn4":
                one = np.ones((1, 1, 1, 1))
                v1 = float(O.sdn_ex4_scale(one, L["p"], iso).ravel()[0]) ** 2
                v0 = float(O.sdn_ex4_scale(0 * one, L["p"], iso).ravel()[0]) ** 2
                s += [[3.0, v1 - v0, v0]]
            elif t == "sdn":
                one = np.ones((1, 1, 1, 1))
                v1 = float(O.sdn_plain_scale(one, L["p"]).ravel()[0]) ** 2
                v0 = float(O.sdn_plain_scale(0 * one, L["p"]).ravel()[0]) ** 2
                s += [[3.0, v1 - v0, v0]]
            elif t == "gain":
                g = float(O.gain_plain_scale(L["p"], iso, np.float64))
                s += [[4.0, g, -np.log(g)]]
            else:  # gain4
                g = float(np.asarray(L["gain_val"]).reshape(-1)[0])
                s += [[4.0, g, -H * W * 4 * np.log(g)]]
        return np.ascontiguousarray(np.concatenate([np.asarray(a, np.float64).ravel() for a in s]), dtype=np.float32)

    def nll(self, x, y=None, iso=100.0, cam=2.0, want_z=False):
        """→ (nll[B], sd[B], z or None) float32."""
        lib = load()
        x = np.ascontiguousarray(x, np.float32)
        y = None if y is None else np.ascontiguousarray(y, np.float32)
        B, H, W, _ = x.shape
        st = self._stream(H, W, iso, cam)
        nll, sd = np.empty(B, np.float32), np.empty(B, np.float32)
        z = np.empty_like(x) if want_z else None
        rc = lib.nfo_nll(_fp(st), len(self.layers), H, W, _fp(x), _fp(y), B, _fp(nll), _fp(sd), _fp(z))
        if rc:
            raise RuntimeError("nfo_nll failed (%d)" % rc)
        return nll, sd, z

    def sample(self, eps, temp, y=None, iso=100.0, cam=2.0):
        lib = load()
        eps = np.ascontiguousarray(eps, np.float32)
        y = None if y is None else np.ascontiguousarray(y, np.float32)
        B, H, W, _ = eps.shape
        st = self._stream(H, W, iso, cam)
        out = np.empty_like(eps)
        rc = lib.nfo_sample(_fp(st), len(self.layers), H, W, _fp(eps), float(temp), _fp(y), B, _fp(out))
        if rc:
            raise RuntimeError("nfo_sample failed (%d)" % rc)
        return out

    @staticmethod
    def set_threads(n):
        load().nfo_set_threads(int(n))

    @staticmethod
    def threads():
        return int(load().nfo_threads())
