"""CPU ORACLE, baseline flavour (test infrastructure only) — the same restatement as
``oracle/nf_oracle.py`` but op-per-layer on torch-CPU fp32 with all host cores,
mirroring how the reference's TF1 graph executes: three separate convolutions,
separate BN / ReLU / tanh / exp / reduce ops per coupling block, every
intermediate materialised (SURVEY.md §8d "CPU baseline").  Used ONLY by
``bench.py``'s ``cpu_baseline`` leg (kind = "port": TF1 itself is unavailable)
and by tests that pin it to the fp64 numpy oracle.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .nf_oracle import NoiseFlowOracle, sdn_ex5_scalars


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32)))


class TorchCpuFlow:
    """NHWC in/out like the reference; NCHW internally (what MKL-DNN convs want)."""

    def __init__(self, arch: str, variables: Dict[str, np.ndarray], binding: str = "loss_first"):
        ref = NoiseFlowOracle(arch, variables, binding, dtype=np.float64)
        self.layers = []
        for L in ref.layers:
            if L["type"] == "conv1x1":
                # z @ A per pixel == conv2d with weight[k, c] = A[c, k]
                self.layers.append(("conv1x1", _t(L["A"].T)[:, :, None, None], _t(L["A_inv"].T)[:, :, None, None],
                                    float(L["log_abs_det"])))
            elif L["type"] == "coupling":
                p = L["p"]
                q = {
                    "w1": _t(np.transpose(p["l_1/W"], (3, 2, 0, 1))), "b1": _t(p["l_1/b"]),
                    "m1": _t(p["bn1/mean"]), "v1": _t(p["bn1/var"]),
                    "w2": _t(np.transpose(p["l_2/W"], (3, 2, 0, 1))), "b2": _t(p["l_2/b"]),
                    "m2": _t(p["bn2/mean"]), "v2": _t(p["bn2/var"]),
                    "w3": _t(np.transpose(p["l_last/W"], (3, 2, 0, 1))), "b3": _t(p["l_last/b"]),
                    "logs": _t(p["l_last/logs"]), "s": float(p["rescaling_scale"]),
                }
                self.layers.append(("coupling", q))
            elif L["type"] == "sdn5":
                self.layers.append(("sdn5", L["p"]))
            else:
                self.layers.append(("gain4", float(np.asarray(L["gain_val"]).reshape(-1)[0])))

    @staticmethod
    def _cnn(z0, q):
        h = F.conv2d(z0, q["w1"], q["b1"], padding=1)
        h = (h - q["m1"][None, :, None, None]) / torch.sqrt(q["v1"][None, :, None, None] + 1e-4)
        h = torch.relu(h)
        h = F.conv2d(h, q["w2"], q["b2"])
        h = (h - q["m2"][None, :, None, None]) / torch.sqrt(q["v2"][None, :, None, None] + 1e-4)
        h = torch.relu(h)
        hp = F.pad(h, (1, 1, 1, 1))
        e = torch.ones_like(hp[:, :1])
        e[:, :, 1:-1, 1:-1] = 0
        o = F.conv2d(torch.cat([hp, e], 1), q["w3"], q["b3"])
        o = o * torch.exp(q["logs"] * 3.0)[None, :, None, None]
        return o[:, :2], o[:, 2:]

    def nll(self, x, y, iso, cam):
        """→ (nll[B] float32 tensor, sd_z float)."""
        with torch.no_grad():
            z = _t(x).permute(0, 3, 1, 2).contiguous()
            yy = None if y is None else _t(y).permute(0, 3, 1, 2).contiguous()
            obj = torch.zeros(z.shape[0])
            hw = z.shape[2] * z.shape[3]
            for L in self.layers:
                if L[0] == "conv1x1":
                    z = F.conv2d(z, L[1])
                    obj = obj + hw * L[3]
                elif L[0] == "coupling":
                    q = L[1]
                    z0, z1 = z[:, :2], z[:, 2:]
                    shift, raw = self._cnn(z0, q)
                    ls = q["s"] * torch.tanh(raw)
                    z = torch.cat([z0, z1 * torch.exp(ls) + shift], 1)
                    obj = obj + ls.sum(dim=(1, 2, 3))
                elif L[0] == "sdn5":
                    b1, b2, gain = sdn_ex5_scalars(L[1], iso, cam, dtype=np.float32)
                    scale = torch.sqrt(float(b1) * yy / float(gain) + float(b2))
                    z = z / scale
                    obj = obj - torch.log(scale).sum(dim=(1, 2, 3))
                else:
                    z = z / L[1]
                    obj = obj - z[0].numel() * math.log(L[1])
            obj = obj + (-0.5 * (math.log(2 * math.pi) + z * z)).sum(dim=(1, 2, 3))
            sd = torch.sqrt(z.var(dim=(1, 2, 3), unbiased=False)).mean()
            return -obj, float(sd)

    def sample(self, eps, temp, y, iso, cam):
        with torch.no_grad():
            x = _t(eps).permute(0, 3, 1, 2).contiguous() * float(temp)
            yy = None if y is None else _t(y).permute(0, 3, 1, 2).contiguous()
            for L in reversed(self.layers):
                if L[0] == "conv1x1":
                    x = F.conv2d(x, L[2])
                elif L[0] == "coupling":
                    q = L[1]
                    x0, x1 = x[:, :2], x[:, 2:]
                    shift, raw = self._cnn(x0, q)
                    ls = q["s"] * torch.tanh(raw)
                    x = torch.cat([x0, (x1 - shift) * torch.exp(-ls)], 1)
                elif L[0] == "sdn5":
                    b1, b2, gain = sdn_ex5_scalars(L[1], iso, cam, dtype=np.float32)
                    x = x * torch.sqrt(float(b1) * yy / float(gain) + float(b2))
                else:
                    x = x * L[1]
            return x.permute(0, 2, 3, 1).contiguous()
