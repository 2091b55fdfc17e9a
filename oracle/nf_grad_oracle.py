"""CPU ORACLE for the TRAINING step (test infrastructure only — never imported by the product).

Restates, on torch-CPU float64 with autograd, what one ``sess.run([train_op, loss, sd_z],
{..., is_training: True})`` of the reference computes (train_noise_flow.py:64-66, 187-198):

* forward in the NLL direction with batch-statistics BN (layers.py:386-398) from the RAW
  checkpoint variables (PLU factors, CNN weights, sdn/gain parameters) — the same arithmetic as
  ``oracle/nf_oracle.py`` (which it is pinned against, ``tests/test_oracle.py``), written so that
  autograd can differentiate it;
* ``loss = mean_b nll_b`` (noise_flow_model.py:482-484) and its gradient w.r.t. every trainable
  variable (everything except the LU permutation / signs and the BN running statistics,
  train_noise_flow.py:309-312);
* the BN running-statistics EMA (layers.py:392-393);
* ``tf.train.AdamOptimizer(lr, 0.9, 0.999, 1e-8)`` / ``MomentumOptimizer(lr, 0.9)`` updates
  (train_noise_flow.py:187-198) restated from TensorFlow 1.12's documented update rules
  (``adam.py``: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMAs; theta -= lr_t*m/(sqrt(v)+eps);
  ``momentum.py``: accum = 0.9*accum + g; theta -= lr*accum).

Parity status: UNPINNED against the reference itself — TensorFlow 1.12 is not installable here, so
no reference-produced gradients exist; the oracle is pinned to ``nf_oracle`` (forward value) and to
central finite differences of that forward (gradients).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import nf_oracle as O

BN_EPS = O.BN_EPS
BN_DECAY = O.BN_DECAY
LOGSCALE_FACTOR = O.LOGSCALE_FACTOR


def is_trainable(name: str) -> bool:
    """train_noise_flow.py:309-312 / tf.trainable_variables(): P and sign_S are created with
    trainable=False (matrix_param.py:113-118), BN statistics too (layers.py:380-385)."""
    if "/P_conv2d_1x1" in name or "/sign_S_conv2d_1x1" in name:       # decomp = LU2 (matrix_param.py:151-157)
        return False
    return not ("/P_matpar" in name or "/sign_S_matpar" in name or name.endswith("/mean") or name.endswith("/var"))


def _tri_positions(n: int, upper: bool):
    """(rows, cols) of the matrix entry each vector element lands on (matrix_param.py:31-56)."""
    m = (n - 1) * n // 2
    probe = O.vec2stricttri(np.arange(1, m + 1, dtype=np.float64), upper)
    rows, cols = np.zeros(m, np.int64), np.zeros(m, np.int64)
    for i in range(n):
        for j in range(n):
            k = int(probe[i, j])
            if k:
                rows[k - 1], cols[k - 1] = i, j
    return torch.from_numpy(rows), torch.from_numpy(cols)


class GradOracle:
    def __init__(self, arch: str, variables: Dict[str, np.ndarray], binding: str = "loss_first", c_i: float = 1.0,
                 flow_permutation: int = 1, decomp: str = "LU", dtype=torch.float64):
        """``dtype=torch.float32`` gives the SAME op sequence in single precision — not a second oracle but a yardstick: how far a
        plain fp32 evaluation of the reference's graph is from the fp64 one on this input (gradients that are sums of large
        cancelling terms lose relative accuracy in ANY fp32 evaluation; the tests allow the kernels that much)."""
        self.dt = dtype
        self.npdt = np.float64 if dtype == torch.float64 else np.float32
        self.arch = arch
        self.binding = binding
        self.flow_permutation = int(flow_permutation)
        self.decomp = decomp
        self.c_i = float(c_i)
        self.names = list(variables.keys())
        self.shapes = {k: np.asarray(v).shape for k, v in variables.items()}
        self.t = {k: torch.tensor(np.asarray(v, self.npdt), dtype=self.dt, requires_grad=is_trainable(k))
                  for k, v in variables.items()}
        arch_l = O.parse_arch(arch)
        unc_ids = [i for lyr, i in arch_l if lyr == "unc"]
        order = unc_ids if binding == "loss_first" else unc_ids[::-1]
        self.tmpl_of = {i: k for k, i in enumerate(order)}
        self.arch_l = arch_l
        self.kink_ulps = 32.0          # see _relu
        self.kinks = []                # filled by forward(): activations within kink_ulps of their ReLU kink
        self.relu_flips = set()        # {(site, flat index)}: take the other ReLU branch there
        self._bcasts = []              # (u, u expanded) of every broadcast parameter-derived value (see _bc)
        self._convs = []               # (weight name, bias name, |input|, conv output, padding) of every convolution

    def _bc(self, u, like):
        """`u` (a value computed from variables: a scalar, a per-channel vector (1,C,1,1), a 4x4 matrix) broadcast against
        `like`, as an explicit expanded tensor whose gradient can be asked for: element e of that gradient is the TERM the
        element contributes to d loss / d u, and sum_e |term_e| is what the round-off of any fp32 evaluation of that sum
        scales with (loss_and_grads -> grad_abs_terms).  Numerically the identity."""
        shape = like if isinstance(like, (tuple, list, torch.Size)) else like.shape
        ue = u.expand(*shape)
        self._bcasts.append((u, ue))
        return ue

    # -- pieces -----------------------------------------------------------------------------
    def _A(self, i):
        """(A, log|det A|) of the mixing layer in front of coupling i for hps.flow_permutation / hps.decomp, or None."""
        if self.flow_permutation == 0:                                    # tfb.Permute(channels reversed)
            return torch.flip(torch.eye(4, dtype=self.dt), dims=[1]), torch.zeros((), dtype=self.dt)
        if self.flow_permutation != 1:
            return None
        n = O.conv1x1_variable_names(i, self.decomp)
        if self.decomp == "NONE":                                         # matrix_param.py:23-29
            A = self.t[n["A"]]
            return A, torch.linalg.slogdet(A)[1]
        if self.decomp == "LU2":                                          # matrix_param.py:143-188
            P, L, U = self.t[n["P"]], self.t[n["L"]], self.t[n["U"]]
            sgn, logS = self.t[n["sign_S"]].reshape(-1), self.t[n["log_S"]].reshape(-1)
            mask = torch.tril(torch.ones(4, 4, dtype=self.dt), -1)
            Lm = L * mask + torch.eye(4, dtype=self.dt)
            Um = U * mask.t() + torch.diag(sgn * torch.exp(logS))
            return P @ (Lm @ Um), logS.sum()
        return self._A_lu(i)

    def _A_lu(self, i):
        pre = "level0/bijector%d/Conv2d_1x1_%d/" % (i, i)
        sfx = "_matpar_lu_conv2d_1x1_%d_0" % i
        P, sgn, logS = self.t[pre + "P" + sfx], self.t[pre + "sign_S" + sfx], self.t[pre + "log_S" + sfx]
        Lv, Uv = self.t[pre + "L_vec" + sfx].reshape(-1), self.t[pre + "U_vec" + sfx].reshape(-1)
        n = logS.numel()
        lr, lc = _tri_positions(n, False)
        ur, uc = _tri_positions(n, True)
        L = torch.eye(n, dtype=self.dt).index_put((lr, lc), Lv)
        U = torch.diag(sgn.reshape(-1) * torch.exp(logS.reshape(-1))).index_put((ur, uc), Uv)
        return P @ (L @ U), logS.sum()                                    # matrix_param.py:130,138

    def _bn_train(self, h, scope, which, new_running):
        m = h.mean(dim=(0, 2, 3))
        v = h.var(dim=(0, 2, 3), unbiased=False)                          # tf.nn.moments
        for nm, val in (("mean", m), ("var", v)):
            key = scope + "bn_nvp_conv_%d/%s" % (which, nm)
            old = self.t[key].detach()
            new_running[key] = (old - BN_DECAY * (old - val.detach())).numpy().reshape(self.shapes[key])
        return (h - m[None, :, None, None]) / torch.sqrt(v[None, :, None, None] + BN_EPS)

    def _relu(self, hn, h, amp, site, err_in=None):
        """ReLU written as a constant 0/1 gate, so that a test can ask what the gradient would be with the OTHER branch at
        activations that no float32 evaluation can place on one side of the kink.

        An activation is "on its kink" when |h - batch mean| < kink_ulps * u, u = 2^-24 * (amp + |batch mean|) + err_in:
        amp = sum |input| * |weight| + |bias| is the magnitude of the terms whose rounded sum an fp32 evaluation compares
        with the mean, err_in the round-off its inputs already carry (per ulp; for l_2 that is l_1's u / sqrt(var + eps)
        pushed through |W2| — it dominates when a batch variance is far below BN's epsilon and the normalised activations
        are differences of nearly equal numbers).  So the margin is below the round-off of ANY fp32 summation order.  Those
        are recorded in self.kinks as (site, flat index, margin / u); `self.relu_flips` = set of (site, flat index) inverts
        the gate there (the forward value moves by the activation itself, i.e. by less than its own round-off).
        Returns (relu(hn), round-off unit of the output)."""
        with torch.no_grad():
            gate = (hn > 0)
            m = h.mean(dim=(0, 2, 3))
            v = h.var(dim=(0, 2, 3), unbiased=False)
            u = 2.0 ** -24 * (amp + m.abs()[None, :, None, None])
            if err_in is not None:
                u = u + err_in
            margin = (h - m[None, :, None, None]).abs() / (u + 1e-300)
            idx = torch.nonzero(margin.reshape(-1) < self.kink_ulps).reshape(-1)
            for k in idx.tolist():
                self.kinks.append((site, k, float(margin.reshape(-1)[k])))
            if self.relu_flips:
                flat = gate.reshape(-1).clone()
                for (st, k) in self.relu_flips:
                    if st == site:
                        flat[k] = ~flat[k]
                gate = flat.reshape(gate.shape)
            err_out = u / torch.sqrt(v[None, :, None, None] + BN_EPS)
        return hn * gate.to(hn.dtype), err_out

    def _cnn(self, z0, i, new_running):
        t = O.template_name(self.tmpl_of[i]) + "/"
        g = lambda k: self.t[t + k]
        w1 = g("l_1/W").permute(3, 2, 0, 1)
        h = F.conv2d(z0, w1, g("l_1/b").reshape(-1), padding=1)           # layers.py:469
        self._convs.append((t + "l_1/W", t + "l_1/b", z0.detach().abs(), h, 1))
        with torch.no_grad():
            amp = F.conv2d(z0.abs(), w1.abs(), g("l_1/b").reshape(-1).abs(), padding=1)
        a1, err1 = self._relu(self._bn_train(h, t, 1, new_running), h, amp, (i, 1))
        w2 = g("l_2/W").reshape(g("l_2/W").shape[-2], g("l_2/W").shape[-1]).t()[:, :, None, None]
        h = F.conv2d(a1, w2, g("l_2/b").reshape(-1))                      # :480
        self._convs.append((t + "l_2/W", t + "l_2/b", a1.detach().abs(), h, 0))
        with torch.no_grad():
            amp = F.conv2d(a1.abs(), w2.abs(), g("l_2/b").reshape(-1).abs())
            err2 = F.conv2d(err1, w2.abs())
        h, _ = self._relu(self._bn_train(h, t, 2, new_running), h, amp, (i, 2), err2)
        hp = F.pad(h, (1, 1, 1, 1))                                       # add_edge_padding, :555-583
        e = torch.zeros_like(hp[:, :1])
        e[:, :, 0, :] = 1
        e[:, :, -1, :] = 1
        e[:, :, :, 0] = 1
        e[:, :, :, -1] = 1
        w3 = g("l_last/W").permute(3, 2, 0, 1)
        o = F.conv2d(torch.cat([hp, e], 1), w3, g("l_last/b").reshape(-1))   # :665-670
        self._convs.append((t + "l_last/W", t + "l_last/b", torch.cat([hp, e], 1).detach().abs(), o, 0))
        o = o * self._bc(torch.exp(g("l_last/logs").reshape(1, -1, 1, 1) * LOGSCALE_FACTOR), o)   # :671-673
        c2 = o.shape[1] // 2
        return o[:, :c2], o[:, c2:]

    def _sdn5_scale(self, y, iso, cam):
        c = self.c_i
        cam_idx = int(cam)
        if float(cam) not in (0.0, 1.0, 2.0, 3.0, 4.0):
            raise IndexError("unknown camera id %r" % (cam,))
        cp = torch.exp(c * self.t["model/sdn_gain/cam_params"][:, cam_idx])
        ks = [k for k, v in enumerate(O.ISO_VALS) if float(v) == float(iso)]
        g = self.t["model/sdn_gain/gain_params"].reshape(-1)[ks[0]] if ks else torch.zeros((), dtype=self.dt)
        gain = torch.exp(c * g * cp[2]) * float(iso)
        b1 = torch.exp(c * self.t["model/sdn_gain/beta1"].reshape(-1)[0] * cp[0])
        b2 = torch.exp(c * self.t["model/sdn_gain/beta2"].reshape(-1)[0] * cp[1])
        return torch.sqrt(self._bc(b1, y) * y / self._bc(gain, y) + self._bc(b2, y))

    def _table(self, fmt, iso):
        """Entry of a per-ISO variable table: ISO 100..3200, anything else -> the ISO-800 entry (cond_utils.py:69-88)."""
        iso = int(iso) if float(iso) in [float(v) for v in O.ISO_TABLE] else 800
        return self.t[fmt % iso].reshape(-1)[0]

    def _sdn_other_scale(self, kind, y, iso, cam):
        """sdn_model_params / _ex1 / _ex2 / _ex3 / _ex6 (cond_utils.py:41-175, 242-276) in torch."""
        if kind == "sdn6":
            c = self.c_i
            if float(cam) not in (0.0, 1.0, 2.0, 3.0, 4.0):
                raise IndexError("unknown camera id %r" % (cam,))
            cp = torch.exp(c * self.t["model/sdn_gain/cam_params"].reshape(-1)[int(cam)])
            ks = [k for k, v in enumerate(O.ISO_VALS) if float(v) == float(iso)]
            g = self.t["model/sdn_gain/gain_params"].reshape(-1)[ks[0]] if ks else torch.zeros((), dtype=self.dt)
            gain = torch.exp(c * g * cp) * float(iso)
            b1 = torch.exp(c * self.t["model/sdn_gain/beta1"].reshape(-1)[0])
            b2 = torch.exp(c * self.t["model/sdn_gain/beta2"].reshape(-1)[0])
            return torch.sqrt(self._bc(b1, y) * y / self._bc(gain, y) + self._bc(b2, y))
        b1 = torch.sigmoid(self.t["model/b1"].reshape(-1)[0])
        b2 = torch.sigmoid(self.t["model/b2"].reshape(-1)[0])
        b1, b2 = self._bc(b1, y), self._bc(b2, y)
        if kind == "sdn":
            return torch.sqrt(b1 * y + b2)
        if kind == "sdn1":
            gain = self._bc(torch.exp(1e-2 * self._table("model/r_gain_param_%05d", iso)) * float(iso), y)
            return torch.sqrt(b1 * y / gain + b2)
        gain = self._bc(torch.exp(1e-1 * self._table("model/gain_param_%05d", iso)) * float(iso), y)
        if kind == "sdn2":
            return torch.sqrt(gain * (b1 * y / gain + b2))
        return gain * torch.sqrt(b1 * y / gain + b2)

    def _gain_other_scale(self, kind, iso):
        """gain_model_params / _ex1 / _ex2 / _ex3 (cond_utils.py:319-429) -> (scale, log-det over the whole patch?)."""
        if kind == "gain":
            return torch.sigmoid(self.t["model/g1"].reshape(-1)[0]) * float(iso) + torch.sigmoid(self.t["model/g2"].reshape(-1)[0]), False
        if kind == "gain1":
            return torch.exp(1e-5 * self.t["model/g1"].reshape(-1)[0]) * float(iso) + torch.exp(1e-5 * self.t["model/g2"].reshape(-1)[0]), False
        if kind == "gain2":
            return torch.exp(1e-1 * self._table("model/gain_param_%05d", iso)) * float(iso), True
        return torch.exp(1e-5 * self._table("model/gain_param_%05d", iso)), False

    # -- the step's forward -----------------------------------------------------------------
    def forward(self, x, y, iso, cam) -> Tuple[torch.Tensor, torch.Tensor, Dict[str, np.ndarray]]:
        """→ (loss, sd_z, new running statistics)."""
        z = torch.tensor(np.asarray(x, self.npdt)).permute(0, 3, 1, 2)
        yt = torch.tensor(np.asarray(y, self.npdt)).permute(0, 3, 1, 2) if y is not None else None
        B, C, H, W = z.shape
        obj = torch.zeros(B, dtype=self.dt)
        new_running: Dict[str, np.ndarray] = {}
        for lyr, i in self.arch_l:
            if lyr == "unc":
                mix = self._A(i)
                if mix is not None:
                    A, lad = mix
                    z = torch.einsum("bchw,bhwck->bkhw", z, self._bc(A, (B, H, W, C, C)))   # layers.py:117-130
                    obj = obj + H * W * self._bc(lad, obj)
                c2 = C // 2
                z0, z1 = z[:, :c2], z[:, c2:]
                shift, raw = self._cnn(z0, i, new_running)
                ls = self._bc(self.t["level0/bijector%d/rescaling_scale0" % i].reshape(()), raw) * torch.tanh(raw)
                z = torch.cat([z0, z1 * torch.exp(ls) + shift], 1)        # layers.py:355-375
                obj = obj + ls.sum(dim=(1, 2, 3))
            elif lyr == "sdn5":
                scale = self._sdn5_scale(yt, iso, cam)
                z = z / scale
                obj = obj - torch.log(scale).sum(dim=(1, 2, 3))
            elif lyr == "sdn4":                                           # cond_utils.py:178-202
                ks = [k for k, v in enumerate(O.ISO_VALS) if float(v) == float(iso)]
                gp = self.t["model/sdn_gain/gain_params"].reshape(-1)[ks[0]] if ks else torch.zeros((), dtype=self.dt)
                gain = self._bc(torch.exp(gp) * float(iso), yt)
                scale = torch.sqrt(self._bc(torch.exp(self.t["model/sdn_gain/beta1"].reshape(-1)[0]), yt) * yt / gain
                                   + self._bc(torch.exp(self.t["model/sdn_gain/beta2"].reshape(-1)[0]), yt))
                z = z / scale
                obj = obj - torch.log(scale).sum(dim=(1, 2, 3))
            elif lyr == "gain4":
                g = self.t["model/sdn_gain/gain_val"].reshape(-1)[0]
                z = z / self._bc(g, z)
                obj = obj - C * H * W * torch.log(self._bc(g, obj))
            elif lyr in ("sdn", "sdn1", "sdn2", "sdn3", "sdn6"):
                scale = self._sdn_other_scale(lyr, yt, iso, cam)
                z = z / scale
                obj = obj - torch.log(scale).sum(dim=(1, 2, 3))
            elif lyr in ("gain", "gain1", "gain2", "gain3"):
                s, full = self._gain_other_scale(lyr, iso)
                z = z / self._bc(s, z)
                obj = obj - (C * H * W if full else 1) * torch.log(self._bc(s, obj))     # GainEx2: the full sum; Gain / Ex1 / Ex3: once per patch
            else:
                raise ValueError("unknown layer %r" % lyr)
        logp = (-0.5 * (np.log(2 * np.pi) + z * z)).sum(dim=(1, 2, 3))
        self._parts = (-(obj.mean()), -(logp.mean()))     # loss = (- log-det part) + (- prior part)
        nll = -(obj + logp)
        sd_z = torch.sqrt(z.var(dim=(1, 2, 3), unbiased=False)).mean()
        return nll.mean(), sd_z, new_running

    def loss_and_grads(self, x, y, iso, cam, relu_flips=()):
        """→ (loss, sd_z, {name: d loss / d variable} for trainables, new running statistics).  `self.kinks` afterwards lists
        the activations that sit on their ReLU kink (see _relu); `relu_flips` takes the other branch at some of them."""
        self.relu_flips = set((tuple(s), int(k)) for s, k in relu_flips)
        self.kinks = []
        self._bcasts, self._convs = [], []
        for v in self.t.values():
            if v.grad is not None:
                v.grad = None
        loss, sd_z, new_running = self.forward(x, y, iso, cam)
        # d loss / d theta = (gradient of the log-det part) + (gradient of the prior part).  Near the optimum the two cancel
        # (that IS the optimality condition: e.g. d/d gain of  -sum log s  against  sum z^2 / 2), so a gradient can be 1e-4 of
        # the terms it is the sum of — and no fp32 evaluation resolves it better than ~1e-7 of THOSE.
        #   self.grad_abs_terms[name]  = per entry, sum_e |term_e| over the batch x pixel elements whose contributions the
        #                                entry is the sum of, the two parts counted separately: the scale the round-off of ANY
        #                                fp32 evaluation of that sum goes with, computed here in fp64 from the model and the
        #                                input alone (the tests' noise allowance: c * 2^-24 * this — nothing a kernel computes)
        #   self.grad_terms[name]      = max |part a| + |part b| (coarser, kept for reference)
        names = [k for k, v in self.t.items() if v.requires_grad]
        leaves = [self.t[k] for k in names]
        self._bcasts = [(u, ue) for u, ue in self._bcasts if ue.requires_grad]     # constants (a channel permutation) have no terms
        probes = [ue for _, ue in self._bcasts] + [o for (_, _, _, o, _) in self._convs]
        nb = len(self._bcasts)
        ga_all = torch.autograd.grad(self._parts[0], leaves + probes, retain_graph=True, allow_unused=True)
        gb_all = torch.autograd.grad(self._parts[1], leaves + probes, retain_graph=True, allow_unused=True)
        zero = lambda g, ref: torch.zeros_like(ref) if g is None else g   # noqa: E731
        terms = {k: torch.zeros_like(v) for k, v in zip(names, leaves)}
        # (i) broadcast values: |d u_i / d theta| x sum_e |term_e(u_i)|
        for j, (u, ue) in enumerate(self._bcasts):
            ta = zero(ga_all[len(leaves) + j], ue).abs() + zero(gb_all[len(leaves) + j], ue).abs()
            while ta.dim() > u.dim():                       # sum over the broadcast dimensions: leading ones first ...
                ta = ta.sum(dim=0)
            for d in range(u.dim()):                        # ... then the size-1 dimensions of u
                if u.shape[d] == 1 and ta.shape[d] != 1:
                    ta = ta.sum(dim=d, keepdim=True)
            if not u.requires_grad:
                continue
            uf, tf = u.reshape(-1), ta.reshape(-1)
            for i in range(uf.numel()):
                if float(tf[i]) == 0.0:
                    continue
                gi = torch.autograd.grad(uf[i], leaves, retain_graph=True, allow_unused=True)
                for k, g in zip(names, gi):
                    if g is not None:
                        terms[k] = terms[k] + g.abs() * tf[i]
        # (ii) convolutions: d/dW[k][j] = sum_{b,p} in[b, p + tap, k] * gout[b, p, j]  ->  the same sum over |in| |gout|
        for c, (wn, bn, ain, o, pad) in enumerate(self._convs):
            gout = zero(ga_all[len(leaves) + nb + c], o).abs() + zero(gb_all[len(leaves) + nb + c], o).abs()
            W = self.t[wn]
            kh, kw = (W.shape[0], W.shape[1]) if W.dim() == 4 else (1, 1)
            tw = torch.nn.grad.conv2d_weight(ain, (gout.shape[1], ain.shape[1], kh, kw), gout, padding=pad)   # [out, in, kh, kw]
            if W.dim() == 4:
                terms[wn] = terms[wn] + tw.permute(2, 3, 1, 0).reshape(W.shape)
            else:
                terms[wn] = terms[wn] + tw[:, :, 0, 0].t().reshape(W.shape)
            terms[bn] = terms[bn] + gout.sum(dim=(0, 2, 3)).reshape(self.t[bn].shape)
        grads, self.grad_terms, self.grad_abs_terms = {}, {}, {}
        for k, v, a, b in zip(names, leaves, ga_all, gb_all):
            a, b = zero(a, v), zero(b, v)
            grads[k] = (a + b).detach().numpy().reshape(self.shapes[k]).copy()
            self.grad_terms[k] = float((a.abs() + b.abs()).max())
            self.grad_abs_terms[k] = terms[k].detach().numpy().reshape(self.shapes[k]).copy()
        return float(loss.detach()), float(sd_z.detach()), grads, new_running


# ------------------------------------------------------------------------------------------
# optimizers (train_noise_flow.py:187-198)
# ------------------------------------------------------------------------------------------
def adam_step(variables, grads, state, lr, beta1=0.9, beta2=0.999, eps=1e-8, dtype=np.float64):
    """One ``tf.train.AdamOptimizer.minimize`` update.  ``state`` = {"t": int, "m": {}, "v": {}}
    (mutated); returns the updated variables (new dict)."""
    state["t"] = t = state.get("t", 0) + 1
    lr_t = dtype(lr) * np.sqrt(1.0 - dtype(beta2) ** t) / (1.0 - dtype(beta1) ** t)
    out = dict(variables)
    for k, g in grads.items():
        g = np.asarray(g, dtype)
        m = state.setdefault("m", {}).get(k, np.zeros_like(g))
        v = state.setdefault("v", {}).get(k, np.zeros_like(g))
        m = beta1 * m + (1 - beta1) * g
        v = beta2 * v + (1 - beta2) * g * g
        state["m"][k], state["v"][k] = m, v
        out[k] = (np.asarray(variables[k], dtype) - lr_t * m / (np.sqrt(v) + eps)).astype(np.asarray(variables[k]).dtype)
    return out


def momentum_step(variables, grads, state, lr, momentum=0.9, dtype=np.float64):
    """One ``tf.train.MomentumOptimizer(lr, 0.9).minimize`` update."""
    out = dict(variables)
    for k, g in grads.items():
        g = np.asarray(g, dtype)
        a = momentum * state.setdefault("a", {}).get(k, np.zeros_like(g)) + g
        state["a"][k] = a
        out[k] = (np.asarray(variables[k], dtype) - dtype(lr) * a).astype(np.asarray(variables[k]).dtype)
    return out


def train_step(arch, variables, x, y, iso, cam, state, lr, binding="loss_first", optim="adam"):
    """The whole ``sess.run([train_op, loss, sd_z])``: → (new variables, loss, sd_z)."""
    o = GradOracle(arch, variables, binding)
    loss, sd_z, grads, new_running = o.loss_and_grads(x, y, iso, cam)
    step = adam_step if optim == "adam" else momentum_step
    new_vars = step(variables, grads, state, lr)
    for k, v in new_running.items():
        new_vars[k] = v.astype(np.asarray(variables[k]).dtype)
    return new_vars, loss, sd_z
