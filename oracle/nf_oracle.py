"""CPU ORACLE — test infrastructure only.

A numpy restatement of the reference's Noise Flow bijector stack (SURVEY.md
Appendix A), used ONLY as the checker by ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py``.  Nothing under ``noise_flow_amd/``
imports it; the product path is the HIP library and fails loudly without it.

PARITY PINNING: the reference cannot run here (TensorFlow 1.12 / TFP 0.5 are not
installed, there is no network) and it ships no tests or golden vectors, so this
oracle is **"parity unpinned"** by reference outputs.  What pins it instead
(tests/test_oracle.py): the shipped checkpoint (143 tensors / 2433 trainable
parameters), fresh-init analytic known answers (NLL = ½·HWC·log2π + ½‖x‖² for
``unc`` stacks, closed-form ``sdn5`` scale), structural invariants
(sample∘nll = id, slogdet(A) = Σ log_S, log-det vs a finite-difference Jacobian
on a toy patch) and the plausibility band of the shipped model on S6-NLF noise
(NLL/dim within 0.05 nat of the generating density, sd_z in [0.8, 1.0]).

Every function cites the reference file:line (relative to /root/reference) it
follows.  ``dtype=np.float64`` is the truth the HIP path is compared against;
``dtype=np.float32`` follows the same op order in single precision (what the TF1
CPU graph would compute, up to its unknowable accumulation order).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

ISO_VALS = (100.0, 400.0, 800.0, 1600.0, 3200.0)   # cond_utils.py:225
CAM_NAMES = ("IP", "GP", "S6", "N6", "G4")          # cond_utils.py:213, sidd_utils.py:262
BN_EPS = 1e-4                                       # layers.py:378
BN_DECAY = 0.1                                      # layers.py:378
LOGSCALE_FACTOR = 3.0                               # layers.py:653


# ----------------------------------------------------------------------------
# matrix_param.py — PLU parameterisation of the 1x1 conv
# ----------------------------------------------------------------------------
def fill_triangular(v: np.ndarray, upper: bool) -> np.ndarray:
    """``tfdist.fill_triangular`` (TF 1.12 ``distributions/util.py``) for a 1-D
    vector of length n(n+1)/2, as called from matrix_param.py:44."""
    v = np.asarray(v)
    m = v.shape[-1]
    n = int(round((np.sqrt(8 * m + 1) - 1) / 2))
    if n * (n + 1) // 2 != m:
        raise ValueError("vector length %d is not triangular" % m)
    if upper:
        flat = np.concatenate([v, v[n:][::-1]])
        return np.triu(flat.reshape(n, n))
    flat = np.concatenate([v[n:], v[::-1]])
    return np.tril(flat.reshape(n, n))


def fill_triangular_inverse(mat: np.ndarray, upper: bool) -> np.ndarray:
    """``tfdist.fill_triangular_inverse`` as used at matrix_param.py:87: element k
    of the vector is read from where :func:`fill_triangular` would put it."""
    mat = np.asarray(mat)
    n = mat.shape[-1]
    m = n * (n + 1) // 2
    probe = fill_triangular(np.arange(1, m + 1, dtype=np.float64), upper)
    out = np.zeros(m, dtype=mat.dtype)
    for i in range(n):
        for j in range(n):
            k = int(probe[i, j])
            if k:
                out[k - 1] = mat[i, j]
    return out


def vec2stricttri(v: np.ndarray, upper: bool) -> np.ndarray:
    """matrix_param.py:31-56: fill the (n-1)x(n-1) triangle, then pad to n x n
    so the result is *strictly* triangular."""
    base = fill_triangular(v, upper)
    if upper:   # one zero row at the bottom, one zero column on the left
        return np.pad(base, [(0, 1), (1, 0)])
    return np.pad(base, [(1, 0), (0, 1)])   # zero row on top, zero column on the right


def stricttri2vec(mat: np.ndarray, upper: bool) -> np.ndarray:
    """matrix_param.py:59-97 (inverse of :func:`vec2stricttri`)."""
    mat = np.asarray(mat)
    if upper:
        trim = np.triu(mat[:-1, 1:])
    else:
        trim = np.tril(mat[1:, :-1])
    return fill_triangular_inverse(trim, upper)


def matrix_param_lu(P, sign_S, log_S, L_vec, U_vec, dtype=np.float64):
    """matrix_param.py:100-140 → (A, A_inv, log_abs_det)."""
    P = np.asarray(P, dtype)
    log_S = np.asarray(log_S, dtype)
    n = log_S.shape[0]
    L = vec2stricttri(np.asarray(L_vec, dtype), upper=False) + np.eye(n, dtype=dtype)
    U = vec2stricttri(np.asarray(U_vec, dtype), upper=True) + np.diag(np.asarray(sign_S, dtype) * np.exp(log_S))
    A = P @ (L @ U)                                              # :130
    import scipy.linalg as sla
    inner = sla.solve_triangular(L, P.T, lower=True)             # :135-136
    A_inv = sla.solve_triangular(U, inner, lower=False)
    return A.astype(dtype), A_inv.astype(dtype), dtype(np.sum(log_S))   # :138


def matrix_param_none(A, dtype=np.float64):
    """matrix_param.py:23-29 (decomp 'NONE'): the matrix itself is the variable."""
    A = np.asarray(A, dtype)
    sign, lad = np.linalg.slogdet(A.astype(np.float64))
    return A, np.linalg.inv(A.astype(np.float64)).astype(dtype), dtype(lad)


def matrix_param_lu2(P, L, sign_S, log_S, U, dtype=np.float64):
    """matrix_param.py:143-188 (decomp 'LU2'): full L / U variables masked to their strict
    triangles, everything evaluated in float64 and cast back (``dtype2 = 'float64'``, :164)."""
    P, L, U = (np.asarray(a, np.float64) for a in (P, L, U))
    sign_S, log_S = np.asarray(sign_S, np.float64), np.asarray(log_S, np.float64)
    n = log_S.shape[0]
    mask = np.tril(np.ones((n, n)), -1)
    Lm = L * mask + np.eye(n)                                      # :172
    Um = U * mask.T + np.diag(sign_S * np.exp(log_S))              # :173
    A = P @ (Lm @ Um)                                              # :174
    A_inv = np.linalg.inv(Um) @ (np.linalg.inv(Lm) @ np.linalg.inv(P))   # :177-180
    return A.astype(dtype), A_inv.astype(dtype), dtype(np.sum(log_S))


def conv1x1_variable_names(i: int, decomp: str = "LU") -> Dict[str, str]:
    """Checkpoint names of Conv2d_1x1_i's variables for each ``hps.decomp`` (matrix_param.py:24, :109-123, :151-161)."""
    pre = "level0/bijector%d/Conv2d_1x1_%d/" % (i, i)
    nm = "conv2d_1x1_%d_0" % i
    if decomp == "LU":
        return {k: pre + k + "_matpar_lu_" + nm for k in ("P", "sign_S", "log_S", "L_vec", "U_vec")}
    if decomp == "LU2":
        return {"P": pre + "P_" + nm, "L": pre + "L_filters_" + nm, "sign_S": pre + "sign_S_" + nm,
                "log_S": pre + "log_S_filters_" + nm, "U": pre + "U_filters_" + nm}
    if decomp == "NONE":
        return {"A": pre + "A_matpar_none_" + nm}
    raise ValueError("hps.decomp must be 'LU', 'LU2' or 'NONE' (matrix_param.py:191-193), got %r" % (decomp,))


def conv1x1_from_variables(variables, i: int, decomp: str, dtype):
    n = conv1x1_variable_names(i, decomp)
    if decomp == "LU":
        return matrix_param_lu(variables[n["P"]], variables[n["sign_S"]], variables[n["log_S"]], variables[n["L_vec"]],
                               variables[n["U_vec"]], dtype)
    if decomp == "LU2":
        return matrix_param_lu2(variables[n["P"]], variables[n["L"]], variables[n["sign_S"]], variables[n["log_S"]],
                                variables[n["U"]], dtype)
    return matrix_param_none(variables[n["A"]], dtype)


def conv1x1_init_variables(q: np.ndarray, i: int, decomp: str) -> Dict[str, np.ndarray]:
    """Initial variables of Conv2d_1x1_i from its QR-orthogonal start matrix (layers.py:95)."""
    import scipy.linalg as sla
    n = conv1x1_variable_names(i, decomp)
    if decomp == "LU":
        return {n[k]: a for k, a in lu_init_from_matrix(q).items()}
    if decomp == "NONE":
        return {n["A"]: q.astype(np.float32)}
    p, l, u = sla.lu(q)                                            # matrix_param.py:145-149
    s = np.diag(u)
    return {n["P"]: p.astype(np.float32), n["L"]: l.astype(np.float32), n["sign_S"]: np.sign(s).astype(np.float32),
            n["log_S"]: np.log(np.abs(s)).astype(np.float32), n["U"]: np.triu(u, k=1).astype(np.float32)}


def lu_init_from_matrix(A0: np.ndarray) -> Dict[str, np.ndarray]:
    """matrix_param.py:100-123: initial (P, sign_S, log_S, L_vec, U_vec) from a matrix."""
    import scipy.linalg as sla
    p, l, u = sla.lu(A0)
    s = np.diag(u)
    return {
        "P": p.astype(np.float32),
        "sign_S": np.sign(s).astype(np.float32),
        "log_S": np.log(np.abs(s)).astype(np.float32),
        "L_vec": stricttri2vec(l, upper=False).astype(np.float32),
        "U_vec": stricttri2vec(np.triu(u, k=1), upper=True).astype(np.float32),
    }


# ----------------------------------------------------------------------------
# layers.py — conv wrappers, batch norm, coupling CNN
# ----------------------------------------------------------------------------
def conv2d_nhwc(x: np.ndarray, w: np.ndarray, pad_same: bool) -> np.ndarray:
    """``tf.nn.conv2d`` NHWC, stride 1 (cross-correlation), 'SAME' zero pad or
    'VALID' (layers.py:604, :665)."""
    kh, kw, cin, cout = w.shape
    if pad_same:
        a, b = (kh - 1) // 2, (kw - 1) // 2
        x = np.pad(x, [(0, 0), (a, a), (b, b), (0, 0)])
    n, hp, wp, _ = x.shape
    ho, wo = hp - kh + 1, wp - kw + 1
    out = np.zeros((n, ho, wo, cout), dtype=x.dtype)
    for di in range(kh):
        for dj in range(kw):
            out += x[:, di:di + ho, dj:dj + wo, :] @ w[di, dj]
    return out


def add_edge_padding(x: np.ndarray) -> np.ndarray:
    """layers.py:555-583 for a 3x3 filter: zero-pad by one pixel and append a
    channel that is 1 on the outer ring of the padded map."""
    xp = np.pad(x, [(0, 0), (1, 1), (1, 1), (0, 0)])
    e = np.zeros(xp.shape[:3] + (1,), dtype=x.dtype)
    e[:, :1, :, 0] = 1
    e[:, -1:, :, 0] = 1
    e[:, :, :1, 0] = 1
    e[:, :, -1:, 0] = 1
    return np.concatenate([xp, e], axis=3)


def batch_norm(h, mean, var, training: bool):
    """layers.py:378-401.  Eval: stored statistics.  Training: moments over
    (N,H,W); returns the EMA-updated running stats as well."""
    dt = h.dtype.type
    if training:
        m = h.mean(axis=(0, 1, 2))
        v = h.var(axis=(0, 1, 2))
        new_mean = mean - dt(BN_DECAY) * (mean - m)
        new_var = var - dt(BN_DECAY) * (var - v)
        return (h - m) / np.sqrt(v + dt(BN_EPS)), new_mean, new_var
    return (h - mean) / np.sqrt(var + dt(BN_EPS)), mean, var


def coupling_cnn(z0: np.ndarray, p: Dict[str, np.ndarray], training: bool = False, record=None):
    """real_nvp_conv_template._fn, layers.py:463-497 → (shift, raw_log_scale).
    ``record`` (a dict, training mode) receives the batch moments used and the EMA-updated
    running statistics the reference's assign_sub ops would leave behind (layers.py:392-393)."""
    dt = z0.dtype.type
    h = conv2d_nhwc(z0, p["l_1/W"], True) + p["l_1/b"].reshape(1, 1, 1, -1)       # :469, 586-613
    if training and record is not None:
        record["mean1"], record["var1"] = h.mean(axis=(0, 1, 2)), h.var(axis=(0, 1, 2))
    h, nm1, nv1 = batch_norm(h, p["bn1/mean"], p["bn1/var"], training)           # :472-477
    h = np.maximum(h, dt(0))                                                     # :478
    h = conv2d_nhwc(h, p["l_2/W"], True) + p["l_2/b"].reshape(1, 1, 1, -1)        # :480
    if training and record is not None:
        record["mean2"], record["var2"] = h.mean(axis=(0, 1, 2)), h.var(axis=(0, 1, 2))
    h, nm2, nv2 = batch_norm(h, p["bn2/mean"], p["bn2/var"], training)           # :483-488
    if training and record is not None:
        record.update(new_mean1=nm1, new_var1=nv1, new_mean2=nm2, new_var2=nv2)
    h = np.maximum(h, dt(0))                                                     # :489
    o = conv2d_nhwc(add_edge_padding(h), p["l_last/W"], False)                   # :491, 651-666
    o = o + p["l_last/b"].reshape(1, 1, 1, -1)                                   # :670
    o = o * np.exp(p["l_last/logs"].reshape(1, 1, 1, -1) * dt(LOGSCALE_FACTOR))  # :671-673
    c2 = o.shape[-1] // 2
    return o[..., :c2], o[..., c2:]                                              # :494 tf.split


def _h(a):
    """Round to IEEE half (round-to-nearest-even) and come back in the working dtype."""
    a = np.asarray(a)
    return a.astype(np.float16).astype(a.dtype)


def coupling_cnn_fp16(z0: np.ndarray, p: Dict[str, np.ndarray]):
    """The coupling CNN as the HIP library's fp16 mode evaluates it (BASELINE configs[4]:
    "fp16 coupling CNN with fp32 log-det accumulate"): BN-eval and exp(3*logs) are folded
    into the conv weights FIRST (as csrc/nf_host.hip does), then the folded weights and the
    three CNN inputs (z0, relu(h1), relu(h2)) are rounded to fp16 (the raw half of l_last's output channels
    after one more fold, see below); biases, the border table
    and every accumulation stay in fp32 (here: the working dtype).  Same math as
    :func:`coupling_cnn` when nothing is rounded."""
    dt = z0.dtype.type
    s1 = 1.0 / np.sqrt(p["bn1/var"] + dt(BN_EPS))
    s2 = 1.0 / np.sqrt(p["bn2/var"] + dt(BN_EPS))
    es = np.exp(p["l_last/logs"].reshape(-1) * dt(LOGSCALE_FACTOR))
    w = p["l_2/W"].shape[-1]
    W1 = _h(p["l_1/W"] * s1)
    b1 = ((p["l_1/b"] - p["bn1/mean"]) * s1).astype(np.float32).astype(z0.dtype)
    W2 = _h(p["l_2/W"] * s2)
    b2 = ((p["l_2/b"] - p["bn2/mean"]) * s2).astype(np.float32).astype(z0.dtype)
    W3 = p["l_last/W"] * es
    W3h = W3.copy()
    W3h[:, :, :w, :] = _h(W3[:, :, :w, :])                       # activations x fp16 weights ...
    # ... the raw (log-scale) half of the output channels carries the 2*log2(e) of tanh's  t = exp2(2 log2(e) raw)  inside the
    # rounded weight (csrc/nf_host.hip::to_half_w3: folded BEFORE the rounding, like the BN scale and exp(3 logs))
    c2o = W3.shape[-1] // 2
    k2 = dt(2.0 * 1.4426950408889634)
    W3h[:, :, :w, c2o:] = _h(W3[:, :, :w, c2o:] * k2) / k2
    h = np.maximum(conv2d_nhwc(_h(z0), W1, True) + b1, dt(0))
    h = np.maximum(conv2d_nhwc(_h(h), W2, True) + b2, dt(0))
    hp = add_edge_padding(_h(h))                                 # ... the edge channel stays exact (fp32 table)
    o = conv2d_nhwc(hp, W3h, False) + (p["l_last/b"] * es).reshape(1, 1, 1, -1)
    c2 = o.shape[-1] // 2
    return o[..., :c2], o[..., c2:]


def coupling_cnn_fp16_plain(z0: np.ndarray, p: Dict[str, np.ndarray]):
    """An independent half-precision evaluation of real_nvp_conv_template (layers.py:463-497), op by op as the reference
    writes it and WITHOUT the library's folding: every conv takes its input and its RAW weights rounded to fp16 and
    accumulates in the working dtype; bias, batch norm (stored statistics), ReLU and the exp(3*logs) scaling are applied
    afterwards, unrounded.  It shares no rounding point with csrc/nf_host.hip's folded weights, so it measures how far
    ANY fp16 evaluation of this CNN sits from the fp32 one — the yardstick the library's fp16 mode is held to."""
    dt = z0.dtype.type
    h = conv2d_nhwc(_h(z0), _h(p["l_1/W"]), True) + p["l_1/b"].reshape(1, 1, 1, -1)
    h, _, _ = batch_norm(h, p["bn1/mean"], p["bn1/var"], False)
    h = np.maximum(h, dt(0))
    h = conv2d_nhwc(_h(h), _h(p["l_2/W"]), True) + p["l_2/b"].reshape(1, 1, 1, -1)
    h, _, _ = batch_norm(h, p["bn2/mean"], p["bn2/var"], False)
    h = np.maximum(h, dt(0))
    w = p["l_2/W"].shape[-1]
    W3 = p["l_last/W"].copy()
    W3[:, :, :w, :] = _h(W3[:, :, :w, :])                      # the 0/1 edge channel and its weights stay exact
    o = conv2d_nhwc(add_edge_padding(_h(h)), W3, False) + p["l_last/b"].reshape(1, 1, 1, -1)
    o = o * np.exp(p["l_last/logs"].reshape(1, 1, 1, -1) * dt(LOGSCALE_FACTOR))
    c2 = o.shape[-1] // 2
    return o[..., :c2], o[..., c2:]


# ----------------------------------------------------------------------------
# bijectors
# ----------------------------------------------------------------------------
def _coupling_cnn_any(z0, p, training, cnn_fp16, record):
    """``cnn_fp16``: False = all working-dtype; True = the library's rounding points; 'plain' = the unfolded fp16 yardstick."""
    if cnn_fp16 == "plain":
        return coupling_cnn_fp16_plain(z0, p)
    return coupling_cnn_fp16(z0, p) if cnn_fp16 else coupling_cnn(z0, p, training, record)


def affine_coupling_inverse(z, p, training=False, cnn_fp16=False, record=None):
    """AffineCoupling._inverse_and_log_det_jacobian, layers.py:355-375 (NLL direction)."""
    c2 = z.shape[-1] // 2
    z0, z1 = z[..., :c2], z[..., c2:]
    shift, raw = _coupling_cnn_any(z0, p, training, cnn_fp16, record)
    ls = p["rescaling_scale"] * np.tanh(raw)
    x1 = z1 * np.exp(ls) + shift
    return np.concatenate([z0, x1], axis=-1), ls.sum(axis=(1, 2, 3))


def affine_coupling_forward(x, p, training=False, cnn_fp16=False, record=None):
    """AffineCoupling._forward, layers.py:275-291 (sampling direction)."""
    c2 = x.shape[-1] // 2
    x0, x1 = x[..., :c2], x[..., c2:]
    shift, raw = _coupling_cnn_any(x0, p, training, cnn_fp16, record)
    ls = p["rescaling_scale"] * np.tanh(raw)
    y1 = (x1 - shift) * np.exp(-ls)
    return np.concatenate([x0, y1], axis=-1)


def conv1x1_inverse(z, A, log_abs_det):
    """Conv2d1x1._inverse_and_log_det_jacobian, layers.py:117-130,137-140: z @ A per pixel."""
    h, w = z.shape[1:3]
    ld = np.full((z.shape[0],), log_abs_det * (h * w), dtype=z.dtype)
    return z @ A, ld


def conv1x1_forward(x, A_inv):
    """Conv2d1x1._forward, layers.py:108-115: x @ A_inv per pixel."""
    return x @ A_inv


def sdn_ex5_scalars(p: Dict[str, np.ndarray], iso: float, cam: float, c_i: float = 1.0, dtype=np.float64):
    """Host scalars of sdn_model_params_ex5, cond_utils.py:205-239 → (beta1, beta2, gain)."""
    dt = dtype
    cam_vals = np.arange(5, dtype=np.float64)
    idx = np.where(cam_vals == float(cam))[0]
    if idx.size == 0:
        raise IndexError("unknown camera id %r" % (cam,))        # cam_idx[0] on an empty tensor, :216-217
    cp = np.exp(dt(c_i) * np.asarray(p["cam_params"], dt)[:, idx[0]])     # :218-220
    k = np.where(np.asarray(ISO_VALS) == float(iso))[0]
    g = np.asarray(p["gain_params"], dt)[k[0]] if k.size else dt(0)        # :227-229 (unknown ISO → 0)
    gain = np.exp(dt(c_i) * g * cp[2]) * dt(iso)                          # :230
    beta1 = np.exp(dt(c_i) * np.asarray(p["beta1"], dt).reshape(-1)[0] * cp[0])   # :236
    beta2 = np.exp(dt(c_i) * np.asarray(p["beta2"], dt).reshape(-1)[0] * cp[1])   # :237
    return dt(beta1), dt(beta2), dt(gain)


def sdn_ex5_scale(y, p, iso, cam):
    dt = y.dtype.type
    b1, b2, gain = sdn_ex5_scalars(p, iso, cam, dtype=dt)
    return np.sqrt(b1 * y / gain + b2)                                    # cond_utils.py:238


def _sigmoid(v):
    return 1.0 / (1.0 + np.exp(-v))


def sdn_ex4_scale(y, p, iso):
    """sdn_model_params_ex4, cond_utils.py:178-202 (c = 1, no camera parameters)."""
    dt = y.dtype.type
    k = np.where(np.asarray(ISO_VALS) == float(iso))[0]
    g = np.asarray(p["gain_params"], dt)[k[0]] if k.size else dt(0)
    gain = np.exp(g) * dt(iso)
    beta1 = np.exp(np.asarray(p["beta1"], dt).reshape(-1)[0])
    beta2 = np.exp(np.asarray(p["beta2"], dt).reshape(-1)[0])
    return np.sqrt(beta1 * y / gain + beta2)


def sdn_plain_scale(y, p):
    """sdn_model_params, cond_utils.py:41-52."""
    dt = y.dtype.type
    return np.sqrt(dt(_sigmoid(float(np.asarray(p["b1"]).reshape(-1)[0]))) * y + dt(_sigmoid(float(np.asarray(p["b2"]).reshape(-1)[0]))))


def gain_plain_scale(p, iso, dtype):
    """gain_model_params(iso), cond_utils.py:319-330 via AffineCouplingGain.py:52."""
    return dtype(_sigmoid(float(np.asarray(p["g1"]).reshape(-1)[0])) * float(iso) + _sigmoid(float(np.asarray(p["g2"]).reshape(-1)[0])))


ISO_TABLE = (100, 400, 800, 1600, 3200)


def _iso_entry(table, iso):
    """The nested tf.cond of the Ex1-Ex3 layers (cond_utils.py:69-88): the entry of ISO 100/400/800/1600/3200,
    any other ISO falls through to the ISO-800 entry."""
    iso = float(np.asarray(iso).reshape(-1)[0])
    k = ISO_TABLE.index(int(iso)) if iso in ISO_TABLE else 2
    return np.asarray(table).reshape(-1)[k]


def sdn_ex123_scale(y, p, iso, kind):
    """sdn_model_params_ex1 / _ex2 / _ex3 (cond_utils.py:55-175)."""
    dt = y.dtype.type
    iso_v = dt(float(np.asarray(iso).reshape(-1)[0]))
    c = dt(1e-2) if kind == "sdn1" else dt(1e-1)
    gain = np.exp(c * dt(_iso_entry(p["table"], iso))) * iso_v
    b1 = dt(_sigmoid(float(np.asarray(p["b1"]).reshape(-1)[0])))
    b2 = dt(_sigmoid(float(np.asarray(p["b2"]).reshape(-1)[0])))
    if kind == "sdn1":
        return np.sqrt(b1 * y / gain + b2)
    if kind == "sdn2":
        return np.sqrt(gain * (b1 * y / gain + b2))
    return gain * np.sqrt(b1 * y / gain + b2)


def sdn_ex6_scale(y, p, iso, cam, c_i=1.0):
    """sdn_model_params_ex6 (cond_utils.py:242-276): ONE camera parameter, on the gain exponent only."""
    dt = y.dtype.type
    cam = float(np.asarray(cam).reshape(-1)[0])
    iso_f = float(np.asarray(iso).reshape(-1)[0])
    if cam not in (0.0, 1.0, 2.0, 3.0, 4.0):
        raise ValueError("unknown camera id %r" % cam)
    cp = np.exp(dt(c_i) * dt(np.asarray(p["cam_params"]).reshape(-1)[int(cam)]))
    g = dt(np.asarray(p["gain_params"]).reshape(-1)[ISO_TABLE.index(int(iso_f))]) if iso_f in ISO_TABLE else dt(0.0)
    gain = np.exp(dt(c_i) * g * cp) * dt(iso_f)
    beta1 = np.exp(dt(c_i) * dt(np.asarray(p["beta1"]).reshape(-1)[0]))
    beta2 = np.exp(dt(c_i) * dt(np.asarray(p["beta2"]).reshape(-1)[0]))
    return np.sqrt(beta1 * y / gain + beta2)


def gain_ex123_scale(p, iso, kind, dtype):
    """gain_model_params_ex1 / _ex2 / _ex3 (cond_utils.py:333-429); the layers feed gain = iso."""
    iso_f = float(np.asarray(iso).reshape(-1)[0])
    if kind == "gain1":
        return dtype(np.exp(1e-5 * float(np.asarray(p["g1"]).reshape(-1)[0])) * iso_f + np.exp(1e-5 * float(np.asarray(p["g2"]).reshape(-1)[0])))
    if kind == "gain2":
        return dtype(np.exp(1e-1 * float(_iso_entry(p["table"], iso))) * iso_f)
    return dtype(np.exp(1e-5 * float(_iso_entry(p["table"], iso))))


def sdn_ex5_inverse(x, y, p, iso, cam):
    """AffineCouplingSdnEx5._inverse_and_log_det_jacobian, AffineCouplingSdnEx5.py:118-132."""
    scale = sdn_ex5_scale(y, p, iso, cam)
    return x / scale, -np.log(scale).sum(axis=(1, 2, 3))


def sdn_ex5_forward(z, y, p, iso, cam):
    """AffineCouplingSdnEx5._forward, AffineCouplingSdnEx5.py:50-66."""
    return z * sdn_ex5_scale(y, p, iso, cam)


def gain_ex4_inverse(z, gain_val):
    """AffineCouplingGainEx4._inverse_and_log_det_jacobian, AffineCouplingGainEx4.py:114-127."""
    dt = z.dtype.type
    g = dt(np.asarray(gain_val).reshape(-1)[0])
    n = z.shape[1] * z.shape[2] * z.shape[3]
    return z / g, np.full((z.shape[0],), -n * np.log(g), dtype=z.dtype)


def gain_ex4_forward(x, gain_val):
    """AffineCouplingGainEx4._forward, AffineCouplingGainEx4.py:49-65."""
    return x * x.dtype.type(np.asarray(gain_val).reshape(-1)[0])


def prior_logp(z):
    """gaussian_diag.logp with mean = logsd = 0, noise_flow_model.py:486-497,537-539."""
    dt = z.dtype.type
    return (dt(-0.5) * (dt(np.log(2 * np.pi)) + z * z)).sum(axis=(1, 2, 3))


# ----------------------------------------------------------------------------
# variable naming (checkpoint names, Appendix B of SURVEY.md) and binding
# ----------------------------------------------------------------------------
def parse_arch(arch: str) -> List[Tuple[str, int]]:
    """noise_flow_model.py:71-235: 'a|b|c' → [(layer_type, i)], i = position in arch."""
    out = []
    for i, lyr in enumerate(arch.split("|")):
        if lyr not in ("unc", "sdn", "sdn1", "sdn2", "sdn3", "sdn4", "sdn5", "sdn6", "gain", "gain1", "gain2", "gain3", "gain4"):
            raise ValueError("oracle supports unc, sdn[1-6], gain[1-4] only, got %r" % lyr)
        out.append((lyr, i))
    return out


def layer_names(arch: str) -> List[str]:
    """NoiseFlow.get_layer_names, noise_flow_model.py:508-513 (= hps.txt:1-18)."""
    names = []
    for lyr, i in parse_arch(arch):
        if lyr == "unc":
            names += ["Conv2d_1x1_%d" % i, "unc_%d" % i]
        elif lyr.startswith("sdn"):
            names.append("sdn_%d" % i)
        else:
            names.append("gain_%d" % i)
    return names


def template_name(k: int) -> str:
    return "model/real_nvp_conv_template" + ("" if k == 0 else "_%d" % k)


def bind_variables(arch: str, variables: Dict[str, np.ndarray], binding: str = "loss_first", dtype=np.float64,
                   flow_permutation: int = 1, decomp: str = "LU"):
    """Attach checkpoint variables to layers.

    ``binding``: tf.make_template scopes are numbered in the
    order the coupling CNNs are FIRST CALLED — NLL order when the loss graph is
    built first (train_noise_flow.py:302), reversed when only the sampling graph
    is built (NoiseFlowWrapper.py:64, noise_flow_model.py:435).
    """
    if binding not in ("loss_first", "sample_first"):
        raise ValueError("binding must be loss_first or sample_first")
    arch_l = parse_arch(arch)
    unc_ids = [i for lyr, i in arch_l if lyr == "unc"]
    order = unc_ids if binding == "loss_first" else unc_ids[::-1]
    tmpl_of = {i: k for k, i in enumerate(order)}
    f = lambda a: np.asarray(a, dtype)
    layers = []
    for lyr, i in arch_l:
        if lyr == "unc":
            if flow_permutation == 1:      # noise_flow_model.py:85-90
                A, A_inv, lad = conv1x1_from_variables(variables, i, decomp, dtype)
                layers.append({"type": "conv1x1", "name": "Conv2d_1x1_%d" % i, "A": A, "A_inv": A_inv, "log_abs_det": lad})
            elif flow_permutation == 0:    # noise_flow_model.py:80-84: tfb.Permute(channels reversed), log|det| = 0
                J = np.eye(4, dtype=dtype)[::-1].copy()
                layers.append({"type": "conv1x1", "name": "permute", "A": J, "A_inv": J, "log_abs_det": dtype(0.0)})
            # any other value: "No permutation specified. Not using any." (noise_flow_model.py:91-92)
            t = template_name(tmpl_of[i]) + "/"
            p = {
                "l_1/W": f(variables[t + "l_1/W"]), "l_1/b": f(variables[t + "l_1/b"]).reshape(-1),
                "bn1/mean": f(variables[t + "bn_nvp_conv_1/mean"]), "bn1/var": f(variables[t + "bn_nvp_conv_1/var"]),
                "l_2/W": f(variables[t + "l_2/W"]), "l_2/b": f(variables[t + "l_2/b"]).reshape(-1),
                "bn2/mean": f(variables[t + "bn_nvp_conv_2/mean"]), "bn2/var": f(variables[t + "bn_nvp_conv_2/var"]),
                "l_last/W": f(variables[t + "l_last/W"]), "l_last/b": f(variables[t + "l_last/b"]).reshape(-1),
                "l_last/logs": f(variables[t + "l_last/logs"]).reshape(-1),
                "rescaling_scale": dtype(variables["level0/bijector%d/rescaling_scale0" % i]),
            }
            layers.append({"type": "coupling", "name": "unc_%d" % i, "p": p})
        elif lyr == "sdn5":
            p = {k: f(variables["model/sdn_gain/" + k]) for k in ("beta1", "beta2", "gain_params", "cam_params")}
            layers.append({"type": "sdn5", "name": "sdn_%d" % i, "p": p})
        elif lyr == "sdn4":
            p = {k: f(variables["model/sdn_gain/" + k]) for k in ("beta1", "beta2", "gain_params")}
            layers.append({"type": "sdn4", "name": "sdn_%d" % i, "p": p})
        elif lyr == "sdn":
            layers.append({"type": "sdn", "name": "sdn_%d" % i, "p": {"b1": f(variables["model/b1"]), "b2": f(variables["model/b2"])}})
        elif lyr == "gain":
            layers.append({"type": "gain", "name": "gain_%d" % i, "p": {"g1": f(variables["model/g1"]), "g2": f(variables["model/g2"])}})
        elif lyr in ("sdn1", "sdn2", "sdn3"):
            tab = "model/r_gain_param_%05d" if lyr == "sdn1" else "model/gain_param_%05d"
            p = {"b1": f(variables["model/b1"]), "b2": f(variables["model/b2"]),
                 "table": np.concatenate([f(variables[tab % iso]).reshape(-1) for iso in ISO_TABLE])}
            layers.append({"type": lyr, "name": "sdn_%d" % i, "p": p})
        elif lyr == "sdn6":
            p = {k: f(variables["model/sdn_gain/" + k]) for k in ("beta1", "beta2", "gain_params", "cam_params")}
            layers.append({"type": "sdn6", "name": "sdn_%d" % i, "p": p})
        elif lyr == "gain1":
            layers.append({"type": "gain1", "name": "gain_%d" % i, "p": {"g1": f(variables["model/g1"]), "g2": f(variables["model/g2"])}})
        elif lyr in ("gain2", "gain3"):
            p = {"table": np.concatenate([f(variables["model/gain_param_%05d" % iso]).reshape(-1) for iso in ISO_TABLE])}
            layers.append({"type": lyr, "name": "gain_%d" % i, "p": p})
        else:
            layers.append({"type": "gain4", "name": "gain_%d" % i, "gain_val": f(variables["model/sdn_gain/gain_val"])})
    return layers


def fresh_variables(arch: str, width: int = 4, channels: int = 4, seed: int = 0, flow_permutation: int = 1,
                    decomp: str = "LU") -> Dict[str, np.ndarray]:
    """Fresh-init variables under the reference's names and initialisers:
    QR-orthogonal 1x1 matrix (layers.py:95) decomposed by scipy LU
    (matrix_param.py:100-123); l_1/l_2 ~ N(0, (width/512*0.05)^2), biases 0
    (layers.py:598-609); l_last W=b=logs=0 (layers.py:662-673); BN mean 0 / var 1
    (layers.py:382-387); rescaling_scale 1e-4 (layers.py:271-273); sdn/gain
    parameters from train_noise_flow.py:201-214 and cond_utils.py:438."""
    import scipy.linalg as sla
    rng = np.random.RandomState(seed)
    v: Dict[str, np.ndarray] = {}
    c2 = channels // 2
    k = 0
    for lyr, i in parse_arch(arch):
        v["level0/bijector%d/rescaling_scale0" % i] = np.float32(1e-4)
        if lyr == "unc":
            if flow_permutation == 1:
                q = sla.qr(rng.randn(channels, channels))[0].astype(np.float32)
                v.update(conv1x1_init_variables(q, i, decomp))
            t = template_name(k) + "/"
            k += 1
            std = width / 512 * 0.05
            v[t + "l_1/W"] = (rng.randn(3, 3, c2, width) * std).astype(np.float32)
            v[t + "l_1/b"] = np.zeros((1, 1, 1, width), np.float32)
            v[t + "l_2/W"] = (rng.randn(1, 1, width, width) * std).astype(np.float32)
            v[t + "l_2/b"] = np.zeros((1, 1, 1, width), np.float32)
            v[t + "l_last/W"] = np.zeros((3, 3, width + 1, 2 * c2), np.float32)
            v[t + "l_last/b"] = np.zeros((1, 1, 1, 2 * c2), np.float32)
            v[t + "l_last/logs"] = np.zeros((1, 2 * c2), np.float32)
            for b in ("bn_nvp_conv_1", "bn_nvp_conv_2"):
                v[t + b + "/mean"] = np.zeros((width,), np.float32)
                v[t + b + "/var"] = np.ones((width,), np.float32)
    if any(l == "sdn" for l, _ in parse_arch(arch)):
        v["model/b1"] = np.full((1,), -3.0, np.float32)
        v["model/b2"] = np.full((1,), 3.0, np.float32)
    if any(l == "gain" for l, _ in parse_arch(arch)):
        v["model/g1"] = np.full((1,), -3.0, np.float32)
        v["model/g2"] = np.full((1,), 3.0, np.float32)
    kinds = {l for l, _ in parse_arch(arch)}
    if kinds & {"sdn1", "sdn2", "sdn3"}:                       # cond_utils.py:90-93
        v["model/b1"] = np.full((1,), -3.0, np.float32)
        v["model/b2"] = np.full((1,), 3.0, np.float32)
    if "sdn1" in kinds:                                        # cond_utils.py:60-68
        for iso in ISO_TABLE:
            v["model/r_gain_param_%05d" % iso] = np.zeros((1,), np.float32)
    if kinds & {"sdn2", "sdn3", "gain2"}:                      # cond_utils.py:102-112, 361-366 (hps.gain_init = -5)
        for iso in ISO_TABLE:
            v["model/gain_param_%05d" % iso] = np.full((1,), -5.0 / 1e-1, np.float32)
    elif "gain3" in kinds:                                     # cond_utils.py:401-405
        for iso in ISO_TABLE:
            v["model/gain_param_%05d" % iso] = np.full((1,), -5.0 / 1e-5, np.float32)
    if "gain1" in kinds:                                       # cond_utils.py:341-342
        v["model/g1"] = np.full((1,), -5.0 / 1e-5, np.float32)
        v["model/g2"] = np.zeros((1,), np.float32)
    if kinds & {"sdn5", "gain4", "sdn4", "sdn6"}:
        v["model/sdn_gain/beta1"] = np.full((1,), -5.0, np.float32)
        v["model/sdn_gain/beta2"] = np.zeros((1,), np.float32)
        v["model/sdn_gain/gain_params"] = np.full((5,), -5.0, np.float32)
        v["model/sdn_gain/cam_params"] = np.ones((1, 5) if "sdn6" in kinds else (3, 5), np.float32)   # cond_utils.py:254 / :219
        v["model/sdn_gain/gain_val"] = np.ones((1,), np.float32)
    return v


def count_trainable(variables: Dict[str, np.ndarray]) -> int:
    """num_params as written to hps.txt (train_noise_flow.py:309-312): everything
    except the LU permutation/sign and the BN running statistics."""
    n = 0
    for name, arr in variables.items():
        if "/P_matpar" in name or "/sign_S_matpar" in name or name.endswith("/mean") or name.endswith("/var"):
            continue
        n += int(np.asarray(arr).size)
    return n


# ----------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------
class NoiseFlowOracle:
    """NoiseFlow (noise_flow_model.py:44-513) restated on numpy, eval-mode BN by
    default.  ``x`` = noise, ``y`` = clean image, ``z`` = latent."""

    def __init__(self, arch: str, variables: Dict[str, np.ndarray], binding: str = "loss_first",
                 dtype=np.float64, sidd_cond: str = "mix", cnn_dtype: str = "fp32", flow_permutation: int = 1,
                 decomp: str = "LU"):
        self.arch = arch
        self.dtype = dtype
        # 'fp16': emulate the library's fp16 coupling-CNN mode;  'fp16_plain': the independent, unfolded fp16 evaluation
        self.cnn_fp16 = "plain" if cnn_dtype == "fp16_plain" else cnn_dtype == "fp16"
        self.sidd_cond = sidd_cond
        self.layers = bind_variables(arch, variables, binding, dtype, flow_permutation, decomp)

    def _record(self, L, training):
        """Training mode: ``last_batch_moments[layer name]`` ← the moments of the latest call."""
        if not training:
            return None
        if not hasattr(self, "last_batch_moments"):
            self.last_batch_moments = {}
        return self.last_batch_moments.setdefault(L["name"], {})

    # -- NLL direction: NoiseFlow.inverse, noise_flow_model.py:394-428 --------
    def inverse(self, x, y=None, iso=None, cam=None, training=False, return_layers=False):
        dt = self.dtype
        z = np.asarray(x, dt)
        y = None if y is None else np.asarray(y, dt)
        obj = np.zeros((z.shape[0],), dt)
        per_layer = []
        for L in self.layers:
            if L["type"] == "conv1x1":
                z, ld = conv1x1_inverse(z, L["A"], L["log_abs_det"])
            elif L["type"] == "coupling":
                z, ld = affine_coupling_inverse(z, L["p"], training, self.cnn_fp16, self._record(L, training))
            elif L["type"] == "sdn5":
                z, ld = sdn_ex5_inverse(z, y, L["p"], iso, cam)
            elif L["type"] in ("sdn4", "sdn"):
                scale = sdn_ex4_scale(y, L["p"], iso) if L["type"] == "sdn4" else sdn_plain_scale(y, L["p"])
                z, ld = z / scale, -np.log(scale).sum(axis=(1, 2, 3))
            elif L["type"] in ("sdn1", "sdn2", "sdn3", "sdn6"):
                scale = sdn_ex6_scale(y, L["p"], iso, cam) if L["type"] == "sdn6" else sdn_ex123_scale(y, L["p"], iso, L["type"])
                z, ld = z / scale, -np.log(scale).sum(axis=(1, 2, 3))
            elif L["type"] in ("gain", "gain1", "gain3"):
                # AffineCouplingGain.py:113-127 (and GainEx1 / GainEx3): log|det| = -log(scale), a [1]-tensor
                # broadcast over the batch — the reference omits the H*W*C factor; restated as written
                g = gain_plain_scale(L["p"], iso, dt) if L["type"] == "gain" else gain_ex123_scale(L["p"], iso, L["type"], dt)
                z, ld = z / g, np.full((z.shape[0],), -np.log(g), dtype=z.dtype)
            elif L["type"] == "gain2":
                # AffineCouplingGainEx2.py:112-126: scale += y*0 broadcasts first, so the sum runs over H*W*C
                g = gain_ex123_scale(L["p"], iso, "gain2", dt)
                z, ld = z / g, np.full((z.shape[0],), -z[0].size * np.log(g), dtype=z.dtype)
            else:
                z, ld = gain_ex4_inverse(z, L["gain_val"])
            obj = obj + ld
            if return_layers:
                per_layer.append((L["name"], z.copy(), ld.copy()))
        if return_layers:
            return z, obj, per_layer
        return z, obj

    # -- NoiseFlow._loss / loss, noise_flow_model.py:458-484 ------------------
    def nll(self, x, y=None, iso=None, cam=None, training=False):
        """→ (nll[B], sd_z scalar, z)."""
        z, obj = self.inverse(x, y, iso, cam, training)
        obj = obj + prior_logp(z)
        var = z.var(axis=(1, 2, 3))                   # tf.nn.moments → population variance, :477
        return -obj, np.sqrt(var).mean(), z

    def loss(self, x, y=None, iso=None, cam=None, training=False):
        nll, sd_z, _ = self.nll(x, y, iso, cam, training)
        return nll.mean(), sd_z

    # -- sampling direction: NoiseFlow.forward/sample, noise_flow_model.py:430-456
    def forward(self, z, y=None, iso=None, cam=None, training=False):
        dt = self.dtype
        x = np.asarray(z, dt)
        y = None if y is None else np.asarray(y, dt)
        for L in reversed(self.layers):
            if L["type"] == "conv1x1":
                x = conv1x1_forward(x, L["A_inv"])
            elif L["type"] == "coupling":
                x = affine_coupling_forward(x, L["p"], training, self.cnn_fp16, self._record(L, training))
            elif L["type"] == "sdn5":
                x = sdn_ex5_forward(x, y, L["p"], iso, cam)
            elif L["type"] == "sdn4":
                x = x * sdn_ex4_scale(y, L["p"], iso)
            elif L["type"] == "sdn":
                x = x * sdn_plain_scale(y, L["p"])
            elif L["type"] in ("sdn1", "sdn2", "sdn3"):
                x = x * sdn_ex123_scale(y, L["p"], iso, L["type"])
            elif L["type"] == "sdn6":
                x = x * sdn_ex6_scale(y, L["p"], iso, cam)
            elif L["type"] == "gain":
                x = x * gain_plain_scale(L["p"], iso, dt)
            elif L["type"] in ("gain1", "gain2", "gain3"):
                x = x * gain_ex123_scale(L["p"], iso, L["type"], dt)
            else:
                x = gain_ex4_forward(x, L["gain_val"])
        return x

    def sample(self, eps, temp, y=None, iso=None, cam=None, training=False):
        """prior.sample(eps_std) = eps * temp (noise_flow_model.py:499-504), then forward."""
        return self.forward(np.asarray(eps, self.dtype) * self.dtype(temp), y, iso, cam, training)


# ----------------------------------------------------------------------------
# closed-form baselines (sidd/PatchStatsCalculator.py:104-115)
# ----------------------------------------------------------------------------
def nll_gauss(x, sd):
    """Per-patch NLL of x under N(0, sd^2) i.i.d."""
    x = np.asarray(x, np.float64)
    n = x[0].size
    return 0.5 * n * np.log(2 * np.pi * sd * sd) + 0.5 * (x * x).sum(axis=(1, 2, 3)) / (sd * sd)


def nll_sdn(x, y, b1, b2):
    """Per-patch NLL of x under the camera NLF N(0, b1*y + b2)."""
    x = np.asarray(x, np.float64)
    var = b1 * np.asarray(y, np.float64) + b2
    return 0.5 * (np.log(2 * np.pi * var) + x * x / var).sum(axis=(1, 2, 3))
