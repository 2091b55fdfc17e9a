"""Architecture string → layer list → flat raw parameter block of the C ABI.

Host-side mirror of ``NoiseFlow.noise_flow_arch`` (reference
``borealisflows/noise_flow_model.py:71-235``), of the variable naming the
reference's checkpoints use and of the reference
initialisers.  The *folding* of these raw parameters (PLU → A, BN-eval, edge
channel, exp(3·logs)) happens inside the HIP library (``csrc/nf_host.hip``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple, Optional

import numpy as np

from . import _lib

SUPPORTED_LAYERS = ("unc", "sdn", "sdn1", "sdn2", "sdn3", "sdn4", "sdn5", "sdn6", "gain", "gain1", "gain2", "gain3", "gain4")
ISO_TABLE = (100, 400, 800, 1600, 3200)   # per-ISO variables of the Ex1-Ex3 layers (cond_utils.py:62-68)
_SCALAR_KINDS = {   # arch key -> (NF_LAYER_*, display prefix)   noise_flow_model.py:106-223
    "sdn1": ("NF_LAYER_SDN1", "sdn"), "sdn2": ("NF_LAYER_SDN2", "sdn"), "sdn3": ("NF_LAYER_SDN3", "sdn"),
    "sdn6": ("NF_LAYER_SDN6", "sdn"), "gain1": ("NF_LAYER_GAIN1", "gain"), "gain2": ("NF_LAYER_GAIN2", "gain"),
    "gain3": ("NF_LAYER_GAIN3", "gain"),
}
C_I = 1.0   # train_noise_flow.py:207 / NoiseFlowWrapper.py:125


@dataclass
class LayerSpec:
    kind: str        # 'conv1x1' | 'coupling' | 'sdn5' | 'gain4' | 'sdn4' | 'sdn' | 'gain'
    name: str        # display name as in hps.txt:1-18 (get_layer_names)
    arch_index: int  # position i in arch.split('|')
    nf_type: int     # NF_LAYER_*
    width: int = 0


_CONV1X1_KINDS = {"LU": ("conv1x1", "NF_LAYER_CONV1X1"), "LU2": ("conv1x1_lu2", "NF_LAYER_CONV1X1_LU2"),
                  "NONE": ("conv1x1_none", "NF_LAYER_CONV1X1_NONE")}     # matrix_param.py:191-193


def parse_arch(arch: str, flow_permutation: int = 1, decomp: str = "LU") -> List[LayerSpec]:
    """noise_flow_model.py:71-235.  ``flow_permutation`` = 1 (shipped): a Conv2d1x1 parameterised by ``decomp`` before
    every AffineCoupling (:85-90); 0: tfb.Permute reversing the channels (:80-84); anything else: no mixing layer
    (:91-92)."""
    if decomp not in _CONV1X1_KINDS:
        raise ValueError("hps.decomp must be one of %s (matrix_param.py:191-193), got %r" % (sorted(_CONV1X1_KINDS), decomp))
    if not arch:
        raise ValueError("hps.arch must be a non-empty 'a|b|c' string (revnet2d stacks are out of scope)")
    layers: List[LayerSpec] = []
    for i, lyr in enumerate(arch.split("|")):
        if lyr == "unc":
            if int(flow_permutation) == 1:
                kind, typ = _CONV1X1_KINDS[decomp]
                layers.append(LayerSpec(kind, "Conv2d_1x1_%d" % i, i, getattr(_lib, typ)))
            elif int(flow_permutation) == 0:
                layers.append(LayerSpec("permute", "permute", i, _lib.NF_LAYER_PERMUTE))
            layers.append(LayerSpec("coupling", "unc_%d" % i, i, _lib.NF_LAYER_COUPLING))
        elif lyr == "sdn5":
            layers.append(LayerSpec("sdn5", "sdn_%d" % i, i, _lib.NF_LAYER_SDN5))
        elif lyr == "gain4":
            layers.append(LayerSpec("gain4", "gain_%d" % i, i, _lib.NF_LAYER_GAIN4))
        elif lyr == "sdn4":      # noise_flow_model.py:148-158 (job_noise_flow.sh: "sdn4|gain4")
            layers.append(LayerSpec("sdn4", "sdn_%d" % i, i, _lib.NF_LAYER_SDN4))
        elif lyr == "sdn":       # noise_flow_model.py:106-114
            layers.append(LayerSpec("sdn", "sdn_%d" % i, i, _lib.NF_LAYER_SDN))
        elif lyr == "gain":      # noise_flow_model.py:183-192
            layers.append(LayerSpec("gain", "gain_%d" % i, i, _lib.NF_LAYER_GAIN))
        elif lyr in _SCALAR_KINDS:   # noise_flow_model.py:116-147, 171-182, 193-223
            typ, prefix = _SCALAR_KINDS[lyr]
            layers.append(LayerSpec(lyr, "%s_%d" % (prefix, i), i, getattr(_lib, typ)))
        else:
            raise NotImplementedError(
                "arch layer %r is not on the MI355X hot path (supported: %s)" % (lyr, "|".join(SUPPORTED_LAYERS)))
    return layers


def template_scope(k: int) -> str:
    """tf.make_template scope of the k-th coupling CNN created (layers.py:449,498)."""
    return "model/real_nvp_conv_template" + ("" if k == 0 else "_%d" % k)


def template_binding(layers: List[LayerSpec], binding: str) -> Dict[int, int]:
    """arch_index of each ``unc`` → template number.

    Template scopes are numbered in the order the coupling CNNs are first CALLED
    (SURVEY.md quirk Q1): NLL order when the loss graph is built first
    (``train_noise_flow.py:302``) — ``loss_first`` — and reversed when only the
    sampling graph exists (``NoiseFlowWrapper.py:64``) — ``sample_first``.
    """
    if binding not in ("loss_first", "sample_first"):
        raise ValueError("binding must be 'loss_first' or 'sample_first'")
    ids = [l.arch_index for l in layers if l.kind == "coupling"]
    if binding == "sample_first":
        ids = ids[::-1]
    return {i: k for k, i in enumerate(ids)}


def conv1x1_names(i: int, decomp: str = "LU") -> Dict[str, str]:
    """Variable names per decomposition (matrix_param.py:24, :109-123, :151-161), under the layer's scope."""
    pre = "level0/bijector%d/Conv2d_1x1_%d/" % (i, i)
    nm = "conv2d_1x1_%d_0" % i
    if decomp == "LU2":
        return {"P": pre + "P_" + nm, "L": pre + "L_filters_" + nm, "sign_S": pre + "sign_S_" + nm,
                "log_S": pre + "log_S_filters_" + nm, "U": pre + "U_filters_" + nm}
    if decomp == "NONE":
        return {"A": pre + "A_matpar_none_" + nm}
    return {k: pre + k + "_matpar_lu_" + nm for k in ("P", "sign_S", "log_S", "L_vec", "U_vec")}


def _f32(a) -> np.ndarray:
    return np.asarray(a, dtype=np.float32).reshape(-1)


def pack(arch: str, variables: Dict[str, np.ndarray], width: int, binding: str = "loss_first", flow_permutation: int = 1,
         decomp: str = "LU"):
    """→ (layers, descs ctypes array, params float32 ndarray) in the canonical raw
    layout documented in include/noiseflow_hip.h."""
    layers = parse_arch(arch, flow_permutation, decomp)
    return pack_layers(layers, variables, width, template_binding(layers, binding))


def layer_variable_names(L: LayerSpec, tmpl: Dict[int, int]) -> List[Optional[str]]:
    """Checkpoint names of one layer's variables in the order of the raw layout
    (include/noiseflow_hip.h); ``None`` marks a constant that is not a variable (sdn5's c_i)."""
    if L.kind == "conv1x1":
        n = conv1x1_names(L.arch_index)
        return [n["P"], n["sign_S"], n["log_S"], n["L_vec"], n["U_vec"]]
    if L.kind == "conv1x1_lu2":
        n = conv1x1_names(L.arch_index, "LU2")
        return [n["P"], n["L"], n["sign_S"], n["log_S"], n["U"]]
    if L.kind == "conv1x1_none":
        return [conv1x1_names(L.arch_index, "NONE")["A"]]
    if L.kind == "permute":
        return []
    if L.kind == "coupling":
        t = template_scope(tmpl[L.arch_index]) + "/"
        return [t + "l_1/W", t + "l_1/b", t + "bn_nvp_conv_1/mean", t + "bn_nvp_conv_1/var",
                t + "l_2/W", t + "l_2/b", t + "bn_nvp_conv_2/mean", t + "bn_nvp_conv_2/var",
                t + "l_last/W", t + "l_last/b", t + "l_last/logs",
                "level0/bijector%d/rescaling_scale0" % L.arch_index]
    if L.kind == "sdn5":
        return ["model/sdn_gain/beta1", "model/sdn_gain/beta2", "model/sdn_gain/gain_params",
                "model/sdn_gain/cam_params", None]
    if L.kind == "sdn4":   # variables of sdn_model_params_ex4 (cond_utils.py:178-202), scope 'sdn_gain'
        return ["model/sdn_gain/beta1", "model/sdn_gain/beta2", "model/sdn_gain/gain_params"]
    if L.kind == "sdn":    # sdn_model_params (cond_utils.py:41-52): created under the 'model' scope
        return ["model/b1", "model/b2"]
    if L.kind == "gain":   # gain_model_params (cond_utils.py:319-330)
        return ["model/g1", "model/g2"]
    if L.kind == "sdn1":   # sdn_model_params_ex1 (cond_utils.py:55-98)
        return ["model/b1", "model/b2"] + ["model/r_gain_param_%05d" % iso for iso in ISO_TABLE]
    if L.kind in ("sdn2", "sdn3"):   # sdn_model_params_ex2 / _ex3 (cond_utils.py:101-175)
        return ["model/b1", "model/b2"] + ["model/gain_param_%05d" % iso for iso in ISO_TABLE]
    if L.kind == "sdn6":   # sdn_model_params_ex6 (cond_utils.py:242-276): cam_params is [1, 5]
        return ["model/sdn_gain/beta1", "model/sdn_gain/beta2", "model/sdn_gain/gain_params",
                "model/sdn_gain/cam_params", None]
    if L.kind == "gain1":  # gain_model_params_ex1 (cond_utils.py:333-350)
        return ["model/g1", "model/g2"]
    if L.kind in ("gain2", "gain3"):   # gain_model_params_ex2 / _ex3 (cond_utils.py:353-429)
        return ["model/gain_param_%05d" % iso for iso in ISO_TABLE]
    return ["model/sdn_gain/gain_val"]   # gain4


def pack_layers(layers: List[LayerSpec], variables: Dict[str, np.ndarray], width: int, tmpl: Dict[int, int]):
    """Pack an explicit layer list (e.g. ONE bijector of a larger architecture);
    ``tmpl`` maps the arch index of each coupling to its template scope number."""
    chunks: List[np.ndarray] = []
    offsets: List[int] = []
    pos = 0

    def need(name: str) -> np.ndarray:
        if name not in variables:
            raise KeyError("checkpoint variable %r not found" % name)
        return variables[name]

    for L in layers:
        if L.kind == "coupling":
            L.width = int(width)
            t = template_scope(tmpl[L.arch_index]) + "/"
            w1 = np.asarray(need(t + "l_1/W"), np.float32)
            if w1.shape != (3, 3, 2, width):
                raise ValueError("%sl_1/W has shape %s, expected (3,3,2,%d)" % (t, w1.shape, width))
        blk = np.concatenate([np.zeros((0,), np.float32)] +
                             [_f32(need(nm)) if nm is not None else np.asarray([C_I], np.float32)
                              for nm in layer_variable_names(L, tmpl)])
        expect = _lib.load().nf_layer_param_count(L.nf_type, L.width)
        if blk.size != expect:
            raise ValueError("layer %s: %d raw parameters, C ABI expects %d" % (L.name, blk.size, expect))
        offsets.append(pos)
        chunks.append(blk)
        pos += blk.size
    params = np.ascontiguousarray(np.concatenate(chunks), dtype=np.float32)
    descs = (_lib.nf_layer_desc * len(layers))()
    for d, L, off in zip(descs, layers, offsets):
        d.type, d.width, d.param_offset = L.nf_type, L.width, off
    return layers, descs, params


def unpack_layers(layers: List[LayerSpec], flat: np.ndarray, variables: Dict[str, np.ndarray], tmpl: Dict[int, int]):
    """Inverse of :func:`pack_layers`: a copy of ``variables`` with every variable the layers own
    replaced by its slice of the raw vector ``flat`` (shapes and dtypes of ``variables`` kept)."""
    out = dict(variables)
    pos = 0
    flat = np.asarray(flat, np.float32).reshape(-1)
    for L in layers:
        for nm in layer_variable_names(L, tmpl):
            if nm is None:
                pos += 1
                continue
            ref = np.asarray(variables[nm])
            n = int(ref.size)
            out[nm] = flat[pos:pos + n].reshape(ref.shape).astype(ref.dtype if ref.dtype.kind == "f" else np.float32)
            pos += n
    if pos != flat.size:
        raise ValueError("raw vector has %d floats, the layers own %d" % (flat.size, pos))
    return out


# ----------------------------------------------------------------------------
# fresh initialisation (what tf.global_variables_initializer would produce)
# ----------------------------------------------------------------------------
def _fill_triangular_positions(n: int, upper: bool) -> np.ndarray:
    """Index map of tfdist.fill_triangular: entry (i,j) holds k+1 where vector
    element k lands (0 = outside the triangle)."""
    m = n * (n + 1) // 2
    v = np.arange(1, m + 1)
    if upper:
        return np.triu(np.concatenate([v, v[n:][::-1]]).reshape(n, n))
    return np.tril(np.concatenate([v[n:], v[::-1]]).reshape(n, n))


def stricttri2vec(mat: np.ndarray, upper: bool) -> np.ndarray:
    """matrix_param.py:59-97."""
    trim = mat[:-1, 1:] if upper else mat[1:, :-1]
    n = trim.shape[0]
    pos = _fill_triangular_positions(n, upper)
    out = np.zeros(n * (n + 1) // 2, dtype=mat.dtype)
    for i in range(n):
        for j in range(n):
            if pos[i, j]:
                out[pos[i, j] - 1] = trim[i, j]
    return out


def init_variables(arch: str, width: int = 4, channels: int = 4, seed: int = 0, flow_permutation: int = 1,
                   decomp: str = "LU", gain_init: float = -5.0) -> Dict[str, np.ndarray]:
    """Fresh variables under the reference's names with the reference's
    initialisers: QR-orthogonal 1x1 matrix → scipy LU (layers.py:95,
    matrix_param.py:100-123); l_1/l_2 ~ N(0, (width/512·0.05)²), zero biases
    (layers.py:598-609); zero l_last W/b/logs (layers.py:662-673); BN mean 0,
    var 1 (layers.py:382-387); rescaling_scale 1e-4 (layers.py:271-273);
    sdn/gain parameters of train_noise_flow.py:201-214 and cond_utils.py:438."""
    import scipy.linalg as sla
    rng = np.random.RandomState(seed)
    layers = parse_arch(arch, flow_permutation, decomp)
    v: Dict[str, np.ndarray] = {}
    c2 = channels // 2
    k = 0
    for L in layers:
        if not L.kind.startswith("conv1x1") and L.kind != "permute":
            v["level0/bijector%d/rescaling_scale0" % L.arch_index] = np.float32(1e-4)
        if L.kind in ("conv1x1_none", "conv1x1_lu2"):
            q = sla.qr(rng.randn(channels, channels))[0].astype(np.float32)
            if L.kind == "conv1x1_none":
                v[conv1x1_names(L.arch_index, "NONE")["A"]] = q
            else:                                                        # matrix_param.py:145-161
                p, l, u = sla.lu(q)
                s = np.diag(u)
                n = conv1x1_names(L.arch_index, "LU2")
                v[n["P"]] = p.astype(np.float32)
                v[n["L"]] = l.astype(np.float32)
                v[n["sign_S"]] = np.sign(s).astype(np.float32)
                v[n["log_S"]] = np.log(np.abs(s)).astype(np.float32)
                v[n["U"]] = np.triu(u, 1).astype(np.float32)
        elif L.kind == "conv1x1":
            q = sla.qr(rng.randn(channels, channels))[0].astype(np.float32)
            p, l, u = sla.lu(q)
            s = np.diag(u)
            n = conv1x1_names(L.arch_index)
            v[n["P"]] = p.astype(np.float32)
            v[n["sign_S"]] = np.sign(s).astype(np.float32)
            v[n["log_S"]] = np.log(np.abs(s)).astype(np.float32)
            v[n["L_vec"]] = stricttri2vec(l, False).astype(np.float32)
            v[n["U_vec"]] = stricttri2vec(np.triu(u, 1), True).astype(np.float32)
        elif L.kind == "coupling":
            t = template_scope(k) + "/"
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
    if any(L.kind == "sdn" for L in layers):
        v["model/b1"] = np.full((1,), -3.0, np.float32)     # cond_utils.py:43-46
        v["model/b2"] = np.full((1,), 3.0, np.float32)
    if any(L.kind == "gain" for L in layers):
        v["model/g1"] = np.full((1,), -3.0, np.float32)     # cond_utils.py:321-324
        v["model/g2"] = np.full((1,), 3.0, np.float32)
    kinds = {L.kind for L in layers}
    gain_init = float(gain_init)                             # hps.gain_init (sidd/ArgParser.py default -5.0): sdn2 / sdn3 / gain2
    if kinds & {"sdn1", "sdn2", "sdn3"}:
        v["model/b1"] = np.full((1,), -3.0, np.float32)     # cond_utils.py:90-93
        v["model/b2"] = np.full((1,), 3.0, np.float32)
    if "sdn1" in kinds:
        for iso in ISO_TABLE:
            v["model/r_gain_param_%05d" % iso] = np.zeros((1,), np.float32)               # init / c = 0 (cond_utils.py:60-68)
    if kinds & {"sdn2", "sdn3", "gain2"}:
        for iso in ISO_TABLE:
            v["model/gain_param_%05d" % iso] = np.full((1,), gain_init / 1e-1, np.float32)   # cond_utils.py:102-112, 361-366
    elif "gain3" in kinds:
        for iso in ISO_TABLE:
            v["model/gain_param_%05d" % iso] = np.full((1,), -5.0 / 1e-5, np.float32)      # cond_utils.py:401-405
    if "gain1" in kinds:
        v["model/g1"] = np.full((1,), -5.0 / 1e-5, np.float32)                             # cond_utils.py:341-342
        v["model/g2"] = np.zeros((1,), np.float32)
    if "sdn6" in kinds:
        v["model/sdn_gain/beta1"] = np.full((1,), -5.0 / C_I, np.float32)
        v["model/sdn_gain/beta2"] = np.zeros((1,), np.float32)
        v["model/sdn_gain/gain_params"] = np.full((5,), -5.0 / C_I, np.float32)
        v["model/sdn_gain/cam_params"] = np.ones((1, 5), np.float32)                        # one parameter per camera
        v["model/sdn_gain/gain_val"] = np.ones((1,), np.float32)
    if any(L.kind in ("sdn5", "gain4", "sdn4") for L in layers):
        v["model/sdn_gain/beta1"] = np.full((1,), -5.0 / C_I, np.float32)
        v["model/sdn_gain/beta2"] = np.zeros((1,), np.float32)
        v["model/sdn_gain/gain_params"] = np.full((5,), -5.0 / C_I, np.float32)
        if "sdn6" not in kinds:   # sdn6 owns the (AUTO_REUSE'd) scope's cam_params as [1, 5]
            v["model/sdn_gain/cam_params"] = np.ones((3, 5), np.float32)
        v["model/sdn_gain/gain_val"] = np.ones((1,), np.float32)
    return v


def count_trainable(variables: Dict[str, np.ndarray]) -> int:
    """``num_params`` as logged to hps.txt (train_noise_flow.py:309-312): all
    variables except the LU permutation / signs and the BN running statistics."""
    n = 0
    for name, arr in variables.items():
        if "/P_matpar" in name or "/sign_S_matpar" in name or name.endswith("/mean") or name.endswith("/var"):
            continue
        n += int(np.asarray(arr).size)
    return n
