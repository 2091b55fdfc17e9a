"""ctypes binding of the C ABI in ``include/noiseflow_hip.h``.

The HIP library is the product path: if ``libnoiseflow_hip.so`` is missing this
module raises — there is no CPU fallback anywhere under ``noise_flow_amd/``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libnoiseflow_hip.so")

NF_LAYER_CONV1X1 = 1
NF_LAYER_COUPLING = 2
NF_LAYER_SDN5 = 3
NF_LAYER_GAIN4 = 4
NF_LAYER_SDN4 = 5
NF_LAYER_SDN = 6
NF_LAYER_GAIN = 7
NF_LAYER_SDN1, NF_LAYER_SDN2, NF_LAYER_SDN3, NF_LAYER_SDN6 = 8, 9, 10, 11
NF_LAYER_GAIN1, NF_LAYER_GAIN2, NF_LAYER_GAIN3 = 12, 13, 14
NF_LAYER_CONV1X1_NONE, NF_LAYER_CONV1X1_LU2, NF_LAYER_PERMUTE = 15, 16, 17

NF_CFG_FP16_CNN = 1

NF_ACCUMULATE = 1
NF_NO_PRIOR = 2

NF_OK = 0
NF_SUMS_WIDE = 4
NF_SUMS_SLOTS = 64
NF_SUMS_STRIDE = 16
NF_EINVAL = -1
NF_EHIP = -2
NF_ECOND = -3
NF_ENOMEM = -4

# device op codes (csrc/nf_device.h) — exposed for the folding tests
NF_OP_MIX, NF_OP_COUPLING_FWD, NF_OP_COUPLING_REV, NF_OP_SDN_DIV, NF_OP_SDN_MUL, NF_OP_SCALE, NF_OP_SCALE_COND = 1, 2, 3, 4, 5, 6, 7


class nf_layer_desc(C.Structure):
    _fields_ = [("type", C.c_int32), ("width", C.c_int32), ("param_offset", C.c_int64)]


class nf_config(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("channels", C.c_int32),
                ("n_layers", C.c_int32), ("device", C.c_int32), ("flags", C.c_int32)]


class nf_cond(C.Structure):
    _fields_ = [("iso", C.c_float), ("cam", C.c_float), ("nlf0", C.c_float), ("nlf1", C.c_float)]


class NoiseFlowLibError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("noiseflow_hip error %d: %s" % (code, msg))
        self.code = code


# int fn(void *user, double *buf, int64_t count, void *stream)  — nf_trainer_set_sync
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

_lib = None


def load() -> C.CDLL:
    """Load the shared library once.  torch is imported first so that the HIP
    runtime torch ships (same SONAME, libamdhip64.so.7) is the one both share —
    device pointers and streams then belong to a single runtime."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "HIP extension not built: %s is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C noise_flow_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    try:
        import torch  # noqa: F401  (loads libamdhip64 first)
    except Exception:  # pragma: no cover - torch is plumbing only
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float
    lib.nf_abi_version.restype = C.c_int
    lib.nf_abi_version.argtypes = []
    lib.nf_last_error.restype = C.c_char_p
    lib.nf_last_error.argtypes = []
    lib.nf_layer_param_count.restype = i64
    lib.nf_layer_param_count.argtypes = [i32, i32]
    lib.nf_create.restype = C.c_int
    lib.nf_create.argtypes = [C.POINTER(nf_config), C.POINTER(nf_layer_desc), C.POINTER(C.c_float), C.c_size_t,
                              C.POINTER(vp)]
    lib.nf_destroy.restype = C.c_int
    lib.nf_destroy.argtypes = [vp]
    lib.nf_nll.restype = C.c_int
    lib.nf_nll.argtypes = [vp, vp, vp, i64, C.POINTER(nf_cond), vp, vp, vp, vp, vp, u32, vp]
    lib.nf_sample.restype = C.c_int
    lib.nf_sample.argtypes = [vp, vp, vp, u64, i64, f32, i64, C.POINTER(nf_cond), vp, vp]
    lib.nf_nll_batchstats.restype = C.c_int
    lib.nf_nll_batchstats.argtypes = [vp, vp, vp, i64, C.POINTER(nf_cond), vp, vp, vp, vp, vp, u32, vp, vp]
    lib.nf_sample_batchstats.restype = C.c_int
    lib.nf_sample_batchstats.argtypes = [vp, vp, vp, u64, i64, f32, i64, C.POINTER(nf_cond), vp, vp, vp]
    lib.nf_trainer_create.restype = C.c_int
    lib.nf_trainer_create.argtypes = [C.POINTER(nf_config), C.POINTER(nf_layer_desc), C.POINTER(C.c_float), C.c_size_t,
                                      i64, i32, C.POINTER(vp)]
    lib.nf_trainer_destroy.restype = C.c_int
    lib.nf_trainer_destroy.argtypes = [vp]
    lib.nf_trainer_forward_backward.restype = C.c_int
    lib.nf_trainer_forward_backward.argtypes = [vp, vp, vp, i64, C.POINTER(nf_cond), vp, vp, vp]
    lib.nf_trainer_forward.restype = C.c_int
    lib.nf_trainer_forward.argtypes = [vp, vp, vp, i64, C.POINTER(nf_cond), vp, vp]
    lib.nf_trainer_apply.restype = C.c_int
    lib.nf_trainer_apply.argtypes = [vp, vp, f32, vp]
    lib.nf_trainer_step.restype = C.c_int
    lib.nf_trainer_step.argtypes = [vp, vp, vp, i64, C.POINTER(nf_cond), f32, vp, vp]
    lib.nf_trainer_get_params.restype = C.c_int
    lib.nf_trainer_get_params.argtypes = [vp, vp, C.c_size_t, vp]
    lib.nf_trainer_set_params.restype = C.c_int
    lib.nf_trainer_set_params.argtypes = [vp, vp, C.c_size_t, vp]
    lib.nf_trainer_set_sync.restype = C.c_int
    lib.nf_trainer_set_sync.argtypes = [vp, ALLREDUCE_FN, vp, vp, i32]
    lib.nf_trainer_steps.restype = i64
    lib.nf_trainer_steps.argtypes = [vp]
    lib.nf_sums_reduce.restype = C.c_int
    lib.nf_sums_reduce.argtypes = [vp, vp, u32, vp]
    lib.nf_synth_patches.restype = C.c_int
    lib.nf_synth_patches.argtypes = [u64, i64, i64, i32, i32, f32, f32, vp, vp, vp]
    lib.nf_fold_params.restype = C.c_int
    lib.nf_fold_params.argtypes = [C.POINTER(nf_config), C.POINTER(nf_layer_desc), C.POINTER(C.c_float), C.c_size_t,
                                   i32, C.POINTER(i32), i32, C.POINTER(i32), C.POINTER(C.c_float), C.c_size_t,
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_double)]
    lib.nf_set_sync.restype = C.c_int
    lib.nf_set_sync.argtypes = [vp, ALLREDUCE_FN, vp, vp, i32]
    lib.nf_sample_eps.restype = C.c_int
    lib.nf_sample_eps.argtypes = [u64, i64, i64, i32, i32, vp, vp]
    lib.nf_tile_plan.restype = C.c_int
    lib.nf_tile_plan.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32]
    lib.nf_tile_segments.restype = C.c_int
    lib.nf_tile_segments.argtypes = [C.POINTER(nf_config), C.POINTER(nf_layer_desc), C.POINTER(C.c_float), C.c_size_t, i32,
                                     C.POINTER(i32), i32]
    lib.nf_nll_host.restype = C.c_int
    lib.nf_nll_host.argtypes = [vp, vp, vp, i32, i64, C.POINTER(nf_cond), vp, vp, vp, vp, vp, u32]
    lib.nf_sample_host.restype = C.c_int
    lib.nf_sample_host.argtypes = [vp, vp, i32, vp, u64, i64, f32, i64, C.POINTER(nf_cond), vp]
    lib.nf_fold_layout.restype = C.c_int
    lib.nf_fold_layout.argtypes = [C.POINTER(nf_config), C.POINTER(nf_layer_desc), C.POINTER(C.c_float), C.c_size_t,
                                   i32, i32, C.POINTER(i32), i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_float), C.c_size_t,
                                   C.POINTER(C.c_size_t)]
    lib.nf_kernel_path.restype = C.c_int
    lib.nf_kernel_path.argtypes = [vp, i32]
    lib.nf_workspace_bytes.restype = i64
    lib.nf_workspace_bytes.argtypes = [vp, i32, i64]
    lib.nf_reserve_workspace.restype = C.c_int
    lib.nf_reserve_workspace.argtypes = [vp, i64, i32]
    lib.nf_sdn5_scalars.restype = C.c_int
    lib.nf_sdn5_scalars.argtypes = [C.POINTER(C.c_float), C.POINTER(nf_cond), C.POINTER(C.c_double)]
    if lib.nf_abi_version() != 1:
        raise ImportError("noiseflow_hip ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != NF_OK:
        msg = load().nf_last_error()
        raise NoiseFlowLibError(rc, msg.decode("utf-8", "replace") if msg else "")


EXPORTED_SYMBOLS = (
    "nf_abi_version", "nf_last_error", "nf_layer_param_count", "nf_create", "nf_destroy", "nf_nll",
    "nf_sample", "nf_set_sync", "nf_sample_eps", "nf_tile_plan", "nf_tile_segments", "nf_nll_host", "nf_sample_host", "nf_synth_patches", "nf_fold_params", "nf_fold_layout", "nf_sdn5_scalars",
    "nf_nll_batchstats", "nf_sample_batchstats", "nf_sums_reduce", "nf_kernel_path", "nf_workspace_bytes", "nf_reserve_workspace",
    "nf_trainer_create", "nf_trainer_destroy", "nf_trainer_forward_backward", "nf_trainer_forward", "nf_trainer_apply", "nf_trainer_step",
    "nf_trainer_get_params", "nf_trainer_set_params", "nf_trainer_steps", "nf_trainer_set_sync",
)
NF_PATH_SCALAR, NF_PATH_MFMA4, NF_PATH_FP16, NF_PATH_WIDE32, NF_PATH_WIDE16, NF_PATH_WIDE32_FP16, NF_PATH_GEMM, NF_PATH_GEMM_FP16 = 0, 1, 2, 3, 4, 5, 6, 7
NF_HOST_F32, NF_HOST_F64 = 0, 1
NF_OPT_ADAM = 0
NF_OPT_MOMENTUM = 1
