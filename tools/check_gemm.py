"""Quick GPU check + timing of the GEMM kernel (csrc/nf_gemm.hip): `[B=..] [DTYPE=fp16] [SIDE=64] python tools/check_gemm.py [widths...]`."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from conftest import FULL_ARCH, make_inputs, trained_like_variables  # noqa: E402
from noise_flow_amd import NoiseFlow, default_hps, _lib  # noqa: E402
from oracle.nf_oracle import NoiseFlowOracle  # noqa: E402

widths = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512]
B = int(os.environ.get("B", "512"))
DT = os.environ.get("DTYPE", "fp32")          # fp16: NF_CFG_FP16_CNN (csrc/nf_gemm16.hip)
PEAK = 157.3 if DT == "fp32" else 2500.0
S = int(os.environ.get("SIDE", "32"))         # square patch side (up to 64)
for w in widths:
    v = trained_like_variables(FULL_ARCH, w, seed=3)
    for k in v:
        if k.endswith("l_2/W") or k.endswith("l_last/W"):
            v[k] = (v[k] * np.float32((4.0 / w) ** 0.5)).astype(np.float32)
    m = NoiseFlow([S, S, 4], False, default_hps(arch=FULL_ARCH, width=w), variables=v, cnn_dtype=DT)
    x, y = make_inputs(2, S, S, seed=6)
    nll, _ = m._loss(x, y, [0.0], [0.0], [100], [2])
    ref = NoiseFlowOracle(FULL_ARCH, v, cnn_dtype=DT).nll(x, y, 100, 2)[0]
    err = float(np.max(np.abs(nll - ref) / np.abs(ref)))
    xb, yb = make_inputs(B, S, S, seed=1)
    xb, yb = torch.tensor(xb).cuda(), torch.tensor(yb).cuda()
    lib = _lib.load()
    cond = _lib.nf_cond(100.0, 2.0, 0.0, 0.0)
    out = torch.empty(B, dtype=torch.float32, device="cuda")
    st = int(torch.cuda.current_stream().cuda_stream)

    def step():
        _lib.check(lib.nf_nll(m._flow.ptr, xb.data_ptr(), yb.data_ptr(), B, C.byref(cond), out.data_ptr(), None, None, None, None, 0, st))
    step(); step()
    torch.cuda.synchronize()
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    mac = 16 + 18 * w + w * w + 36 * (w + 1)
    flop = (8 * (2 * mac + 56) + 40) * S * S * B
    print("%dx%d " % (S, S) + "width %d %s: path %d nll rel err %.2e | B=%d %.2f ms = %.0f patches/s = %.1f TFLOP/s = %.3f of the %s matrix peak" % (
        w, DT, lib.nf_kernel_path(m._flow.ptr, 0), err, B, ms, B / ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / PEAK, DT), flush=True)
