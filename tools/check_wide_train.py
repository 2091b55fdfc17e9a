"""Width-32 trainer: matrix-core stages (NF_TRAIN_WIDE_MFMA) against the layer kernels — gradient difference and step time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noise_flow_amd import default_hps, patches
from noise_flow_amd.train import Trainer

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import trained_like_variables

width = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ARCH = sys.argv[2] if len(sys.argv) > 2 else "sdn5|unc|unc|gain4|unc"
v = trained_like_variables(ARCH, width, seed=int(sys.argv[3]) if len(sys.argv) > 3 else 6)
SHAPES = [(32, 32, 16), (20, 12, 7), (16, 16, 7), (8, 8, 9), (5, 7, 3), (48, 48, 2), (40, 56, 2), (56, 40, 3)]
if os.environ.get("CHECK_SHAPES"):   # e.g. CHECK_SHAPES=32x32x2048,16x16x5000 (more patches than slots: the persistent loops)
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["CHECK_SHAPES"].split(",")]
for (H, W, B) in SHAPES:
    x, y = patches.synth_patches(0, 0, B, height=H, width=W, nlf=(0.003696, 2e-6))
    out = {}
    for mode in ("0", "4095"):
        os.environ["NF_TRAIN_WIDE_MFMA"] = mode
        tr = Trainer([H, W, 4], default_hps(width=width, arch=ARCH), variables=v, max_batch=B)
        grads, loss = tr.forward_backward(x, y, [0], [0], [800], [2])
        out[mode] = (grads.cpu().numpy().copy(), loss.cpu().numpy().copy())
        tr.close()
    g0, g1 = out["0"][0], out["4095"][0]
    d = np.abs(g0 - g1)
    print("shape", (H, W, B), "loss", out["0"][1], out["4095"][1], "max|dg| / max|g| = %.3e" % (d.max() / np.abs(g0).max()),
          "worst index", int(d.argmax()), "of", g0.size, "g0", g0.reshape(-1)[d.argmax()], "g1", g1.reshape(-1)[d.argmax()])
for mode in (("0", "4095") if not os.environ.get("CHECK_SHAPES") else ()):
    os.environ["NF_TRAIN_WIDE_MFMA"] = mode
    for B in (138, 1024):
        x, y = patches.synth_patches(0, 0, B, nlf=(0.003696, 2e-6))
        tr = Trainer([32, 32, 4], default_hps(width=width), max_batch=B)
        for _ in range(3):
            tr.step(x, y, [0], [0], [800], [2], lr=1e-4, sync=False)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            tr.step(x, y, [0], [0], [800], [2], lr=1e-4, sync=False)
        torch.cuda.synchronize()
        print("mode", mode, "B", B, "%.3f ms/step" % ((time.perf_counter() - t) / 20 * 1e3))
        tr.close()
