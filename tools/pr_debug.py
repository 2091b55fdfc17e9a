"""Debug aid: the width-32 training step on the patch-resident stages (NF_TRAIN_PR=1) against the stage kernels of nf_train_wide.h
(NF_TRAIN_PR=0), variable by variable:  python tools/pr_debug.py [arch] [B]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import make_inputs, trained_like_variables
from noise_flow_amd import default_hps
from noise_flow_amd.train import Trainer

arch = sys.argv[1] if len(sys.argv) > 1 else "unc"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 5
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 9
brief = len(sys.argv) > 4
v = trained_like_variables(arch, 32, seed=seed)
x, y = make_inputs(B, 32, 32, seed=seed + 20)
res = {}
for pr in ("0", "1"):
    os.environ["NF_TRAIN_PR"] = pr
    tr = Trainer([32, 32, 4], default_hps(arch=arch, width=32), variables=v, max_batch=max(64, B))
    g, loss = tr.forward_backward(x, y, [0.0], [0.0], [400], [1])
    res[pr] = (tr.raw_to_variables(g.cpu().numpy().copy()), loss.cpu().numpy().copy(), tr.variables)
    tr.close()
print("loss", res["0"][1], res["1"][1])
worst = 0.0
for nm in res["0"][0]:
    a, b = np.asarray(res["0"][0][nm], np.float64), np.asarray(res["1"][0][nm], np.float64)
    if np.abs(a).max() == 0 and np.abs(b).max() == 0:
        continue
    if not nm.endswith(("l_1/b", "l_2/b")):
        worst = max(worst, np.abs(a - b).max() / max(np.abs(a).max(), 1e-30))
    if not brief:
        print("%-70s ref %.3e  diff %.3e  rel %.2e" % (nm[-70:], np.abs(a).max(), np.abs(a - b).max(), np.abs(a - b).max() / max(np.abs(a).max(), 1e-30)))
print("worst relative difference (without the analytically-zero biases): %.2e" % worst)
for nm in res["0"][2]:
    a, b = np.asarray(res["0"][2][nm], np.float64), np.asarray(res["1"][2][nm], np.float64)
    d = np.abs(a - b).max()
    if d > 1e-6 * max(np.abs(a).max(), 1e-6):
        print("variable moved differently:", nm, np.abs(a).max(), d)
