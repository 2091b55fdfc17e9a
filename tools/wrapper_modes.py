"""The two wrapper modes on the shipped checkpoint against S6-NLF noise:
sample standard deviation relative to the NLF's sqrt(b1 y + b2) and the marginal KL of metrics.kl_div_3_data.
    python tools/wrapper_modes.py"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noise_flow_amd import NoiseFlowWrapper, metrics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "models", "NoiseFlow")
rng = np.random.RandomState(0)
y = rng.rand(64, 32, 32, 4).astype(np.float32)
out = {}
for iso, b1, b2 in ((100.0, 0.000479, 0.000002), (800.0, 0.003696, 0.00001)):
    sd = np.sqrt(b1 * y + b2)
    real = (rng.randn(*y.shape) * sd).astype(np.float32)
    for name, kw in (("default", {}), ("batch_bn_only", {"bn_mode": "batch"}), ("sample_first_only", {"binding": "sample_first"}),
                     ("reference", {"compat": "reference"})):
        for temp in (1.0, 0.6):
            w = NoiseFlowWrapper(path, sampling_temperature=temp, seed=1, **kw)
            x = np.asarray(w.sample_noise_nf(y, 0.0, 0.0, iso, 2.0))
            ratio = float(np.sqrt(np.mean((x / sd) ** 2)))
            kl = float(metrics.kl_div_3_data(real, x, bin_edges=metrics.noise_bin_edges())[0])
            out["iso%d/%s/temp%.1f" % (iso, name, temp)] = {"sd_ratio": round(ratio, 4), "kl": kl}
for k, v in out.items():
    print(k, json.dumps(v))
