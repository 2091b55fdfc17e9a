"""A/B aid: average HIP-event duration of the headline launch (forward NLL, 1024 patches of 32x32x4, shipped-shape model)
through a given build of the library.  python tools/ab_headline.py /path/to/libnoiseflow_hip.so [launches]
(an older build without the newest entry points loads too: missing symbols are stubbed for the binding step)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noise_flow_amd import _lib

path = os.path.abspath(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
_real = ctypes.CDLL


class _Tolerant:
    def __init__(self, lib):
        self.__dict__["_l"] = lib

    def __getattr__(self, name):
        try:
            return getattr(self._l, name)
        except AttributeError:
            return type("missing", (), {})()


_lib.LIB_PATH = path
_lib.C.CDLL = lambda p, *a, **k: _Tolerant(_real(p, *a, **k))
import torch
from noise_flow_amd import NoiseFlow, default_hps, params
from noise_flow_amd.patches import synth_patches
hps = default_hps()
if os.environ.get("NF_AB_WEIGHTS", "shipped") == "shipped":   # the bench's model; "init": fresh initialisation (l_last = 0)
    from noise_flow_amd.ckpt import load_checkpoint
    var = load_checkpoint(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models", "NoiseFlow", "ckpt", "model.ckpt.best"))
else:
    var = params.init_variables(hps.arch, 4, 4, 1234)
m = NoiseFlow([32, 32, 4], False, hps, variables=var)
BLOCK = int(os.environ.get("NF_AB_BLOCK", "0"))   # > 0: synchronise after every BLOCK launches (bench.py times blocks of 20)
R = int(os.environ.get("NF_AB_ROTATE", "12"))   # rotating input batches: 12 x 34 MB does not fit the 256 MB Infinity Cache (as bench.py)
xy = [synth_patches(0, 1024 * r, 1024, 32, 32) for r in range(R)]
lib = _lib.load()
cond = _lib.nf_cond(100, 2, 0, 0)
acc = torch.zeros(_lib.NF_SUMS_SLOTS * _lib.NF_SUMS_STRIDE, dtype=torch.float64, device="cuda")
nll = torch.empty(1024, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream()
it = [0]


def step():
    x, y = xy[it[0] % R]
    it[0] += 1
    rc = lib.nf_nll(m._flow.ptr, x.data_ptr(), y.data_ptr(), 1024, ctypes.byref(cond), nll.data_ptr(), None, None, None, acc.data_ptr(),
                    _lib.NF_ACCUMULATE | _lib.NF_SUMS_WIDE, int(st.cuda_stream))
    assert rc == 0


for _ in range(200):
    step()
torch.cuda.synchronize()
best = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    if BLOCK:
        tot = 0.0
        for _ in range(n // BLOCK):
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0.record(st)
            for _ in range(BLOCK):
                step()
            b1.record(st)
            torch.cuda.synchronize()
            tot += b0.elapsed_time(b1)
        best.append(tot / (n // BLOCK * BLOCK) * 1e3)
        continue
    for _ in range(n):
        step()
    e1.record(st)
    torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / n * 1e3)
it[0] = 0
step()
torch.cuda.synchronize()
print("%s: %s us per launch (median %.2f)  nll checksum %.9e" % (os.path.basename(path), " ".join("%.2f" % b for b in best), sorted(best)[2],
                                                               float(nll.double().sum())))
