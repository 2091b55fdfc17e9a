"""One-off: tests/test_gpu_random_sweep.py::test_random_model_training_gradients over seeds beyond the committed 300..359
(deterministic kink handling: no re-draws).   python tools/oneoff_grad_sweep.py <first> <count>  |  seeds <s1> <s2> ..."""
import os
import sys

R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, R)
import test_gpu_random_sweep as S  # noqa: E402

if sys.argv[1] == "seeds":
    seeds = [int(a) for a in sys.argv[2:]]
else:
    seeds = list(range(int(sys.argv[1]), int(sys.argv[1]) + int(sys.argv[2])))
count = len(seeds)
bad = []
for seed in seeds:
    try:
        S.test_random_model_training_gradients(seed)
    except Exception as e:
        bad.append(seed)
        print("FAIL seed %d: %s" % (seed, str(e)[:500]), flush=True)
print("training gradients: %d / %d agree; failures %s" % (count - len(bad), count, bad), flush=True)
