"""One-off: the randomised sweeps of tests/test_gpu_random_sweep.py at coupling widths their committed draws do not visit
(NF_SWEEP_WIDTHS overrides the drawn width; everything else about a draw stays).
    python tools/oneoff_width_sweeps.py grad 2,5,12,24,40,64,100,200 300 40
    python tools/oneoff_width_sweeps.py bs 3,5,12,24 200 40          (batch-statistics evaluation, both directions)
    python tools/oneoff_width_sweeps.py eval 3,5,12,24 0 60          (evaluation, both directions)"""
import os
import sys

R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, R)
kind, widths, first, count = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
# NF_SWEEP_MAXHW=12 caps the patch side (fewer activations -> fewer of them on a ReLU kink at large widths); NF_SWEEP_MAXKINKS=200
# lets the branch solver of tests/conftest.py::grads_match_up_to_kinks work on draws with that many on-kink activations
os.environ["NF_SWEEP_WIDTHS"] = widths
import test_gpu_random_sweep as S  # noqa: E402

fn = {"grad": S.test_random_model_training_gradients, "bs": S.test_random_model_batch_statistics, "eval": S.test_random_model_matches_oracle}[kind]
bad = []
for seed in range(first, first + count):
    try:
        fn(seed)
    except BaseException as e:          # pytest.skip / pytest.raises outcomes included
        if type(e).__name__ in ("Skipped",):
            continue
        bad.append(seed)
        print("FAIL seed %d: %s" % (seed, str(e)[:600]), flush=True)
print("%s sweep at widths %s: %d / %d agree; failures %s" % (kind, widths, count - len(bad), count, bad), flush=True)
