"""One-off: the randomised large-patch sweep of tests/test_gpu_random_sweep.py over fresh seeds.
python tools/oneoff_large_sweep.py first_seed count   (OMP_NUM_THREADS=4 recommended: the oracle is numpy)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_random_sweep as sw

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    arch, width, _, fp, decomp, iso, cam, B = sw._draw_case(seed)
    try:
        sw._check_case(seed, (arch, width, sw._large_shape(seed), fp, decomp, iso, cam, min(B, 2)))
    except AssertionError as e:
        bad += 1
        print("seed", seed, str(e)[:300])
print("large-patch sweep: %d / %d draws agree" % (count - bad, count))
