"""One-off: the randomised sweeps of tests/test_gpu_random_sweep.py over seeds beyond the committed ones.
    python tools/oneoff_sweeps.py <n_eval> <n_fp16> <n_batchstats> <n_wide_fp16>"""
import os
import sys

R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, R)
import numpy as np  # noqa: E402
import test_gpu_random_sweep as S  # noqa: E402

n_eval, n_fp16, n_bs, n_wf = (int(v) for v in (sys.argv[1:5] + ["0"] * 4)[:4])
bad = []


def run(name, fn, seeds):
    ok = 0
    for seed in seeds:
        try:
            fn(seed)
            ok += 1
        except Exception as e:
            bad.append((name, seed))
            print("FAIL %s seed %d: %s" % (name, seed, str(e)[:400]), flush=True)
    print("%s: %d / %d agree" % (name, ok, len(list(seeds))), flush=True)


def wide_fp16(seed):
    """A random model at a width beyond 32 in the fp16-CNN mode (nf_gemm16.hip, both variants) against the oracle's emulation."""
    from noise_flow_amd import NoiseFlow, default_hps, params
    from oracle.nf_oracle import NoiseFlowOracle
    from conftest import make_inputs, trained_like_variables
    arch, _, (H, W), fp, decomp, iso, cam, B = S._draw_case(seed)
    rng = np.random.RandomState(seed)
    width = int(rng.choice([33, 48, 64, 96, 128, 160, 200, 256, 384, 512]))
    if "unc" not in arch.split("|"):
        arch = "unc|" + arch
    while H * W > 2048:
        H = max(1, H // 2)
    v = params.init_variables(arch, width, 4, seed, fp, decomp)
    base = trained_like_variables(arch, width, seed=seed)
    for k in v:
        if k in base:
            v[k] = base[k]
    v = S._condition(v, arch, width, iso, rng)
    m = NoiseFlow([H, W, 4], False, default_hps(arch=arch, width=width, flow_permutation=fp, decomp=decomp), variables=v, cnn_dtype="fp16")
    o16 = NoiseFlowOracle(arch, v, cnn_dtype="fp16", flow_permutation=fp, decomp=decomp)
    o32 = NoiseFlowOracle(arch, v, flow_permutation=fp, decomp=decomp)
    oplain = NoiseFlowOracle(arch, v, cnn_dtype="fp16_plain", flow_permutation=fp, decomp=decomp)
    B = min(B, 2)
    x, y = make_inputs(B, H, W, seed=seed + 7)
    args = ([0.0], [0.0], [iso], [cam])
    nll, _ = m._loss(x, y, *args)
    ref, _, rz = o16.nll(x, y, iso, cam)
    n32 = o32.nll(x, y, iso, cam)[0]
    noise = np.abs(oplain.nll(x, y, iso, cam)[0] - n32).max()
    assert (np.abs(nll - ref) <= 1e-4 * np.abs(ref) + 0.05 * noise + 1e-3).all(), "nll vs emulation %.3e (noise %.3e) w=%d %dx%d %s" % (
        np.abs(nll - ref).max(), noise, width, H, W, arch)
    z, _ = m.inverse(x, None, y, *args)
    assert np.abs(np.asarray(z, np.float64) - rz).max() <= 2e-3 * np.abs(rz).max(), "z vs emulation w=%d %dx%d %s" % (width, H, W, arch)
    eps = np.random.RandomState(seed + 3).randn(B, H, W, 4).astype(np.float32)
    xs = m.sample(y, 0.8, y, *args, eps=eps)
    rx = o16.sample(eps, 0.8, y, iso, cam)
    assert np.abs(np.asarray(xs, np.float64) - rx).max() <= 2e-3 * np.abs(rx).max(), "sample vs emulation w=%d" % width


OFF = int(os.environ.get("NF_SWEEP_OFFSET", "0"))      # fresh seeds per round: round 3 ran offset 0, round 4 offset 4000
run("eval", lambda s: S._check_case(s, S._draw_case(s)), range(20000 + OFF, 20000 + OFF + n_eval))
run("fp16", S.test_random_model_fp16_cnn_mode, range(21000 + OFF, 21000 + OFF + n_fp16))
run("batchstats", S.test_random_model_batch_statistics, range(22000 + OFF, 22000 + OFF + n_bs))
run("wide_fp16", wide_fp16, range(23000 + OFF, 23000 + OFF + n_wf))
print("done, failures:", bad, flush=True)
