"""Generate tests/golden/ref_host_functions.npz: OUTPUTS OF THE REFERENCE ITSELF for the host-side
functions either side of the hot path that do not need TensorFlow to run.

The reference's modules cannot be imported here (they import tensorflow / h5py / imageio at module
level), but these functions are pure Python / numpy.  This script — run in the build container only,
where /root/reference exists — takes the definitions it names out of the reference's source files
with `ast`, executes THOSE definitions (nothing is restated or stubbed), applies them to seeded inputs
and stores inputs + outputs as data.  The committed .npz holds no reference source.

    python tools/make_golden_ref_host.py        (needs /root/reference)

Pinned: pack_raw / unpack_raw, sample_indices_uniform (also shuffled), sample_indices_random, get_histogram, kl_div_3_data,
kl_div_forward / kldiv_patch_set / calc_kldiv_mb (sidd/sidd_utils.py:732-764, 830-858, 995-1058, 1202-1274), PatchSampler and
MiniBatchSampler on in-memory image tuples (sidd/PatchSampler.py:20-79, sidd/MiniBatchSampler.py:19-78; their `Thread` is given
as a daemon subclass so that this script can end), NoiseFlowWrapper.hps_loader on the shipped hps.txt
(borealisflows/NoiseFlowWrapper.py:96-138), ResultLogger / hps_logger / hps_loader
(borealisflows/utils.py:90-135), the Gaussian / camera-NLF baseline formulas of
PatchStatsCalculator.calc_baselines (sidd/PatchStatsCalculator.py:92-123), the initial sdn / gain
parameter values of train_noise_flow.init_params (train_noise_flow.py:201-214).
"""
import ast
import gc
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def take(path, names, ns):
    """exec the top-level (or class-level) definitions `names` of a reference file in namespace `ns`."""
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    found = set()
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names and node.name not in found:
            seg = ast.get_source_segment(src, node)
            import textwrap
            exec(compile(textwrap.dedent(seg), path + ":" + node.name, "exec"), ns)
            found.add(node.name)
    missing = set(names) - found
    if missing:
        raise RuntimeError("not found in %s: %s" % (path, missing))
    return ns


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    rng = np.random.RandomState(20190829)
    out = {}

    # ---- sidd_utils: Bayer packing, patch origins, histograms, KL --------------------------------
    ns = {"np": np, "gc": gc}
    take("sidd/sidd_utils.py", ["pack_raw", "unpack_raw", "sample_indices_uniform", "get_histogram", "kl_div_3_data"], ns)
    raw = rng.rand(12, 16).astype(np.float32)
    packed = ns["pack_raw"](raw.copy())
    out["pack_in"] = raw
    out["pack_out"] = packed
    out["unpack_out"] = ns["unpack_raw"](packed)
    cases = [(100, 100, 32, 32, None), (64, 200, 32, 64, None), (33, 33, 32, 32, None), (100, 70, 32, 32, 5), (31, 100, 32, 32, None)]
    out["origins_cases"] = np.asarray([[h, w, ph, pw, -1 if n is None else n] for h, w, ph, pw, n in cases], np.int64)
    for k, (h, w, ph, pw, n) in enumerate(cases):
        ii, jj, n_p = ns["sample_indices_uniform"](h, w, ph, pw, False, n)
        out["origins_%d" % k] = np.asarray([list(map(int, ii)), list(map(int, jj))], np.int64).reshape(2, -1)
        out["origins_n_%d" % k] = np.asarray(n_p)
    p_data = (rng.randn(4000) * 0.05).astype(np.float64)
    q_data = (rng.randn(5000) * 0.07 + 0.01).astype(np.float64)
    out["kl_p"], out["kl_q"] = p_data, q_data
    edges = np.linspace(-0.3, 0.3, 201)
    out["kl_edges"] = edges
    hist, centers = ns["get_histogram"](p_data, edges, -0.3, 0.3, 200)
    out["hist_p"], out["hist_centers"] = hist, centers
    out["kl3_edges"] = np.asarray(ns["kl_div_3_data"](p_data, q_data, edges, -0.3, 0.3, 200))
    u, v = rng.rand(3000), rng.rand(2500) ** 1.3
    out["kl_u"], out["kl_v"] = u, v
    out["kl3_default"] = np.asarray(ns["kl_div_3_data"](u, v))          # default [0,1] range, 1000 bins
    hd, cd = ns["get_histogram"](u)
    out["hist_default"], out["hist_default_centers"] = hd, cd

    # ---- NoiseFlowWrapper.hps_loader on the shipped hps.txt ----------------------------------------
    ns = {"np": np}
    take("borealisflows/NoiseFlowWrapper.py", ["hps_loader"], ns)
    hps = ns["hps_loader"](None, os.path.join(REF, "models", "NoiseFlow", "hps.txt"))
    d = {}
    for k, val in vars(hps).items():
        if k == "param_inits":
            d[k] = [val[0], val[1], val[2], np.asarray(val[3]).tolist(), np.asarray(val[4]).tolist()]
        else:
            d[k] = [type(val).__name__, val]
    out["wrapper_hps_json"] = np.asarray(json.dumps(d, sort_keys=True))

    # ---- utils: ResultLogger, hps_logger, hps_loader ---------------------------------------------
    ns = {}
    take("borealisflows/utils.py", ["ResultLogger", "hps_logger", "hps_loader"], ns)
    tmp = tempfile.mkdtemp()
    cols = ["epoch", "NLL", "NLL_G", "NLL_SDN", "sdz", "msg"]
    lg = ns["ResultLogger"](os.path.join(tmp, "test.txt"), cols)
    rows = [{"epoch": 1, "NLL": -3.25, "NLL_G": -7076.358503, "NLL_SDN": -7695.98, "sdz": 0.5, "msg": 1},
            {"epoch": 10, "NLL": np.float32(-11769.95), "NLL_G": 0.0, "NLL_SDN": 1e-9, "sdz": np.float64(0.9258964), "msg": 0}]
    for r in rows:
        lg.log(r)
    lg.f_log.close()
    lg2 = ns["ResultLogger"](os.path.join(tmp, "test.txt"), cols, True)      # append mode: no second header
    lg2.log(rows[0])
    lg2.f_log.close()
    out["logger_cols"] = np.asarray(cols)
    out["logger_rows_json"] = np.asarray(json.dumps([{k: (float(v) if not isinstance(v, int) else v) for k, v in r.items()} for r in rows]))
    out["logger_file"] = np.asarray(open(os.path.join(tmp, "test.txt")).read())

    class H:
        pass
    h = H()
    h.arch, h.width, h.lr, h.flag, h.none = "sdn5|unc|gain4|unc", 4, 1e-4, True, None
    h.with_comma = "a,b"
    ns["hps_logger"](os.path.join(tmp, "hps.txt"), h, ["sdn_0", "Conv2d_1x1_1", "unc_1"], 2433)
    out["hps_logger_file"] = np.asarray(open(os.path.join(tmp, "hps.txt"), newline="").read())
    back = ns["hps_loader"](os.path.join(tmp, "hps.txt"))
    out["hps_loader_json"] = np.asarray(json.dumps(vars(back), sort_keys=True))

    # ---- PatchStatsCalculator.calc_baselines: the two closed-form NLLs ------------------------------
    ns = {"np": np}
    src = open(os.path.join(REF, "sidd/PatchStatsCalculator.py")).read()
    tree = ast.parse(src)
    fn = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "calc_baselines"][0]
    import textwrap
    body = textwrap.dedent(ast.get_source_segment(src, fn))
    import queue

    class Self:
        pass
    s = Self()
    s.hps = H()
    s.hps.test_its = 2
    s.stats = {"sc_in_vr": 0.0009}
    s.save_dir = tmp
    s.file_postfix = ""
    q = queue.Queue()
    mbs = []
    for k in range(2):
        y = rng.rand(3, 8, 8, 4)
        x = rng.randn(3, 8, 8, 4) * np.sqrt(0.003696 * y + 2e-6)
        mbs.append((x, y))
        q.put({"_x": x, "_y": y, "nlf0": [0.003696], "nlf1": [2e-6]})
    ns.update({"os": os, "save": np.save, "time": __import__("time")})
    exec(compile(body, "PatchStatsCalculator.py:calc_baselines", "exec"), ns)
    nll_gauss, _, nll_sdn, _ = ns["calc_baselines"](s, q)
    out["baseline_x"] = np.stack([m[0] for m in mbs])
    out["baseline_y"] = np.stack([m[1] for m in mbs])
    out["baseline_vr_gauss"] = np.asarray(0.0009)
    out["baseline_nlf"] = np.asarray([0.003696, 2e-6])
    out["baseline_nll_gauss"] = np.asarray(nll_gauss)
    out["baseline_nll_sdn"] = np.asarray(nll_sdn)

    # ---- train_noise_flow.init_params: initial values of the sdn / gain parameters ------------------
    ns = {"np": np}
    take("train_noise_flow.py", ["init_params"], ns)
    h1 = H()
    h1.arch = "sdn5|unc|unc|unc|unc|gain4|unc|unc|unc|unc"
    ns["init_params"](h1)
    c_i, b1, b2, gp, cp = h1.param_inits
    out["init_c_i"], out["init_beta1"], out["init_beta2"] = np.asarray(c_i), np.asarray(b1), np.asarray(b2)
    out["init_gain_params"], out["init_cam_params"] = np.asarray(gp), np.asarray(cp)

    # ---- sidd_utils: random / shuffled patch origins (global numpy RNG) ------------------------------
    from sklearn.utils import shuffle as sk_shuffle                      # what sidd_utils imports as `shuffle`
    ns = {"np": np, "gc": gc, "shuffle": sk_shuffle}
    take("sidd/sidd_utils.py", ["sample_indices_uniform", "sample_indices_random"], ns)
    np.random.seed(1234)
    ii, jj = ns["sample_indices_random"](100, 80, 32, 32, 7)
    out["rand_origins"] = np.asarray([list(map(int, ii)), list(map(int, jj))], np.int64)
    np.random.seed(4321)
    ii, jj, n_p = ns["sample_indices_uniform"](100, 140, 32, 32, True, None)
    out["shuf_origins"] = np.asarray([list(map(int, ii)), list(map(int, jj))], np.int64)

    # ---- PatchSampler / MiniBatchSampler on in-memory image tuples (one worker thread each: queue order) ------
    import threading

    class DaemonThread(threading.Thread):                                  # the reference's workers never return
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.daemon = True
    ns = {"np": np, "queue": queue, "Thread": DaemonThread, "time": __import__("time"), "random": __import__("random"),
          "sample_indices_uniform": ns["sample_indices_uniform"], "sample_indices_random": ns["sample_indices_random"]}
    take("sidd/PatchSampler.py", ["PatchSampler"], ns)
    take("sidd/MiniBatchSampler.py", ["MiniBatchSampler"], ns)
    ims = []
    for k in range(3):
        gt = rng.rand(1, 38, 38, 4)
        ims.append({"in": gt + rng.randn(1, 38, 38, 4) * 0.01, "gt": gt, "nlf0": 0.001 * (k + 1), "nlf1": 1e-6 * (k + 1),
                    "iso": [100.0, 400.0, 800.0][k], "cam": float(k), "fn": "img%d|x" % k, "metadata": None})
    out["samp_in"] = np.stack([im["in"][0] for im in ims])
    out["samp_gt"] = np.stack([im["gt"][0] for im in ims])
    for mode, kw, seed in (("uniform", dict(sampling="uniform", n_pat_per_im=4, shuffle=False), None),
                           ("shuffled", dict(sampling="uniform", n_pat_per_im=4, shuffle=True), 99),
                           ("random", dict(sampling="random", n_pat_per_im=4), 77)):
        imq = queue.Queue()
        if seed is not None:
            np.random.seed(seed)
        ps = ns["PatchSampler"](imq, patch_height=16, max_queue_size=64, n_threads=1, **kw)
        for im in ims:
            imq.put(im)
        pats = [ps.get_queue().get(timeout=30) for _ in range(12)]
        out["samp_%s_pid" % mode] = np.asarray([p["pid"] for p in pats], np.int64)
        out["samp_%s_iso" % mode] = np.asarray([p["iso"] for p in pats])
        pq = queue.Queue()
        ms = ns["MiniBatchSampler"](pq, minibatch_size=6, max_queue_size=4, n_threads=1)
        for p in pats:
            pq.put(p)
        mbs = [ms.get_queue().get(timeout=30) for _ in range(2)]
        for k, mb in enumerate(mbs):
            assert mb["_x"].dtype == np.float64
            out["mb_%s_%d_x" % (mode, k)], out["mb_%s_%d_y" % (mode, k)], out["mb_%s_%d_pid" % (mode, k)] = mb["_x"], mb["_y"], mb["pid"]
            out["mb_%s_%d_cond" % (mode, k)] = np.asarray([mb["nlf0"][0], mb["nlf1"][0], mb["iso"][0], mb["cam"][0]])
            out["mb_%s_%d_fn" % (mode, k)] = np.asarray(mb["fn"])

    # ---- calc_kldiv_mb / kldiv_patch_set: the sampling-epoch KL recipe of the training driver ---------------------
    from scipy.io import savemat
    ns = {"np": np, "os": os, "queue": queue, "savemat": savemat}
    take("sidd/sidd_utils.py", ["unpack_raw", "get_histogram", "kl_div_forward", "kldiv_patch_set", "calc_kldiv_mb"], ns)
    yk = rng.rand(12, 16, 16, 4)
    mbk = {"_y": yk, "_x": rng.randn(12, 16, 16, 4) * np.sqrt(0.003696 * yk + 2e-6), "nlf0": [0.003696], "nlf1": [2e-6],
           "pid": np.arange(12.0), "fn": "0001_001_S6_00800|p"}
    xsk = rng.randn(12, 16, 16, 4) * np.sqrt(0.0042 * yk + 3e-6)
    np.random.seed(2024)
    out["kld_avg"] = np.asarray(ns["calc_kldiv_mb"](mbk, xsk, os.path.join(tmp, "vis"), 0.035))
    out["kld_y"], out["kld_x"], out["kld_xs"] = mbk["_y"], mbk["_x"], xsk
    out["kld_sc_sd"] = np.asarray(0.035)
    out["kld_mat_files"] = np.asarray(sorted(os.listdir(os.path.join(tmp, "vis", "0001_001_S6_00800"))))

    path = os.path.join(ROOT, "tests", "golden", "ref_host_functions.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
