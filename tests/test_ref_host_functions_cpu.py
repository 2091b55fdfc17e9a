"""CPU tier: the host-side functions either side of the hot path against OUTPUTS OF THE REFERENCE
ITSELF (tests/golden/ref_host_functions.npz, produced by tools/make_golden_ref_host.py executing the
reference's own definitions in the build container).  These rows are reference-pinned."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, SHIPPED_DIR


@pytest.fixture(scope="module")
def ref():
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_host_functions.npz"))


def test_bayer_packing_matches_reference(ref):
    from noise_flow_amd.patches import pack_raw, unpack_raw
    packed = pack_raw(ref["pack_in"])
    assert packed.shape == ref["pack_out"].shape and np.array_equal(packed, ref["pack_out"])
    assert np.array_equal(unpack_raw(ref["pack_out"]), ref["unpack_out"])


def test_patch_origins_match_reference(ref):
    from noise_flow_amd.patches import patch_origins
    for k, (h, w, ph, pw, n) in enumerate(ref["origins_cases"]):
        ii, jj, n_p = patch_origins(int(h), int(w), int(ph), int(pw), None if n < 0 else int(n))
        want = ref["origins_%d" % k]
        assert n_p == int(ref["origins_n_%d" % k])
        assert list(ii) == list(want[0]) and list(jj) == list(want[1]), k


def test_histogram_and_kl_match_reference(ref):
    from noise_flow_amd.metrics import get_histogram, kl_div_3_data
    edges = ref["kl_edges"]
    hist, centers = get_histogram(ref["kl_p"], edges, -0.3, 0.3, 200)
    np.testing.assert_allclose(hist, ref["hist_p"], rtol=0, atol=0)
    np.testing.assert_allclose(centers, ref["hist_centers"], rtol=1e-15)
    np.testing.assert_allclose(kl_div_3_data(ref["kl_p"], ref["kl_q"], edges, -0.3, 0.3, 200), ref["kl3_edges"], rtol=1e-12)
    np.testing.assert_allclose(kl_div_3_data(ref["kl_u"], ref["kl_v"]), ref["kl3_default"], rtol=1e-12)
    hd, cd = get_histogram(ref["kl_u"])
    np.testing.assert_allclose(hd, ref["hist_default"], rtol=0, atol=0)
    np.testing.assert_allclose(cd, ref["hist_default_centers"], rtol=1e-15)


def test_wrapper_hps_loader_matches_reference(ref):
    from noise_flow_amd.hps import hps_loader
    want = json.loads(str(ref["wrapper_hps_json"]))
    got = vars(hps_loader(os.path.join(SHIPPED_DIR, "hps.txt")))
    assert set(got) == set(want)
    for k, w in want.items():
        if k == "param_inits":
            g = got[k]
            assert [g[0], g[1], g[2]] == w[:3]
            assert np.asarray(g[3]).tolist() == w[3] and np.asarray(g[4]).tolist() == w[4]
        else:
            assert type(got[k]).__name__ == w[0] and got[k] == w[1], (k, got[k], w)


def test_result_logger_and_hps_files_match_reference(ref, tmp_path):
    from noise_flow_amd.harness import ResultLogger
    from noise_flow_amd.hps import hps_logger, hps_loader_raw
    cols = [str(c) for c in ref["logger_cols"]]
    rows = json.loads(str(ref["logger_rows_json"]))
    # the generator logged python / numpy scalars; replay the same values with the same types
    rows[1]["NLL"] = np.float32(rows[1]["NLL"])
    rows[1]["sdz"] = np.float64(rows[1]["sdz"])
    path = os.path.join(str(tmp_path), "test.txt")
    lg = ResultLogger(path, cols)
    for r in rows:
        lg.log(r)
    lg.close()
    lg2 = ResultLogger(path, cols, True)
    lg2.log(rows[0])
    lg2.close()
    assert open(path).read() == str(ref["logger_file"])

    class H:
        pass
    h = H()
    h.arch, h.width, h.lr, h.flag, h.none = "sdn5|unc|gain4|unc", 4, 1e-4, True, None
    h.with_comma = "a,b"
    hp = os.path.join(str(tmp_path), "hps.txt")
    hps_logger(hp, h, ["sdn_0", "Conv2d_1x1_1", "unc_1"], 2433)
    assert open(hp, newline="").read() == str(ref["hps_logger_file"])
    assert vars(hps_loader_raw(hp)) == json.loads(str(ref["hps_loader_json"]))


def test_closed_form_baselines_match_reference(ref):
    """NLL_G / NLL_SDN of PatchStatsCalculator.calc_baselines: mean over minibatches of per-patch NLLs."""
    from noise_flow_amd.metrics import nll_gauss, nll_sdn
    x, y = ref["baseline_x"], ref["baseline_y"]
    b1, b2 = ref["baseline_nlf"]
    g = np.mean([nll_gauss(x[k], np.sqrt(float(ref["baseline_vr_gauss"]))) for k in range(x.shape[0])])
    s = np.mean([nll_sdn(x[k], y[k], b1, b2) for k in range(x.shape[0])])
    assert abs(g - float(ref["baseline_nll_gauss"])) <= 1e-12 * abs(g)
    assert abs(s - float(ref["baseline_nll_sdn"])) <= 1e-12 * abs(s)


def test_fresh_sdn_gain_initialisation_matches_reference(ref):
    """train_noise_flow.init_params → the values a fresh model starts its sdn / gain parameters from."""
    from noise_flow_amd.params import init_variables, C_I
    v = init_variables("sdn5|unc|unc|unc|unc|gain4|unc|unc|unc|unc", 4, 4, 0)
    assert float(C_I) == float(ref["init_c_i"])
    assert np.all(np.asarray(v["model/sdn_gain/beta1"], np.float64) == float(ref["init_beta1"]))
    assert np.all(np.asarray(v["model/sdn_gain/beta2"], np.float64) == float(ref["init_beta2"]))
    assert np.array_equal(np.asarray(v["model/sdn_gain/gain_params"], np.float64).reshape(-1), ref["init_gain_params"])
    assert np.array_equal(np.asarray(v["model/sdn_gain/cam_params"], np.float64), ref["init_cam_params"])


def test_random_and_shuffled_patch_origins_match_reference(ref):
    """sample_indices_random / the shuffled sample_indices_uniform draw from the GLOBAL numpy RNG; a seeded run reproduces the
    reference's origins (sidd_utils.py:830-858; the shuffle is sklearn.utils.shuffle's)."""
    from noise_flow_amd.samplers import sample_indices_random, sample_indices_uniform
    np.random.seed(1234)
    ii, jj = sample_indices_random(100, 80, 32, 32, 7)
    assert [list(map(int, ii)), list(map(int, jj))] == ref["rand_origins"].tolist()
    np.random.seed(4321)
    ii, jj, n_p = sample_indices_uniform(100, 140, 32, 32, True, None)
    assert n_p == 12 and [list(map(int, ii)), list(map(int, jj))] == ref["shuf_origins"].tolist()


@pytest.mark.parametrize("mode,kw,seed", [("uniform", dict(sampling="uniform", n_pat_per_im=4, shuffle=False), None),
                                          ("shuffled", dict(sampling="uniform", n_pat_per_im=4, shuffle=True), 99),
                                          ("random", dict(sampling="random", n_pat_per_im=4), 77)])
def test_patch_and_minibatch_samplers_match_reference(ref, mode, kw, seed):
    """Image tuples -> PatchSampler -> MiniBatchSampler (one worker each: queue order), against what the reference's own
    classes put on their queues for the same tuples and seed (sidd/PatchSampler.py, sidd/MiniBatchSampler.py)."""
    import queue
    from noise_flow_amd.samplers import MiniBatchSampler, PatchSampler
    ims = [{"in": ref["samp_in"][k][None], "gt": ref["samp_gt"][k][None], "nlf0": 0.001 * (k + 1), "nlf1": 1e-6 * (k + 1),
            "iso": [100.0, 400.0, 800.0][k], "cam": float(k), "fn": "img%d|x" % k, "metadata": None} for k in range(3)]
    imq = queue.Queue()
    if seed is not None:
        np.random.seed(seed)
    ps = PatchSampler(imq, patch_height=16, max_queue_size=64, n_threads=1, **kw)
    for im in ims:
        imq.put(im)
    pats = [ps.get_queue().get(timeout=30) for _ in range(12)]
    ps.close()
    assert [p["pid"] for p in pats] == ref["samp_%s_pid" % mode].tolist()
    assert [p["iso"] for p in pats] == ref["samp_%s_iso" % mode].tolist()
    assert set(pats[0]) == {"in", "gt", "vr", "nlf0", "nlf1", "iso", "cam", "fn", "metadata", "pid"} and pats[0]["vr"] == []
    pq = queue.Queue()
    ms = MiniBatchSampler(pq, minibatch_size=6, max_queue_size=4, n_threads=1)
    for p in pats:
        pq.put(p)
    mbs = [ms.get_queue().get(timeout=30) for _ in range(2)]
    ms.close()
    for k, mb in enumerate(mbs):
        assert mb["_x"].dtype == np.float64 and mb["_y"].dtype == np.float64 and mb["pid"].dtype == np.float64
        assert np.array_equal(mb["_x"], ref["mb_%s_%d_x" % (mode, k)]) and np.array_equal(mb["_y"], ref["mb_%s_%d_y" % (mode, k)])
        assert np.array_equal(mb["pid"], ref["mb_%s_%d_pid" % (mode, k)])
        assert [mb["nlf0"][0], mb["nlf1"][0], mb["iso"][0], mb["cam"][0]] == ref["mb_%s_%d_cond" % (mode, k)].tolist()
        assert mb["fn"] == str(ref["mb_%s_%d_fn" % (mode, k)]) and mb["metadata"] is None
        assert set(mb) == {"_x", "_y", "pid", "nlf0", "nlf1", "iso", "cam", "fn", "metadata"}
    # a patch count that differs from n_pat_per_im is an error (the reference drops into pdb)
    ps2 = PatchSampler(queue.Queue(), patch_height=16, n_threads=0, n_pat_per_im=5)
    with pytest.raises(ValueError):
        ps2.patches_of(ims[0])


def test_sampling_epoch_kl_recipe_matches_reference(ref, tmp_path):
    """calc_kldiv_mb / kldiv_patch_set (sidd_utils.py:995-1058): every 5th patch, Gaussian / camera-NLF / flow / real noise
    against the real noise on the reference's bin edges, the two draws on the global numpy RNG in the reference's order."""
    from noise_flow_amd.metrics import calc_kldiv_mb
    mb = {"_y": ref["kld_y"], "_x": ref["kld_x"], "nlf0": [0.003696], "nlf1": [2e-6], "pid": np.arange(12.0), "fn": "0001_001_S6_00800|p"}
    np.random.seed(2024)
    got = calc_kldiv_mb(mb, ref["kld_xs"], None, float(ref["kld_sc_sd"]))
    np.testing.assert_allclose(got, ref["kld_avg"], rtol=1e-13, atol=0)
    assert got[3] == 0.0 and (got[:3] > 0).all()
    np.random.seed(2024)
    got2 = calc_kldiv_mb(mb, ref["kld_xs"], str(tmp_path), float(ref["kld_sc_sd"]))          # with the .mat dumps
    np.testing.assert_allclose(got2, ref["kld_avg"], rtol=1e-13, atol=0)
    assert sorted(os.listdir(os.path.join(str(tmp_path), "0001_001_S6_00800"))) == ref["kld_mat_files"].tolist()
