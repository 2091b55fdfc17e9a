"""CPU tier: host-side metrics and log formats around the hot path."""
import numpy as np


def test_histogram_and_kl_known_answers():
    from noise_flow_amd.metrics import get_histogram, kl_div_3_data, noise_bin_edges
    rng = np.random.RandomState(0)
    a = rng.rand(200000)
    h, centers = get_histogram(a, n_bins=10)
    assert h.shape == (10,) and abs(h.sum() - 1) < 1e-12 and np.allclose(h, 0.1, atol=5e-3)
    assert np.allclose(centers, np.arange(10) * 0.1 + 0.05)
    assert kl_div_3_data(a, a) == (0.0, 0.0, 0.0)
    # two Gaussians: KL(N(0,s1)||N(0,s2)) = log(s2/s1) + s1^2/(2 s2^2) - 1/2
    s1, s2 = 0.02, 0.03
    p, q = rng.randn(2000000) * s1, rng.randn(2000000) * s2
    fwd, inv, sym = kl_div_3_data(p, q, noise_bin_edges(400, -0.2, 0.2))
    want_f = np.log(s2 / s1) + s1 ** 2 / (2 * s2 ** 2) - 0.5
    want_i = np.log(s1 / s2) + s2 ** 2 / (2 * s1 ** 2) - 0.5
    assert abs(fwd - want_f) < 0.01 and abs(inv - want_i) < 0.02 and abs(sym - (fwd + inv) / 2) < 1e-15
    # out-of-range samples are dropped from the counts but stay in the divisor
    h, _ = get_histogram(np.array([0.5, 2.0]), n_bins=2)
    assert np.allclose(h, [0.0, 0.5])


def test_closed_form_nll_baselines_match_oracle():
    from noise_flow_amd import metrics
    from oracle import nf_oracle as O
    rng = np.random.RandomState(1)
    y = rng.rand(3, 8, 8, 4)
    x = rng.randn(3, 8, 8, 4) * 0.05
    np.testing.assert_allclose(metrics.nll_gauss(x, 0.05), O.nll_gauss(x, 0.05), rtol=1e-13)
    np.testing.assert_allclose(metrics.nll_sdn(x, y, 1e-3, 1e-5), O.nll_sdn(x, y, 1e-3, 1e-5), rtol=1e-13)


def test_result_logger_format(tmp_path):
    from noise_flow_amd.harness import ResultLogger, TEST_COLUMNS
    p = tmp_path / "test.txt"
    lg = ResultLogger(str(p), TEST_COLUMNS)
    lg.log({"epoch": 1, "NLL": -3.5, "NLL_G": -2.8, "NLL_SDN": -3.1, "sdz": 0.93, "msg": 1})
    lg.log({"epoch": 2, "NLL": -3.6, "NLL_G": -2.8, "NLL_SDN": -3.1, "sdz": 0.92, "msg": 0})
    lg.close()
    assert p.read_text() == "epoch\tNLL\tNLL_G\tNLL_SDN\tsdz\tmsg\n1\t-3.5\t-2.8\t-3.1\t0.93\t1\n2\t-3.6\t-2.8\t-3.1\t0.92\t0"
    lg = ResultLogger(str(p), TEST_COLUMNS, append=True)
    lg.log({"epoch": 3, "NLL": 0, "NLL_G": 0, "NLL_SDN": 0, "sdz": 0, "msg": 0})
    lg.close()
    assert p.read_text().count("\n") == 3 and p.read_text().startswith("epoch\tNLL")


def test_epoch_aggregation_is_mean_of_batch_means():
    """Quirk Q12 with unequal batches, on a stand-in model (no GPU needed)."""
    from noise_flow_amd.harness import test_epoch

    class Fake:
        def loss(self, x, y, nlf0, nlf1, iso, cam):
            return float(np.mean(x)), 1.0
    mbs = [{"_x": np.full((4, 2, 2, 4), 1.0), "_y": None, "nlf0": [0], "nlf1": [0], "iso": [100], "cam": [2]},
           {"_x": np.full((1, 2, 2, 4), 5.0), "_y": None, "nlf0": [0], "nlf1": [0], "iso": [100], "cam": [2]}]
    for nthr in (1, 3):
        mean, sd, losses = test_epoch(Fake(), mbs, n_threads=nthr)
        assert mean == 3.0 and sd == 1.0 and losses == [1.0, 5.0]     # NOT the patch-weighted 1.8
