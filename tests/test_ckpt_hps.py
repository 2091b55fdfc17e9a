"""CPU tier: TF-bundle reader/writer and hps loader against the shipped model."""
import os

import numpy as np

from conftest import FULL_ARCH, SHIPPED_CKPT, SHIPPED_DIR


def test_shipped_checkpoint_inventory(shipped_variables):
    v = shipped_variables
    assert len(v) == 143
    assert sum(a.size for a in v.values()) == 2721
    from noise_flow_amd.params import count_trainable
    assert count_trainable(v) == 2433                     # hps.txt:19,117 ; job_noise_flow.sh:33
    assert v["model/real_nvp_conv_template/l_1/W"].shape == (3, 3, 2, 4)
    assert v["model/real_nvp_conv_template_7/l_last/W"].shape == (3, 3, 5, 4)
    assert v["model/real_nvp_conv_template_3/l_last/logs"].shape == (1, 4)
    assert v["model/sdn_gain/cam_params"].shape == (3, 5)
    assert v["level0/bijector0/rescaling_scale0"].shape == ()
    np.testing.assert_allclose(v["model/sdn_gain/beta1"], [-4.31796], rtol=1e-6)
    np.testing.assert_allclose(v["model/sdn_gain/gain_val"], [1.06938], rtol=1e-5)
    np.testing.assert_allclose(v["model/sdn_gain/gain_params"], [-2.7584, -3.4935, -3.8160, -4.1021, -4.4802], atol=1e-4)
    want = {1: -0.1523, 2: 0.0693, 3: 0.2464, 4: 0.0949, 6: 0.0509, 7: 0.3561, 8: 0.0615, 9: 0.0991}
    for i, s in want.items():
        k = "level0/bijector%d/Conv2d_1x1_%d/log_S_matpar_lu_conv2d_1x1_%d_0" % (i, i, i)
        assert abs(float(v[k].sum()) - s) < 1e-3
        p = v[k.replace("log_S", "P")]
        assert sorted(p.sum(0)) == [1, 1, 1, 1] and sorted(p.sum(1)) == [1, 1, 1, 1]


def test_crc_is_verified(tmp_path):
    from noise_flow_amd.ckpt import load_checkpoint, crc32c
    assert crc32c(b"123456789") == 0xE3069283            # CRC-32C check value
    blob = bytearray(open(SHIPPED_CKPT + ".data-00000-of-00001", "rb").read())
    blob[100] ^= 0xFF
    (tmp_path / "m.data-00000-of-00001").write_bytes(bytes(blob))
    (tmp_path / "m.index").write_bytes(open(SHIPPED_CKPT + ".index", "rb").read())
    import pytest
    with pytest.raises(ValueError):
        load_checkpoint(str(tmp_path / "m"))
    assert len(load_checkpoint(str(tmp_path / "m"), verify_crc=False)) == 143


def test_writer_round_trip(tmp_path, shipped_variables):
    from noise_flow_amd.ckpt import load_checkpoint, save_checkpoint
    save_checkpoint(str(tmp_path / "out.ckpt"), shipped_variables)
    back = load_checkpoint(str(tmp_path / "out.ckpt"))
    assert set(back) == set(shipped_variables)
    for k, a in shipped_variables.items():
        assert back[k].shape == a.shape and back[k].dtype == a.dtype and np.array_equal(back[k], a)
    # many variables -> several restart intervals / prefix compression
    many = {"scope/var_%04d" % i: np.full((i % 5, 3), i, np.float32) for i in range(200)}
    many["ints"] = np.arange(7, dtype=np.int64)
    save_checkpoint(str(tmp_path / "many"), many)
    back = load_checkpoint(str(tmp_path / "many"))
    assert all(np.array_equal(back[k], many[k]) for k in many)


def test_hps_loader_matches_reference_coercions():
    from noise_flow_amd.hps import hps_loader, hps_loader_raw, hps_logger
    h = hps_loader(os.path.join(SHIPPED_DIR, "hps.txt"))
    assert h.arch == FULL_ARCH and h.width == 4 and h.decomp == "LU"
    assert h.n_levels == 1 and h.squeeze_factor == 1 and h.flow_permutation == 1
    assert h.n_batch_test == 207 and h.test_its == 56 and h.num_params == 2433
    assert h.do_sample is True and h.learntop is False
    assert isinstance(h.lr, float) and h.lr == 1e-4 and h.n_bins == 1024.0
    assert h.mb_qsize == "" and isinstance(h.top_shape, str)
    assert abs(h.nll_gauss + 11594.405329131772) < 1e-9 and abs(h.nll_sdn + 12718.793855082115) < 1e-9
    # param_inits is rebuilt, not parsed (NoiseFlowWrapper.py:125-137)
    c_i, b1, b2, gp, cp = h.param_inits
    assert (c_i, b1, b2) == (1.0, -5.0, 0.0) and gp.shape == (5,) and cp.shape == (3, 5) and (cp == 1).all()
    assert hps_loader_raw(os.path.join(SHIPPED_DIR, "hps.txt")).width == "4"


def test_hps_logger_round_trip(tmp_path):
    from types import SimpleNamespace
    from noise_flow_amd.hps import hps_loader, hps_logger
    from oracle.nf_oracle import layer_names
    hps = SimpleNamespace(arch=FULL_ARCH, width=4, lr=1e-4, do_sample=True, name="x")
    p = str(tmp_path / "hps.txt")
    hps_logger(p, hps, layer_names(FULL_ARCH), 2433)
    lines = open(p).read().splitlines()
    assert lines[:3] == ["sdn_0", "Conv2d_1x1_1", "unc_1"] and lines[18] == "2433"
    back = hps_loader(p)
    assert back.arch == FULL_ARCH and back.width == 4 and back.do_sample is True and back.name == "x"


def test_fresh_initialisation_honours_hps_gain_init():
    """cond_utils.py:102-112, 361-366: the per-ISO gain tables of sdn2 / sdn3 / gain2 start at hps.gain_init / 0.1 (sidd/ArgParser.py:
    --gain_init, default -5.0), not at a hard-coded -5."""
    from noise_flow_amd import params
    v = params.init_variables("sdn2|unc|gain2", 4, 4, 0)
    assert float(v["model/gain_param_00800"][0]) == -50.0
    v = params.init_variables("sdn2|unc|gain2", 4, 4, 0, gain_init=-3.0)
    assert all(float(v["model/gain_param_%05d" % iso][0]) == -30.0 for iso in (100, 400, 800, 1600, 3200))
