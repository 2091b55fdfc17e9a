import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SHIPPED_DIR = os.path.join(ROOT, "models", "NoiseFlow")
SHIPPED_CKPT = os.path.join(SHIPPED_DIR, "ckpt", "model.ckpt.best")
FULL_ARCH = "sdn5|unc|unc|unc|unc|gain4|unc|unc|unc|unc"
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def shipped_variables():
    from noise_flow_amd.ckpt import load_checkpoint
    return load_checkpoint(SHIPPED_CKPT)


@pytest.fixture(scope="session")
def oracle_full(shipped_variables):
    from oracle.nf_oracle import NoiseFlowOracle
    return NoiseFlowOracle(FULL_ARCH, shipped_variables)


def make_inputs(B, H=32, W=32, seed=0, b1=0.000479, b2=0.000002):
    """Seeded SIDD-like inputs: clean y ~ U(0,1), noise x ~ N(0, b1*y + b2)."""
    rng = np.random.RandomState(seed)
    y = rng.rand(B, H, W, 4).astype(np.float32)
    x = (rng.randn(B, H, W, 4) * np.sqrt(b1 * y + b2)).astype(np.float32)
    return x, y


def trained_like_variables(arch, width, seed=0, channels=4):
    """Fresh variables perturbed so that every term of the stack is exercised
    (non-zero l_last / logs, non-trivial BN statistics, scales ~0.5, gain != 1)."""
    from noise_flow_amd import params
    rng = np.random.RandomState(seed + 1000)
    v = params.init_variables(arch, width, channels, seed)
    for k in list(v):
        a = v[k]
        if k.endswith("l_1/W") or k.endswith("l_2/W"):
            v[k] = (rng.randn(*a.shape) * 0.4).astype(np.float32)
        elif k.endswith("l_last/W"):
            v[k] = (rng.randn(*a.shape) * 0.15).astype(np.float32)
        elif k.endswith("/b"):
            v[k] = (rng.randn(*a.shape) * 0.1).astype(np.float32)
        elif k.endswith("l_last/logs"):
            v[k] = (rng.randn(*a.shape) * 0.1).astype(np.float32)
        elif k.endswith("/mean"):
            v[k] = (rng.randn(*a.shape) * 0.2).astype(np.float32)
        elif k.endswith("/var"):
            v[k] = (0.5 + rng.rand(*a.shape)).astype(np.float32)
        elif "rescaling_scale" in k:
            v[k] = np.float32(0.3 + 0.6 * rng.rand())
        elif "log_S" in k or "L_vec" in k or "U_vec" in k:
            v[k] = (a + rng.randn(*a.shape).astype(np.float32) * 0.1).astype(np.float32)
        elif k.endswith("gain_val"):
            v[k] = np.asarray([1.3], np.float32)
    return v
