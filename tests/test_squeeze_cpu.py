"""squeeze2d / unsqueeze2d (borealisflows/utils.py:30-86): the two index maps the reference's reshape /
transpose chains define, checked element by element, on numpy arrays and torch tensors."""
import numpy as np
import pytest

from noise_flow_amd.squeeze import squeeze2d, unsqueeze2d


def _brute(x, f, kind):
    B, H, W, C = x.shape
    out = np.zeros((B, H // f, W // f, C * f * f), x.dtype)
    for i in range(H // f):
        for j in range(W // f):
            for c in range(C):
                for di in range(f):
                    for dj in range(f):
                        src = x[:, i * f + di, j * f + dj, c] if kind == 'chessboard' else x[:, di * (H // f) + i, dj * (W // f) + j, c]
                        out[:, i, j, c * f * f + di * f + dj] = src
    return out


@pytest.mark.parametrize("kind", ["chessboard", "patch"])
@pytest.mark.parametrize("f,shape", [(2, (3, 4, 6, 4)), (2, (1, 32, 32, 4)), (4, (2, 8, 8, 1))])
def test_squeeze2d_is_the_reference_index_map(kind, f, shape):
    x = np.arange(np.prod(shape), dtype=np.float32).reshape(shape)
    y = squeeze2d(x, f, kind)
    assert y.shape == (shape[0], shape[1] // f, shape[2] // f, shape[3] * f * f)
    np.testing.assert_array_equal(y, _brute(x, f, kind))
    np.testing.assert_array_equal(unsqueeze2d(y, f, kind), x)


def test_factor_one_is_the_identity_and_unknown_type_is_chessboard(capsys):
    x = np.random.RandomState(0).rand(2, 4, 4, 4).astype(np.float32)
    assert squeeze2d(x, 1) is x and unsqueeze2d(x, 1) is x       # utils.py:32, 67
    np.testing.assert_array_equal(squeeze2d(x, 2, 'other'), squeeze2d(x, 2, 'chessboard'))
    assert 'Unknown squeeze type' in capsys.readouterr().out


def test_torch_tensors_take_the_same_path():
    import torch
    x = torch.arange(2 * 4 * 4 * 4, dtype=torch.float32).reshape(2, 4, 4, 4)
    for kind in ("chessboard", "patch"):
        y = squeeze2d(x, 2, kind)
        np.testing.assert_array_equal(y.numpy(), squeeze2d(x.numpy(), 2, kind))
        assert torch.equal(unsqueeze2d(y, 2, kind), x)
