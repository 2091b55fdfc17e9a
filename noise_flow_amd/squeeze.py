"""squeeze2d / unsqueeze2d of the reference (borealisflows/utils.py:30-86) as array reshapes, for numpy arrays and torch tensors.

SURVEY.md §8 row f-4.  The shipped configuration and every driver of the reference run with ``squeeze_factor = 1`` (identity,
``utils.py:32``); a factor > 1 cannot run upstream at all — ``NoiseFlow.prior`` builds its Gaussian from the UNSQUEEZED
``hps.x_shape`` while ``inverse`` hands it a squeezed latent (``noise_flow_model.py:399, 486-492``) — so the fused kernels have
no C = 16 variant and ``NoiseFlow`` refuses such hyper-parameters.  The two tensor rearrangements themselves are provided here
for callers that use them outside the flow (they are pure index maps):

    chessboard  out[b, i, j, c f^2 + di f + dj] = x[b, i f + di, j f + dj, c]            (utils.py:45-48)
    patch       out[b, i, j, c f^2 + di f + dj] = x[b, di H/f + i, dj W/f + j, c]        (utils.py:49-53)

An unknown ``squeeze_type`` falls back to chessboard with the reference's message (utils.py:54-59, 80-83)."""
import numpy as np


def _ops(x):
    try:
        import torch
        if isinstance(x, torch.Tensor):
            return (lambda a, s: a.reshape(s)), (lambda a, p: a.permute(p))
    except ImportError:      # pragma: no cover
        pass
    return (lambda a, s: np.reshape(a, s)), (lambda a, p: np.transpose(a, p))


def squeeze2d(x, factor=2, squeeze_type='chessboard', x_shape=None):
    """[B, H, W, C] -> [B, H/f, W/f, C f^2]  (utils.py:30-62)."""
    assert factor >= 1
    if factor == 1:
        return x
    shape = x.shape if x_shape is None else x_shape
    height, width, n_channels = int(shape[1]), int(shape[2]), int(shape[3])
    assert height % factor == 0 and width % factor == 0
    reshape, transpose = _ops(x)
    if squeeze_type == 'patch':
        x = reshape(x, [-1, factor, height // factor, factor, width // factor, n_channels])
        x = transpose(x, [0, 2, 4, 5, 1, 3])
    else:
        if squeeze_type != 'chessboard':
            print('Unknown squeeze type, using chessboard')
        x = reshape(x, [-1, height // factor, factor, width // factor, factor, n_channels])
        x = transpose(x, [0, 1, 3, 5, 2, 4])
    return reshape(x, [-1, height // factor, width // factor, n_channels * factor * factor])


def unsqueeze2d(x, factor=2, squeeze_type='chessboard'):
    """[B, H, W, C] -> [B, H f, W f, C / f^2], the inverse of :func:`squeeze2d`  (utils.py:65-86)."""
    assert factor >= 1
    if factor == 1:
        return x
    height, width, n_channels = int(x.shape[1]), int(x.shape[2]), int(x.shape[3])
    assert n_channels >= 4 and n_channels % 4 == 0
    reshape, transpose = _ops(x)
    x = reshape(x, (-1, height, width, int(n_channels / factor ** 2), factor, factor))
    if squeeze_type == 'patch':
        x = transpose(x, [0, 4, 1, 5, 2, 3])
    else:
        if squeeze_type != 'chessboard':
            print('Unknown squeeze type, using chessboard')
        x = transpose(x, [0, 1, 4, 2, 5, 3])
    return reshape(x, (-1, int(height * factor), int(width * factor), int(n_channels / factor ** 2)))
