"""CPU ORACLE (test infrastructure only) — numpy Philox4x32-10 and the synthetic
patch generator, restating what ``csrc/nf_kernels.hip`` does on the device so
that "bit-exact patch indexing" can be checked: the u32 stream and the uniform
``y`` are bit-exact; the Box-Muller normals agree to fp32 libm rounding.
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
STREAM_Y, STREAM_XEPS, STREAM_SAMP = 0, 1, 2


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, np.uint32).copy() for c in (c0, c1, c2, c3))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def philox_pixels(seed: int, patches: np.ndarray, n_pixels: int, stream: int):
    """u32[4] per (patch, pixel): counter = (patch_lo, patch_hi, pixel, stream), key = seed."""
    patches = np.asarray(patches, np.uint64)
    pl = (patches & np.uint64(0xFFFFFFFF)).astype(np.uint32)[:, None]
    ph = (patches >> np.uint64(32)).astype(np.uint32)[:, None]
    px = np.arange(n_pixels, dtype=np.uint32)[None, :]
    shape = (patches.size, n_pixels)
    r = philox4x32_10(np.broadcast_to(pl, shape), np.broadcast_to(ph, shape), np.broadcast_to(px, shape),
                      np.full(shape, stream, np.uint32), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return np.stack(r, axis=-1)   # [n_patches, n_pixels, 4]


def u01_24(r):
    return (r >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def u01_open(r):
    return (r >> np.uint32(9)).astype(np.float32) * np.float32(2.0 ** -23) + np.float32(2.0 ** -24)


def normals(r):
    """Box-Muller on (r0,r1) and (r2,r3) → 4 N(0,1) per pixel, fp32 arithmetic."""
    out = np.empty(r.shape, np.float32)
    for a, b in ((0, 1), (2, 3)):
        u1, u2 = u01_open(r[..., a]), u01_open(r[..., b])
        rad = np.sqrt(np.float32(-2.0) * np.log(u1))
        ang = np.float32(6.283185307179586) * u2
        out[..., a] = rad * np.cos(ang)
        out[..., b] = rad * np.sin(ang)
    return out


def synth_patches(seed, first_patch, count, height=32, width=32, beta1=0.000479, beta2=0.000002):
    """→ (x, y) float32 [count, H, W, 4]; see noise_flow_amd.patches.synth_patches."""
    ks = np.arange(first_patch, first_patch + count, dtype=np.uint64)
    hw = height * width
    y = u01_24(philox_pixels(seed, ks, hw, STREAM_Y))
    eps = normals(philox_pixels(seed, ks, hw, STREAM_XEPS))
    x = eps * np.sqrt(np.float32(beta1) * y + np.float32(beta2))
    return x.reshape(count, height, width, 4), y.reshape(count, height, width, 4)


def sample_eps(seed, first_patch, count, height=32, width=32):
    """The in-kernel base draw of ``nf_sample(eps=NULL)``."""
    ks = np.arange(first_patch, first_patch + count, dtype=np.uint64)
    return normals(philox_pixels(seed, ks, height * width, STREAM_SAMP)).reshape(count, height, width, 4)
