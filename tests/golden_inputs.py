"""Seeded inputs rebuilt identically by the fixture generators (tools/gen_golden*.py, in the build container) and by the tests (on the
GPU box) for fixtures that are too large to store whole: numpy only, float64 arithmetic made of individually rounded operations,
cast to float32 at the end -- no BLAS / LAPACK / torch kernels whose result could depend on the host's instruction set."""
import numpy as np


def bilinear_field(rng, shape, coarse, lo=0.0, hi=1.0):
    """random coarse grid (H // coarse x W // coarse) up-sampled bilinearly (align_corners) to (..., H, W), values in [lo, hi]"""
    *lead, H, W = shape
    n = int(np.prod(lead)) if lead else 1
    h, w = max(2, H // coarse), max(2, W // coarse)
    c = rng.random((n, h, w))
    ys, xs = np.arange(H, dtype=np.float64) * ((h - 1) / (H - 1)), np.arange(W, dtype=np.float64) * ((w - 1) / (W - 1))
    y0, x0 = np.minimum(np.floor(ys).astype(int), h - 2), np.minimum(np.floor(xs).astype(int), w - 2)
    wy, wx = (ys - y0)[None, :, None], (xs - x0)[None, None, :]
    a, b = c[:, y0][:, :, x0], c[:, y0][:, :, x0 + 1]
    cc, d = c[:, y0 + 1][:, :, x0], c[:, y0 + 1][:, :, x0 + 1]
    out = (a * (1 - wx) + b * wx) * (1 - wy) + (cc * (1 - wx) + d * wx) * wy
    return np.ascontiguousarray((lo + (hi - lo) * out).reshape(*shape), dtype=np.float32)


def warp_fullres_inputs(seed=11, B=2, H=192, W=640):
    """image, depth and upstream gradient of the full-resolution warp fixture (tests/golden/warp_fullres.npz holds K, inv_K, T and the
    reference's outputs for them)"""
    rng = np.random.default_rng(seed)
    img = bilinear_field(rng, (B, 3, H, W), 8)
    depth = bilinear_field(rng, (B, 1, H, W), 16, 2.0, 22.0)
    gout = bilinear_field(rng, (B, 3, H, W), 8, 0.2, 1.0)
    return img, depth, gout
