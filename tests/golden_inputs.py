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


def costvol_launch_inputs(seed=12, B=6, C=32, G=16, h=48, w=160, D=96):
    """reference / source feature maps, depth prior and upstream gradient of the plane-sweep fixture at BASELINE config 2's launch
    shape (tests/golden/costvol_launch.npz holds K, inv_K, pose and the reference's small outputs).  The upstream gradient of the
    (B, D, G, h, w) volume is an outer product of two small random tensors: one rounded float32 multiplication per element."""
    rng = np.random.default_rng(seed)
    ref = bilinear_field(rng, (B, C, h, w), 3, -1.0, 1.0)
    src = bilinear_field(rng, (B, C, h, w), 3, -1.0, 1.0)
    prior = bilinear_field(rng, (B, 1, h, w), 8, 2.0, 22.0)
    a = rng.standard_normal((B, D, G)).astype(np.float32)
    sp = (0.5 + rng.random((B, h, w))).astype(np.float32)
    gout = a[:, :, :, None, None] * sp[:, None, None, :, :]
    return ref, src, prior, gout


def check_costvol_launch(g, vol, d_ref, d_src, rtol=1e-4):
    """vol (B, D, G, h, w), d_ref / d_src (B, C, h, w) as numpy arrays against tests/golden/costvol_launch.npz `g`: plane sums within
    rtol of the planes' absolute sums, lattice values within rtol norm-wise.  Returns the worst ratios for printing."""
    worst = {}
    for name, t, lat in (("vol", vol, vol[:, ::8, ::2, ::8, ::16]), ("d_ref", d_ref, d_ref[..., ::8, ::16]), ("d_src", d_src, d_src[..., ::8, ::16])):
        s = t.astype(np.float64).sum((-1, -2))
        r = float((np.abs(s - g[name + "_sum"]) / np.maximum(g[name + "_abs_sum"], 1e-30)).max())
        want = g[name + "_lattice"].astype(np.float64)
        n = float(np.linalg.norm(lat.astype(np.float64) - want) / np.linalg.norm(want))
        worst[name] = (r, n)
        assert r <= rtol, "%s: a plane sum differs from the reference's by %.2e of the plane's absolute sum" % (name, r)
        assert n <= rtol, "%s: lattice values differ from the reference's by %.2e (norm-wise)" % (name, n)
    return worst


def losses_fullres_inputs(seed=13, B=2, H=192, W=640):
    """colour pyramids of the three frames (scale s = 2^s x 2^s block means of scale 0, float64 sums rounded once) and the four
    disparity maps of the full-resolution photometric fixture (tests/golden/losses_mono_fullres.npz).  -> ({(f, s): image}, {s: disp})"""
    rng = np.random.default_rng(seed)
    colors = {}
    for f in (0, -1, 1):
        base = bilinear_field(rng, (B, 3, H, W), 6).astype(np.float64)
        # a common structure so that the frames resemble each other (a photometric loss of unrelated images has no useful minimum)
        if f == 0:
            common = base
        else:
            base = 0.8 * common + 0.2 * base
        for s in range(4):
            k = 2 ** s
            colors[(f, s)] = np.ascontiguousarray(base.reshape(B, 3, H // k, k, W // k, k).mean((3, 5)), dtype=np.float32)
    disps = {s: (0.004 + 0.1 * bilinear_field(rng, (B, 1, H // 2 ** s, W // 2 ** s), 8).astype(np.float64)).astype(np.float32) for s in range(4)}
    return colors, disps


def mvs_fullres_inputs(seed=14, B=2, H=192, W=640):
    """MVS depth, mono depth and trust mask of the full-resolution MVS / fused-depth loss fixture (tests/golden/losses_mvs_fullres.npz;
    the images are those of losses_fullres_inputs)"""
    rng = np.random.default_rng(seed)
    depth_mvs = bilinear_field(rng, (B, H, W), 8, 2.0, 22.0)
    mono_depth = bilinear_field(rng, (B, 1, H, W), 8, 2.0, 22.0)
    trust = bilinear_field(rng, (B, 1, H, W), 8)
    return depth_mvs, mono_depth, trust


def postvol_launch_inputs(seed=15, B=6, D=96, h=48, w=160):
    """logits of the regulariser, depth prior, and the convex up-sampling's inputs at BASELINE config 2's launch shape
    (tests/golden/postvol_launch.npz holds the reference's small outputs).  numpy's own generators and float64 -> float32 casts only."""
    rng = np.random.default_rng(seed)
    logits = (rng.standard_normal((B, D, h, w)) * 2).astype(np.float32)
    prior = bilinear_field(rng, (B, 1, h, w), 8, 2.0, 22.0)
    g_depth = rng.standard_normal((B, h, w)).astype(np.float32)
    g_ent = rng.standard_normal((B, 1, h, w)).astype(np.float32)
    up_depth = bilinear_field(rng, (B, h, w), 8, 2.0, 22.0)
    up_mask = rng.standard_normal((B, 16 * 9, h, w)).astype(np.float32)
    g_up = rng.standard_normal((B, 4 * h, 4 * w)).astype(np.float32)
    return logits, prior, g_depth, g_ent, up_depth, up_mask, g_up
