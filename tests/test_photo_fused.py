"""The fused photometric chain (md_photo_fwd / md_photo_bwd: warp(s) + SSIM/L1 + min over frames + auto-mask + masked mean,
all scales of one compute_losses call in one launch) against the oracle's restatement of the same chain, op by op:
resize_bilinear + disp_to_depth (reference trainer.py:512-514) -> warp (trainer.py:519-529, layers.py:556-621) -> reproj_loss
(trainer.py:535-550, layers.py:663-677) -> masked_min (trainer.py:687-709 / 630-662 / 583-612), forward and backward.
Tolerance: north_star's 1e-4; masks and arg-min selections must agree exactly on these seeded inputs."""
import numpy as np
import pytest
import torch

from conftest import assert_close, assert_close_knife_edge, relerr
from test_hip_parity import dev, host, kitti_K, rand_pose, smooth_field

pytestmark = pytest.mark.gpu

MIN_D, MAX_D = 0.1, 100.0


@pytest.fixture(scope="module")
def ops():
    from movedepth_amd import _lib, ops as o
    _lib.load()
    return o


def oracle_chain(orc, target, srcs, Ts, K, invK, depth_in, is_disp, ssim_w, no_ssim, ident, noise, ext, mvs_mode, gl):
    """One scale: forward values and the gradients of gl * loss w.r.t. depth_in (disparity level or depth) and every T."""
    B, _, H, W = target.shape
    if is_disp:
        h, w = depth_in.shape[-2:]
        up = orc.resize_bilinear(depth_in, H, W)
        _, depth = orc.disp_to_depth(up, MIN_D, MAX_D)
    else:
        depth = depth_in.reshape(B, 1, H, W)
    warped, pix, rl = [], [], []
    for im, T in zip(srcs, Ts):
        wv, pv = orc.warp(im, depth, K, invK, T)
        warped.append(wv); pix.append(pv)
        rl.append(orc.reproj_loss(wv, target, ssim_w, no_ssim))
    reproj = np.concatenate(rl, 1)
    idn = None if ident is None else np.repeat(ident, reproj.shape[1], 1)   # the min over frames is already taken
    mn, mask, loss = orc.masked_min(reproj, idn, noise, ext, mvs_mode)
    d_reproj = orc.masked_min_bwd(gl, reproj, mask)
    d_depth = np.zeros((B, H, W), np.float32)
    d_T = []
    for f, (im, T) in enumerate(zip(srcs, Ts)):
        d_pred = orc.reproj_loss_bwd(d_reproj[:, f:f + 1], warped[f], target, ssim_w, no_ssim)
        dd, dT = orc.warp_bwd(d_pred, im, depth, K, invK, T)
        d_depth += dd
        d_T.append(dT)
    if is_disp:
        g_up = -d_depth.reshape(B, 1, H, W) * (1 / MIN_D - 1 / MAX_D) * depth * depth
        d_in = orc.resize_bilinear_bwd(g_up.astype(np.float32), h, w)
    else:
        d_in = d_depth.reshape(depth_in.shape)
    return dict(depth=depth, warped=warped, pix=pix, reproj=reproj, mn=mn, mask=mask, loss=loss, d_in=d_in, d_T=d_T)


def make_case(orc, rng, B, H, W, F, S, is_disp):
    """A textured target; source frame f = the target seen from pose T_f^-1 over a fronto-parallel plane at depth 6, plus a
    little noise.  With the test's depths (2..22) the warp re-aligns the frames where the depth is near 6 (reprojection loss <
    identity loss: mask 1) and not elsewhere (mask 0); the per-frame noise decides the arg-min."""
    target = smooth_field(rng, (B, 3, H, W), 3)
    K, invK = kitti_K(H, W, B)
    Ts = [rand_pose(orc, rng, B, 0.004, 0.25) for _ in range(F)]
    plane = np.full((B, 1, H, W), 6.0, np.float32)
    srcs = []
    for T in Ts:
        moved, _ = orc.warp(target, plane, K, invK, np.linalg.inv(T.astype(np.float64)).astype(np.float32))
        srcs.append(np.clip(moved + 0.03 * smooth_field(rng, (B, 3, H, W), 4, -1, 1), 0, 1).astype(np.float32))
    depth = (2 + 20 * smooth_field(rng, (B, 1, H, W), 8)).astype(np.float32)
    if is_disp:
        zs = []
        for s in range(S):
            h, w = max(1, H >> s), max(1, W >> s)
            d_lo = torch.nn.functional.interpolate(torch.from_numpy(depth), size=(h, w), mode="bilinear", align_corners=False).numpy()
            jit = 1 + 0.1 * s * smooth_field(rng, (B, 1, h, w), 3, -1, 1)                      # the scales disagree a little
            zs.append((((1 / (d_lo * jit)) - 1 / MAX_D) / (1 / MIN_D - 1 / MAX_D)).astype(np.float32))  # disp_to_depth inverted
    else:
        zs = [depth * (1 + 0.05 * s) for s in range(S)]
    return target, srcs, Ts, K, invK, zs


def run_fused(ops, target, srcs, Ts, K, invK, zs, gl, T_grad=True, **kw):
    tz = [dev(z, True) for z in zs]
    tT = [dev(T, T_grad) for T in Ts]
    out = ops.photometric_loss(dev(target), [dev(s) for s in srcs], tT, dev(K), dev(invK), tz, min_depth=MIN_D, max_depth=MAX_D, **kw)
    total = sum(float(g) * l for g, l in zip(gl, out["loss"]))
    total.backward()
    return out, tz, tT


@pytest.mark.parametrize("B,H,W,F,S", [(2, 64, 96, 2, 4), (1, 37, 71, 1, 2), (2, 48, 80, 3, 3), (1, 40, 72, 4, 1), (2, 192, 640, 2, 4), (6, 192, 640, 2, 4)])
def test_mono_all_scales_vs_oracle(ops, oracle_lib, B, H, W, F, S):
    """trainer.py:510-532 + 675-709: disparity pyramid, auto-mask against the identity loss with per-scale noise.
    (6, 192, 640, 2, 4) is the launch the bench runs (BASELINE config 2): the kernels' (sample, tile) -> XCD item order depends
    on B."""
    rng = np.random.default_rng(100 + H)
    target, srcs, Ts, K, invK, zs = make_case(oracle_lib, rng, B, H, W, F, S, True)
    ident = np.minimum.reduce([oracle_lib.reproj_loss(s, target) for s in srcs])
    noise = (rng.standard_normal((S, B, 1, H, W)) * 1e-5).astype(np.float32)
    gl = [0.25 * (1 + 0.3 * s) for s in range(S)]
    idn = ops.identity_loss(dev(target), [dev(s) for s in srcs])
    assert_close(host(idn), ident, what="identity loss")
    out, tz, tT = run_fused(ops, target, srcs, Ts, K, invK, zs, gl, is_disp=True, ident_min=idn, noise=dev(noise),
                            want_pix=True, want_mask=True)
    dT_exp = [np.zeros((B, 4, 4), np.float64) for _ in range(F)]
    for s in range(S):
        exp = oracle_chain(oracle_lib, target, srcs, Ts, K, invK, zs[s], True, 0.85, False, ident, noise[s], None, False, gl[s])
        # depth, sample positions and warped frames: the oracle's operations one by one (no contraction) => bit-equal
        assert np.array_equal(host(out["depth"][s]).reshape(exp["depth"].shape), exp["depth"]), "depth[%d] not bit-equal" % s
        for f in range(F):
            nbad = int((host(out["pix"][s][f]) != exp["pix"][f]).sum())
            assert nbad == 0, "scale %d frame %d: %d sample coordinates differ from the oracle's bit patterns" % (s, f, nbad)
            assert np.array_equal(host(out["warped"][s][f]), exp["warped"][f]), "warped[%d][%d] not bit-equal" % (s, f)
            dT_exp[f] += exp["d_T"][f]
        assert_close(host(out["min"][s]), exp["mn"], what="min reprojection loss")
        flips = int((host(out["mask"][s]) != exp["mask"]).sum())
        assert flips == 0, "auto-mask differs at %d pixels" % flips
        share = np.bincount(exp["reproj"].argmin(1).ravel(), minlength=F) / exp["mn"].size
        assert share.min() > (0.03 if F <= 3 else 0.01) and 0.02 < exp["mask"].mean() < 0.98, (share, exp["mask"].mean())   # every branch exercised
        assert abs(float(out["loss"][s].detach()) - exp["loss"]) <= 1e-4 * abs(exp["loss"]), (s, float(out["loss"][s].detach()), exp["loss"])
        # a disparity pixel of level s gathers 4^s full-resolution samples: the share of pixels touched by a sample that sits on
        # a texel boundary (assert_close_knife_edge) grows with the level
        assert_close_knife_edge(host(tz[s].grad), exp["d_in"], rtol=2e-4, max_outlier_frac=2e-3 * 2 ** s, what="d_disp[%d]" % s)
    for f in range(F):
        r = relerr(host(tT[f].grad), dT_exp[f])
        print("fused mono %dx%d d_T[%d] rel %.2e" % (H, W, f, r))
        assert r <= 1e-4, (f, r)   # north_star's bound at every size: the texel decisions are the oracle's (bit-equal pix above)


@pytest.mark.parametrize("with_ext", [False, True])
def test_mvs_mode_vs_oracle(ops, oracle_lib, with_ext):
    """trainer.py:498-509 + 621-662: depth map input, T detached, mask = ones (x photo_conf / dist masks), out-of-view mask."""
    rng = np.random.default_rng(7)
    B, H, W, F = 2, 64, 96, 2
    target, srcs, Ts, K, invK, zs = make_case(oracle_lib, rng, B, H, W, F, 1, False)
    ext = (rng.random((B, 1, H, W)) > 0.3).astype(np.float32) if with_ext else None
    out, tz, tT = run_fused(ops, target, srcs, Ts, K, invK, zs, [1.0], T_grad=False, mvs_mode=True, want_oob=True, want_mask=True,
                            ext_mask=None if ext is None else dev(ext))
    exp = oracle_chain(oracle_lib, target, srcs, Ts, K, invK, zs[0], False, 0.85, False, None, None, ext, True, 1.0)
    for f in range(F):
        assert_close(host(out["warped"][0][f]), exp["warped"][f], what="mvs_color")
        oob = (np.abs(exp["pix"][f]) > 1).any(-1)
        assert int((host(out["oob"][f]).astype(bool) != oob).sum()) == 0
        assert tT[f].grad is None
    assert_close(host(out["min"][0]), exp["mn"])
    assert int((host(out["mask"][0]) != exp["mask"]).sum()) == 0
    assert abs(float(out["loss"][0]) - exp["loss"]) <= 1e-4 * abs(exp["loss"])
    assert_close_knife_edge(host(tz[0].grad), exp["d_in"], rtol=2e-4, what="d_depth_mvs")


@pytest.mark.parametrize("automask", [False, True])
def test_fused_depth_l1_only_vs_oracle(ops, oracle_lib, automask):
    """trainer.py:569-612: ssim_lw = 0 (the value is the L1 term), optional auto-mask against the L1-only identity loss."""
    rng = np.random.default_rng(9)
    B, H, W, F = 2, 64, 96, 2
    target, srcs, Ts, K, invK, zs = make_case(oracle_lib, rng, B, H, W, F, 1, False)
    ident = noise = None
    kw = {}
    if automask:
        ident = np.minimum.reduce([oracle_lib.reproj_loss(s, target, 0.0, True) for s in srcs])
        noise = (rng.standard_normal((1, B, 1, H, W)) * 1e-5).astype(np.float32)
        idn = ops.identity_loss(dev(target), [dev(s) for s in srcs], ssim_w=0.0)
        assert_close(host(idn), ident, rtol=1e-6)
        kw = dict(ident_min=idn, noise=dev(noise))
    out, tz, tT = run_fused(ops, target, srcs, Ts, K, invK, zs, [1.0], T_grad=False, ssim_w=0.0, want_mask=True, **kw)
    exp = oracle_chain(oracle_lib, target, srcs, Ts, K, invK, zs[0], False, 0.0, True, ident, None if noise is None else noise[0],
                       None, False, 1.0)
    assert_close(host(out["min"][0]), exp["mn"], rtol=1e-5)
    assert int((host(out["mask"][0]) != exp["mask"]).sum()) == 0
    assert abs(float(out["loss"][0]) - exp["loss"]) <= 1e-4 * abs(exp["loss"])
    assert_close_knife_edge(host(tz[0].grad), exp["d_in"], rtol=2e-4, what="d_fused_depth")


def test_fused_matches_unfused_kernels_bitwise_where_shared(ops):
    """The fused forward evaluates the warp with the arithmetic of md_warp_fwd: warped images, grids and depths are bit-equal."""
    torch.manual_seed(0)
    B, H, W = 2, 96, 160
    target, s0 = torch.rand(B, 3, H, W, device="cuda"), torch.rand(B, 3, H, W, device="cuda")
    Knp, invKnp = kitti_K(H, W, B)
    K, invK = dev(Knp), dev(invKnp)
    T = torch.eye(4, device="cuda").repeat(B, 1, 1)
    T[:, 0, 3], T[:, 2, 3] = 0.05, 0.03
    disp = 0.01 + 0.3 * torch.rand(B, 1, H // 2, W // 2, device="cuda")
    out = ops.photometric_loss(target, [s0], [T], K, invK, [disp], is_disp=True, want_pix=True)
    depth = ops.disp_to_depth_up(disp, H, W, 0.1, 100.0)
    warped, pix, _ = ops.warp_border(s0, depth, K, invK, T, want_pix=True)
    assert torch.equal(out["depth"][0], depth)
    assert torch.equal(out["pix"][0][0], pix)
    assert torch.equal(out["warped"][0][0], warped)


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1.0, 255.0])
def test_fused_loss_bit_equal_to_the_per_operation_kernels(ops, scale):
    """The fused kernels form their quotients (depth = 1 / sd, cam.xy / z, SSIM n / d) as the steps of the IEEE division without
    its range scaling, two per instruction (md_photo.hpp: md_div_core), and every sum on register pairs.  The per-operation
    kernels (warp.hip, ssim.hip) divide with `/` and sum value by value: the per-pixel loss of one frame must carry the same BITS,
    for images in [0, 1] and in [0, 255]."""
    torch.manual_seed(1)
    B, H, W = 2, 96, 160
    target, s0 = scale * torch.rand(B, 3, H, W, device="cuda"), scale * torch.rand(B, 3, H, W, device="cuda")
    Knp, invKnp = kitti_K(H, W, B)
    K, invK = dev(Knp), dev(invKnp)
    T = torch.eye(4, device="cuda").repeat(B, 1, 1)
    T[:, 0, 3], T[:, 2, 3], T[:, 1, 3] = 0.08, -0.05, 0.01
    disp = 0.01 + 0.5 * torch.rand(B, 1, H // 4, W // 4, device="cuda")
    out = ops.photometric_loss(target, [s0], [T], K, invK, [disp], is_disp=True, want_pix=True)
    depth = ops.disp_to_depth_up(disp, H, W, 0.1, 100.0)
    warped, pix, _ = ops.warp_border(s0, depth, K, invK, T, want_pix=True)
    assert torch.equal(out["depth"][0], depth)
    assert torch.equal(out["pix"][0][0], pix)
    assert torch.equal(out["warped"][0][0], warped)
    per_op = ops.reprojection_loss(warped, target)
    fused = out["min"][0].reshape(per_op.shape)
    assert torch.equal(fused, per_op), "max |diff| %.3e" % (fused - per_op).abs().max().item()


def test_mono_chain_bit_equal_to_the_reference_fixture(ops):
    """tests/golden/losses_mono.npz holds what the reference Trainer's own generate_images_pred (trainer.py:510-532) produced:
    the depth of every pyramid level, the sample grid and the warped frames.  The fused kernel evaluates F.interpolate's taps,
    the three matrix products and grid_sample's interpolation in the operation order those fixtures pin (md_common.hpp,
    md_photo.hpp; tools/diag/op_order_search.py) and reproduces them BIT FOR BIT -- every texel decision is the reference's."""
    from conftest import load_golden

    g = load_golden("losses_mono")
    zs = [dev(g["disp_%d" % s]) for s in range(4)]
    out = ops.photometric_loss(dev(g["in_color_0_0"]), [dev(g["in_color_-1_0"]), dev(g["in_color_1_0"])], [dev(g["T_m1"]), dev(g["T_p1"])],
                               dev(g["in_K_0"]), dev(g["in_inv_K_0"]), zs, is_disp=True, min_depth=MIN_D, max_depth=MAX_D, want_pix=True)
    for s in range(4):
        assert np.array_equal(host(out["depth"][s]).reshape(g["depth_0_%d" % s].shape), g["depth_0_%d" % s]), "depth at scale %d" % s
    for f, tag in enumerate(("m1", "p1")):
        assert np.array_equal(host(out["pix"][0][f]), g["sample_%s_0" % tag]), tag
        assert np.array_equal(host(out["warped"][0][f]), g["color_%s_0" % tag]), tag
        assert np.array_equal(host(out["warped"][3][f]), g["color_%s_3" % tag]), tag
