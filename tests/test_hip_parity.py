"""GPU parity: the HIP path (through the C ABI, via movedepth_amd.ops) against the CPU oracle on seeded inputs
and against the golden fixtures generated from the reference.  Tolerance: north_star's 1e-4 relative fp32
(norm-wise + max-abs bound, conftest.assert_close).  Run with `pytest -m gpu` on an MI355X."""
import numpy as np
import pytest
import torch

from conftest import assert_close, assert_close_knife_edge, load_golden, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from movedepth_amd import _lib, ops

    _lib.load()  # must exist: no fallback
    return ops


def dev(a, requires_grad=False):
    t = torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()
    return t.requires_grad_(requires_grad)


def host(t):
    return t.detach().float().cpu().numpy()


# feature-map layouts of the plane sweep: planar [B,C,h,w] and channels-last [B,h,w,C] (md_costvol_*'s feat_cl; what the 2-D
# encoder produces in torch.channels_last).  The gradients come back in the layout of the features.
FEATS = ["nchw", "nhwc"]


def feat_dev(a, feat, requires_grad=True, dtype=torch.float32):
    t = torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dtype).cuda()
    if feat == "nhwc":
        t = t.contiguous(memory_format=torch.channels_last)
        assert t.shape[1] == 1 or not t.is_contiguous()
    return t.requires_grad_(requires_grad)


def check_grad_layout(t, feat):
    if feat == "nhwc" and t.shape[1] > 1:
        assert t.grad.is_contiguous(memory_format=torch.channels_last), "gradient should come back channels-last"


def kitti_K(h, w, B):
    K = np.array([[0.58 * w, 0, 0.5 * w, 0], [0, 1.92 * h, 0.5 * h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    invK = np.linalg.pinv(K).astype(np.float32)
    return np.repeat(K[None], B, 0), np.repeat(invK[None], B, 0)


def smooth_field(rng, shape, coarse=4, lo=0.0, hi=1.0):
    *lead, H, W = shape
    n = int(np.prod(lead)) if lead else 1
    c = torch.from_numpy(rng.random((n, 1, max(2, H // coarse), max(2, W // coarse)), dtype=np.float32))
    x = torch.nn.functional.interpolate(c, size=(H, W), mode="bilinear", align_corners=True)
    return (lo + (hi - lo) * x).reshape(*shape).numpy().copy()


def rand_pose(oracle, rng, B, rot=0.01, trans=0.05):
    return oracle.transformation_from_parameters(rng.standard_normal((B, 3)).astype(np.float32) * rot,
                                                 rng.standard_normal((B, 3)).astype(np.float32) * trans)


# ------------------------------------------------------------------ schedule
@pytest.mark.parametrize("ty", ["inverse", "linear", "log"])
def test_schedule_golden(ops, ty):
    g = load_golden("schedule")
    D, f = int(g["ndepth"]), float(g["scale_fac"])
    assert_close(host(ops.schedule_depth_range(dev(g["prior"]), D, f, None, ty)), g["v2_" + ty], rtol=1e-5)
    full = ops.schedule_depth_range(dev(g["prior"]), D, f, dev(g["z_trans"]), ty)
    assert_close(host(full), g["zv2_" + ty], rtol=1e-5)
    # the trainer asks for the first / last planes only, as a two-bin schedule (interval positions 0 and 1 whatever D): bit-equal
    # for the inverse and linear spacings; the log spacing's last position is exp(log .1 + (log 10 * (D-1)) / (D-1)), which need
    # not round to the two-bin value -- the trainer takes the full schedule's end planes there (trainer.py process_batch)
    ends = ops.schedule_depth_range(dev(g["prior"]), 2, f, dev(g["z_trans"]), ty)
    assert torch.equal(ends[:, 0], full[:, 0])
    if ty != "log":
        assert torch.equal(ends[:, 1], full[:, -1])
    else:
        assert_close(host(ends[:, 1]), host(full[:, -1]), rtol=1e-6)


# ------------------------------------------------------------------ cost volume
COSTVOL_CASES = ["small", "white", "oob", "zv2", "c64g8"]


@pytest.mark.parametrize("layout,feat", [("bgd", "nchw"), ("bdg", "nchw"), ("ndhwc", "nchw"), ("ndhwc", "nhwc"), ("bdg", "nhwc")])
@pytest.mark.parametrize("tag", COSTVOL_CASES)
def test_costvol_golden(ops, tag, layout, feat):
    g = load_golden("costvol_" + tag)
    G = int(g["G"])
    ref, src = feat_dev(g["ref"], feat), feat_dev(g["src0"], feat)
    vol = ops.costvol_grouped(ref, src, dev(g["K"]), dev(g["invK"]), dev(g["pose"][:, 0]), G,
                              depth_priors=dev(g["hyp"]), layout=layout)
    assert vol.shape == g["grouped0"].shape
    assert_close(host(vol), g["grouped0"], what="grouped volume")
    cor, w = ops.fuse_volumes([vol], layout=layout, exact_single_frame=True)
    assert_close(host(cor), g["cor_feats"], what="cor_feats")
    assert_close(host(w[0]), g["cor_weight0"], rtol=1e-5)
    (cor * dev(g["grad_out"])).sum().backward()
    assert_close(host(ref.grad), g["d_ref"], what="d_ref")
    assert_close(host(src.grad), g["d_src0"], what="d_src")
    # single-frame fast path (fusion skipped): identical to 1e-6
    cor_fast, _ = ops.fuse_volumes([vol.detach()], layout=layout)
    assert relerr(host(cor_fast), g["cor_feats"]) < 1e-4
    # ... and its backward (the identity: the term through the confidence weight it drops is O(1e-8) of the gradient)
    ref2, src2 = feat_dev(g["ref"], feat), feat_dev(g["src0"], feat)
    vol2 = ops.costvol_grouped(ref2, src2, dev(g["K"]), dev(g["invK"]), dev(g["pose"][:, 0]), G, depth_priors=dev(g["hyp"]),
                               layout=layout)
    (ops.fuse_volumes([vol2], layout=layout)[0] * dev(g["grad_out"])).sum().backward()
    assert_close(host(ref2.grad), g["d_ref"], what="d_ref through the single-frame fast path")
    assert_close(host(src2.grad), g["d_src0"], what="d_src through the single-frame fast path")


def test_costvol_ungrouped_golden(ops):
    """G == C reproduces the reference's (B,D,C,h,w) generate_costvol output."""
    g = load_golden("costvol_small")
    C = g["ref"].shape[1]
    vol = ops.costvol_grouped(dev(g["ref"]), dev(g["src0"]), dev(g["K"]), dev(g["invK"]), dev(g["pose"][:, 0]), C,
                              depth_priors=dev(g["hyp"]), layout="bdg")
    assert_close(host(vol), g["cost_vol_full0"])


@pytest.mark.parametrize("layout,feat", [("bgd", "nchw"), ("ndhwc", "nchw"), ("ndhwc", "nhwc")])
def test_costvol_twoframe_golden(ops, layout, feat):
    g = load_golden("costvol_twoframe")
    G = int(g["G"])
    ref = feat_dev(g["ref"], feat)
    srcs = [feat_dev(g["src%d" % f], feat) for f in range(2)]
    vols = [ops.costvol_grouped(ref, srcs[f], dev(g["K"]), dev(g["invK"]), dev(g["pose"][:, f]), G,
                                depth_priors=dev(g["hyp"]), layout=layout) for f in range(2)]
    cor, w = ops.fuse_volumes(vols, layout=layout)
    assert_close(host(cor), g["cor_feats"], what="two-frame cor_feats")
    for f in range(2):
        assert_close(host(w[f]), g["cor_weight%d" % f], rtol=1e-5)
    (cor * dev(g["grad_out"])).sum().backward()
    assert_close(host(ref.grad), g["d_ref"], what="d_ref")
    for f in range(2):
        assert_close(host(srcs[f].grad), g["d_src%d" % f], what="d_src%d" % f)


@pytest.mark.parametrize("case", [
    dict(B=2, C=32, G=16, h=24, w=40, D=12, rot=0.01, trans=0.05),            # ragged tiles (w, h not tile multiples)
    dict(B=1, C=32, G=16, h=48, w=160, D=32, rot=0.01, trans=0.05),           # BASELINE feature size, D slice
    dict(B=2, C=32, G=16, h=16, w=64, D=8, rot=0.3, trans=2.0),               # wild pose: window misses, zero padding
    dict(B=1, C=32, G=32, h=12, w=20, D=5, rot=0.02, trans=0.1),              # ungrouped
    dict(B=1, C=64, G=16, h=12, w=20, D=5, rot=0.02, trans=0.1),              # 4 channels per group
])
@pytest.mark.parametrize("fused,layout,feat", [(False, "bgd", "nchw"), (True, "bgd", "nchw"), (True, "ndhwc", "nchw"),
                                               (True, "ndhwc", "nhwc"), (False, "ndhwc", "nhwc")])
def test_costvol_vs_oracle(ops, oracle_lib, case, fused, layout, feat):
    rng = np.random.default_rng(7)
    B, C, G, h, w, D = (case[k] for k in "BCGhwD")
    ref = smooth_field(rng, (B, C, h, w), 3, -1, 1)
    src = smooth_field(rng, (B, C, h, w), 3, -1, 1)
    K, invK = kitti_K(h, w, B)
    prior = (2 + 20 * rng.random((B, 1, h, w))).astype(np.float32)
    pose = rand_pose(oracle_lib, rng, B, case["rot"], case["trans"])
    z = 30.0 * pose[:, 2, 3]
    hyp = oracle_lib.schedule_depth_range(prior, D, 0.3, z, "inverse")
    gout = rng.standard_normal((B, D, G, h, w)).astype(np.float32)
    exp = oracle_lib.costvol_grouped(ref, src, K, invK, hyp, pose, G)
    exp_dref, exp_dsrc = oracle_lib.costvol_grouped_bwd(gout, ref, src, K, invK, hyp, pose)
    r, s = feat_dev(ref, feat), feat_dev(src, feat)
    if fused:  # schedule evaluated inside the kernel
        vol = ops.costvol_grouped(r, s, dev(K), dev(invK), dev(pose), G, prior=dev(prior), ndepth=D, scale_fac=0.3,
                                  z_trans=dev(z), type="inverse", layout=layout)
    else:
        vol = ops.costvol_grouped(r, s, dev(K), dev(invK), dev(pose), G, depth_priors=dev(hyp), layout=layout)
    assert_close(host(vol), exp, what="volume")
    (vol * dev(gout)).sum().backward()
    assert_close(host(r.grad), exp_dref, what="d_ref")
    assert_close(host(s.grad), exp_dsrc, what="d_src")
    if layout == "ndhwc" and C // G in (1, 2, 4) and G in (8, 16):
        check_grad_layout(r, feat)


def relerr_chunked(a, b, chunk=1 << 24):
    """conftest.relerr for arrays too large to hold twice in float64"""
    a, b = np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)
    num = den = 0.0
    mx = 0.0
    for i in range(0, a.size, chunk):
        x, y = a[i:i + chunk].astype(np.float64), b[i:i + chunk].astype(np.float64)
        num += float(((x - y) ** 2).sum())
        den += float((y ** 2).sum())
        mx = max(mx, float(np.abs(x - y).max()))
    return (num / (den if den > 0 else 1.0)) ** 0.5, mx


def full_size_case(oracle, rng, B, C, G, h, w, D, prior_kind, rot=0.01, trans=0.05, dtype=None):
    """Seeded inputs of a whole launch + the oracle's volume and gradients (layers.py:778-794, trainer.py:351-363 and
    their autograd), velocity-guided hypotheses (layers.py:370-398).  dtype: features / gradient rounded to it first."""
    def rounded(a):
        return a if dtype is None else torch.from_numpy(a).to(dtype).float().numpy()
    ref = rounded(smooth_field(rng, (B, C, h, w), 3, -1, 1))
    src = rounded(smooth_field(rng, (B, C, h, w), 3, -1, 1))
    K, invK = kitti_K(h, w, B)
    if prior_kind == "smooth":
        prior = (2 + 20 * smooth_field(rng, (B, 1, h, w), 8, 0, 1)).astype(np.float32)
    else:  # every pixel its own range
        prior = (2 + 20 * rng.random((B, 1, h, w))).astype(np.float32)
    pose = rand_pose(oracle, rng, B, rot, trans)
    z = (30.0 * pose[:, 2, 3]).astype(np.float32)
    hyp = oracle.schedule_depth_range(prior, D, 0.3, z, "inverse")
    gout = rounded(rng.standard_normal((B, D, G, h, w)).astype(np.float32))
    exp = oracle.costvol_grouped(ref, src, K, invK, hyp, pose, G)
    exp_dref, exp_dsrc = oracle.costvol_grouped_bwd(gout, ref, src, K, invK, hyp, pose)
    return dict(ref=ref, src=src, K=K, invK=invK, prior=prior, pose=pose, z=z, gout=gout, exp=exp, exp_dref=exp_dref,
                exp_dsrc=exp_dsrc)


@pytest.mark.parametrize("feat", FEATS)
@pytest.mark.parametrize("prior_kind", ["smooth", "white"])
def test_costvol_config2_launch_shape_vs_oracle(ops, oracle_lib, prior_kind, feat):
    """The launch the bench and the trainer run -- BASELINE config 2: B=6, C=32, G=16, 48x160, D=96, schedule fused,
    channels-last volume, fp32 -- volume, d_ref and d_src against the oracle at 1e-4 (reference layers.py:778-794,
    trainer.py:351-363).  360 workgroup items cut into hypothesis slices: the k-slicing / sub-slice boundaries of this exact
    decomposition are what the small oracle cases cannot reach."""
    c = full_size_case(oracle_lib, np.random.default_rng(21), 6, 32, 16, 48, 160, 96, prior_kind)
    r, s = feat_dev(c["ref"], feat), feat_dev(c["src"], feat)
    vol = ops.costvol_grouped(r, s, dev(c["K"]), dev(c["invK"]), dev(c["pose"]), 16, prior=dev(c["prior"]), ndepth=96,
                              scale_fac=0.3, z_trans=dev(c["z"]), type="inverse", layout="ndhwc")
    assert vol.is_contiguous() is False and vol.permute(0, 1, 3, 4, 2).is_contiguous()     # (B,D,h,w,G) storage
    rel, mx = relerr_chunked(host(vol), c["exp"])
    print("config-2 launch shape (%s prior): volume rel %.2e max-abs %.2e" % (prior_kind, rel, mx))
    assert rel <= 1e-4 and mx <= 1e-3 * float(np.abs(c["exp"]).max())
    vol.backward(dev(c["gout"]))
    assert_close(host(r.grad), c["exp_dref"], what="d_ref")
    assert_close(host(s.grad), c["exp_dsrc"], what="d_src")
    check_grad_layout(r, feat)
    check_grad_layout(s, feat)


@pytest.mark.parametrize("feat", FEATS)
@pytest.mark.parametrize("case", ["driving_1m", "driving_2m", "moderate"])
def test_costvol_parallax_cases_launch_shape_vs_oracle(ops, oracle_lib, case, feat):
    """Config 2's launch shape with the parallax BASELINE's own synthetic case does not have: a driving scene
    (movedepth_amd/synthetic.driving_scene: ground plane 6-80 m + facades, +-1 m / +-2 m per frame along the optical axis, small
    yaw: t_z / depth up to 0.2-0.4) and 'moderate' poses (axis-angle N(0, 0.05^2), translation N(0, 0.3^2) against a steep smooth
    prior of 2-22 m).  These are the cases tools/bench_costvol.py and bench.py's roofline.parallax_cases time; they take the paths
    the sane case never does: several windows per tile in three shapes, sub-slices gathered from L2 with the queued d_src terms,
    tiles whose sweep leaves the image below (the bounding box of a pixel with one end outside the image).  WHITE-NOISE features
    and gradient: a tap taken from a neighbouring cell shows.  Reference: layers.py:778-794, trainer.py:351-363."""
    from movedepth_amd.synthetic import driving_scene
    rng = np.random.default_rng(77)
    B, C, G, h, w, D = 6, 32, 16, 48, 160, 96
    ref = rng.standard_normal((B, C, h, w)).astype(np.float32)
    src = rng.standard_normal((B, C, h, w)).astype(np.float32)
    K, invK = kitti_K(h, w, B)
    if case == "moderate":
        prior = (2 + 20 * smooth_field(rng, (B, 1, h, w), 12, 0, 1)).astype(np.float32)
        pose = rand_pose(oracle_lib, rng, B, 0.05, 0.3)
    else:
        prior, pose = driving_scene(B, h, w, speed=1.0 if case == "driving_1m" else 2.0)
    hyp = oracle_lib.schedule_depth_range(prior, D, 0.3, None, "inverse")
    gout = rng.standard_normal((B, D, G, h, w)).astype(np.float32)
    exp = oracle_lib.costvol_grouped(ref, src, K, invK, hyp, pose, G)
    exp_dref, exp_dsrc = oracle_lib.costvol_grouped_bwd(gout, ref, src, K, invK, hyp, pose)
    r, s = feat_dev(ref, feat), feat_dev(src, feat)
    vol = ops.costvol_grouped(r, s, dev(K), dev(invK), dev(pose), G, prior=dev(prior), ndepth=D, scale_fac=0.3, type="inverse",
                              layout="ndhwc")
    rel, mx = relerr_chunked(host(vol), exp)
    print("parallax case %s: volume rel %.2e max-abs %.2e" % (case, rel, mx))
    assert rel <= 1e-4 and mx <= 1e-3 * float(np.abs(exp).max())
    vol.backward(dev(gout))
    assert_close(host(r.grad), exp_dref, what="d_ref")
    assert_close(host(s.grad), exp_dsrc, what="d_src")


@pytest.mark.parametrize("feat", FEATS)
def test_costvol_config2_launch_shape_reference_fixture(ops, feat):
    """The same launch against the REFERENCE's own generate_costvol + group mean + autograd (tests/golden/costvol_launch.npz from
    tools/gen_golden.py gen_costvol_launch: plane sums and a lattice of the volume and of both feature gradients; the large inputs
    are rebuilt from a seed by tests/golden_inputs.py).  Reference: layers.py:778-794, trainer.py:351-363."""
    from golden_inputs import check_costvol_launch, costvol_launch_inputs
    g = load_golden("costvol_launch")
    ref, src, prior, gout = costvol_launch_inputs()
    r, s = feat_dev(ref, feat), feat_dev(src, feat)
    vol = ops.costvol_grouped(r, s, dev(g["K"]), dev(g["invK"]), dev(g["pose"][:, 0]), int(g["G"]), prior=dev(prior), ndepth=96,
                              scale_fac=0.3, z_trans=None, type="inverse", layout="ndhwc")
    vol.backward(dev(gout))
    print("config-2 launch shape vs the reference's fixture (plane-sum ratio, lattice rel):", check_costvol_launch(g, host(vol), host(r.grad), host(s.grad)))


@pytest.mark.parametrize("feat", FEATS)
@pytest.mark.parametrize("C,B,dtype", [(32, 6, torch.bfloat16), (64, 3, torch.bfloat16), (64, 4, torch.float32)])
def test_costvol_config4_launch_shapes_vs_oracle(ops, oracle_lib, C, B, dtype, feat):
    """BASELINE config 4's volume (80x256 features, D=128) against the oracle at the batch sizes whose launches take the
    schedules the small cases do not: (C=32, B=6, bf16) is the shape with more hypothesis slices than resident workgroup
    slots (two-phase schedule); C=64 -> G=16 is the 4-channels-per-group instantiation, again with more slices than slots
    (in bf16 and, at 48x160 / D=96, in fp32)."""
    fp32 = dtype == torch.float32
    h, w, D = (48, 160, 96) if fp32 else (80, 256, 128)
    c = full_size_case(oracle_lib, np.random.default_rng(22 + C + B), B, C, 16, h, w, D, "smooth", dtype=None if fp32 else dtype)
    r, s = feat_dev(c["ref"], feat, dtype=dtype), feat_dev(c["src"], feat, dtype=dtype)
    vol = ops.costvol_grouped(r, s, dev(c["K"]), dev(c["invK"]), dev(c["pose"]), 16, prior=dev(c["prior"]), ndepth=D,
                              scale_fac=0.3, z_trans=dev(c["z"]), type="inverse", layout="ndhwc")
    assert vol.dtype == dtype
    got = host(vol)
    rel, mx = relerr_chunked(got, c["exp"])
    print("config-4 launch shape C=%d B=%d %s: volume rel %.2e" % (C, B, dtype, rel))
    if fp32:
        assert rel <= 1e-4
    else:
        rel_r, _ = relerr_chunked(got, torch.from_numpy(c["exp"]).to(dtype).float().numpy())
        assert rel_r <= 1.5e-3 and rel <= 4e-3, (rel_r, rel)      # bf16 output rounding (2^-9 relative per element)
    del got
    vol.backward(torch.from_numpy(c["gout"]).to(dtype).cuda())
    tol = 1e-4 if fp32 else 4e-3
    assert relerr(host(r.grad), c["exp_dref"]) <= tol
    assert relerr(host(s.grad), c["exp_dsrc"]) <= tol


def _half_launch_check(ops, r, s, c_or, exp, exp_dref, exp_dsrc, gout, run, what, dtype=torch.float16, tol_rounded=2e-4, tol_exact=5e-4):
    """Volume against the oracle's fp32 volume rounded to `dtype` (norm-wise `tol_rounded`) and against the exact one (`tol_exact`),
    gradients (fp32 sums of rounded records, returned in `dtype`) against the oracle's: the bounds of test_costvol_half_io_vs_oracle."""
    vol = run(r, s)
    assert vol.dtype == dtype
    got = host(vol)
    rel_r, _ = relerr_chunked(got, torch.from_numpy(exp).to(dtype).float().numpy())
    rel, _ = relerr_chunked(got, exp)
    print("%s: volume rel %.2e against the rounded oracle, %.2e against the exact one" % (what, rel_r, rel))
    assert rel_r <= tol_rounded and rel <= tol_exact, (rel_r, rel)
    del got
    vol.backward(torch.from_numpy(gout).to(dtype).cuda())
    assert r.grad.dtype == dtype and s.grad.dtype == dtype
    e_r, e_s = relerr(host(r.grad), exp_dref), relerr(host(s.grad), exp_dsrc)
    print("%s: d_ref rel %.2e, d_src rel %.2e" % (what, e_r, e_s))
    assert e_r <= tol_exact and e_s <= tol_exact, (e_r, e_s)


@pytest.mark.parametrize("feat", FEATS)
@pytest.mark.parametrize("prior_kind", ["smooth", "white"])
def test_costvol_config5_launch_shape_vs_oracle(ops, oracle_lib, prior_kind, feat):
    """BASELINE config 5's volume launch -- B=6, C=32, G=16, 48x160, D=96, fp16 feature maps and volume, velocity-guided hypotheses
    (layers.py:370-398 with z = z_scale * T[2,3], trainer.py:337-341), schedule fused, channels-last volume -- against the oracle at
    the launch shape bench.py's config-5 line times: the 720-item decomposition, its hypothesis slices and
    sub-slices in the 2-byte build (8-byte records, the raw prefetch queue of the backward).  Both feature layouts, smooth and
    white-noise priors.  The oracle gets the ROUNDED features / gradient as floats; tolerances of test_costvol_half_io_vs_oracle:
    2e-4 norm-wise against the oracle's volume rounded to fp16, 5e-4 against the exact one and for both gradients.
    Reference: layers.py:778-794, trainer.py:349-363 (one lookup frame's volume; the N-frame fusion is a10 / test_fuse_*)."""
    c = full_size_case(oracle_lib, np.random.default_rng(23), 6, 32, 16, 48, 160, 96, prior_kind, dtype=torch.float16)
    r, s = feat_dev(c["ref"], feat, dtype=torch.float16), feat_dev(c["src"], feat, dtype=torch.float16)

    def run(r, s):
        return ops.costvol_grouped(r, s, dev(c["K"]), dev(c["invK"]), dev(c["pose"]), 16, prior=dev(c["prior"]), ndepth=96,
                                   scale_fac=0.3, z_trans=dev(c["z"]), type="inverse", layout="ndhwc")
    _half_launch_check(ops, r, s, c, c["exp"], c["exp_dref"], c["exp_dsrc"], c["gout"], run, "config-5 launch shape (%s prior, %s)" % (prior_kind, feat))
    check_grad_layout(r, feat)
    check_grad_layout(s, feat)


@pytest.mark.parametrize("feat", FEATS)
@pytest.mark.parametrize("case", ["driving_1m", "driving_2m", "moderate"])
def test_costvol_parallax_cases_launch_shape_fp16_vs_oracle(ops, oracle_lib, case, feat):
    """The three parallax cases of test_costvol_parallax_cases_launch_shape_vs_oracle in fp16 (config 5's element type): the 2-byte
    build's windows in three shapes, gathered sub-slices with queued d_src terms, tiles whose sweep leaves the image.  White-noise
    features and gradient, rounded to fp16 before the oracle sees them.  Reference: layers.py:778-794, trainer.py:351-363."""
    from movedepth_amd.synthetic import driving_scene
    rng = np.random.default_rng(78)
    B, C, G, h, w, D = 6, 32, 16, 48, 160, 96
    rnd = lambda a: torch.from_numpy(a).to(torch.float16).float().numpy()
    ref = rnd(rng.standard_normal((B, C, h, w)).astype(np.float32))
    src = rnd(rng.standard_normal((B, C, h, w)).astype(np.float32))
    K, invK = kitti_K(h, w, B)
    if case == "moderate":
        prior = (2 + 20 * smooth_field(rng, (B, 1, h, w), 12, 0, 1)).astype(np.float32)
        pose = rand_pose(oracle_lib, rng, B, 0.05, 0.3)
    else:
        prior, pose = driving_scene(B, h, w, speed=1.0 if case == "driving_1m" else 2.0)
    hyp = oracle_lib.schedule_depth_range(prior, D, 0.3, None, "inverse")
    gout = rnd(rng.standard_normal((B, D, G, h, w)).astype(np.float32))
    exp = oracle_lib.costvol_grouped(ref, src, K, invK, hyp, pose, G)
    exp_dref, exp_dsrc = oracle_lib.costvol_grouped_bwd(gout, ref, src, K, invK, hyp, pose)
    r, s = feat_dev(ref, feat, dtype=torch.float16), feat_dev(src, feat, dtype=torch.float16)

    def run(r, s):
        return ops.costvol_grouped(r, s, dev(K), dev(invK), dev(pose), G, prior=dev(prior), ndepth=D, scale_fac=0.3, type="inverse",
                                   layout="ndhwc")
    _half_launch_check(ops, r, s, None, exp, exp_dref, exp_dsrc, gout, run, "fp16 parallax case %s (%s)" % (case, feat))


@pytest.mark.parametrize("feat", FEATS)
@pytest.mark.parametrize("w,h", [(256, 80), (512, 160)])
def test_costvol_wide_coordinates_white_noise(ops, oracle_lib, w, h, feat):
    """The channels-last kernels snap a coordinate within 2^-17 + 5e-7*|coord| px below an integer to that integer
    (csrc/costvol_cl.inc); the band grows with the coordinate.  White-noise features (no smoothness to hide a wrong cell)
    at config 4's widths: 256 (prior_scale 2 of a 1024-px frame) and 512 (prior_scale 1), static and moving camera."""
    rng = np.random.default_rng(31)
    B, C, G, D = 2, 32, 16, 16
    ref = rng.standard_normal((B, C, h, w)).astype(np.float32)
    src = rng.standard_normal((B, C, h, w)).astype(np.float32)
    K, invK = kitti_K(h, w, B)
    prior = (2 + 20 * rng.random((B, 1, h, w))).astype(np.float32)
    pose = rand_pose(oracle_lib, rng, B, 0.01, 0.05)
    pose[1] = np.eye(4, dtype=np.float32)               # static camera: every coordinate sits on an integer
    hyp = oracle_lib.schedule_depth_range(prior, D, 0.3, None, "inverse")
    gout = rng.standard_normal((B, D, G, h, w)).astype(np.float32)
    exp = oracle_lib.costvol_grouped(ref, src, K, invK, hyp, pose, G)
    exp_dref, exp_dsrc = oracle_lib.costvol_grouped_bwd(gout, ref, src, K, invK, hyp, pose)
    r, s = feat_dev(ref, feat), feat_dev(src, feat)
    vol = ops.costvol_grouped(r, s, dev(K), dev(invK), dev(pose), G, prior=dev(prior), ndepth=D, scale_fac=0.3,
                              type="inverse", layout="ndhwc")
    assert_close(host(vol), exp, what="volume")
    vol.backward(dev(gout))
    assert_close(host(r.grad), exp_dref, what="d_ref")
    assert_close(host(s.grad), exp_dsrc, what="d_src")


@pytest.mark.parametrize("feat", FEATS)
@pytest.mark.parametrize("C", [32, 16, 64])
def test_costvol_near_prior_carried_sums_vs_oracle(ops, oracle_lib, C, feat):
    """A near scene (prior 0.4-1.2 m, translation 5 cm): the sample position crosses a source cell every few hypotheses, to
    the right for one sample and to the left for the other (a third moves up / down / diagonally).  The backward keeps the
    pending sums of the tap column that the next cell still covers instead of flushing them (csrc/costvol_cl.inc flush_col);
    white-noise features and gradients so that a sum carried into the wrong column shows.  1, 2 and 4 channels per group."""
    rng = np.random.default_rng(91)
    B, G, h, w, D = 3, 16, 24, 80, 48
    ref = rng.standard_normal((B, C, h, w)).astype(np.float32)
    src = rng.standard_normal((B, C, h, w)).astype(np.float32)
    K, invK = kitti_K(h, w, B)
    prior = (0.4 + 0.8 * smooth_field(rng, (B, 1, h, w), 6, 0, 1)).astype(np.float32)
    pose = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    pose[0, 0, 3], pose[1, 0, 3] = 0.05, -0.05
    pose[2, 0, 3], pose[2, 1, 3], pose[2, 2, 3] = 0.02, 0.03, 0.02
    hyp = oracle_lib.schedule_depth_range(prior, D, 0.3, None, "inverse")
    gout = rng.standard_normal((B, D, G, h, w)).astype(np.float32)
    exp = oracle_lib.costvol_grouped(ref, src, K, invK, hyp, pose, G)
    exp_dref, exp_dsrc = oracle_lib.costvol_grouped_bwd(gout, ref, src, K, invK, hyp, pose)
    r, s = feat_dev(ref, feat), feat_dev(src, feat)
    vol = ops.costvol_grouped(r, s, dev(K), dev(invK), dev(pose), G, prior=dev(prior), ndepth=D, scale_fac=0.3, layout="ndhwc")
    assert_close(host(vol), exp, what="volume")
    vol.backward(dev(gout))
    assert_close(host(r.grad), exp_dref, what="d_ref")
    assert_close(host(s.grad), exp_dsrc, what="d_src")


@pytest.mark.parametrize("feat", FEATS)
def test_costvol_wild_poses_backward_fallback_vs_oracle(ops, oracle_lib, feat):
    """Poses of an untrained pose network (axis-angle ~ N(0, 0.3^2) rad, translation ~ N(0, 2^2)); one sample of the batch keeps a
    sane pose.  One launch in either feature layout: channels-last features switch the wild sub-slices to L2 gathers with queued
    d_src terms inside the kernel, planar features take the kernel's per-tap miss path.  Volume and both gradients against the
    oracle (reference layers.py:778-794, zeros padding: no cliff in grid_sample); the launch's census word reports the gathered
    share the trainer's table policy reads (ops.GatherTablePolicy)."""
    rng = np.random.default_rng(77)
    B, C, G, h, w, D = 3, 32, 16, 48, 160, 32
    ref = smooth_field(rng, (B, C, h, w), 3, -1, 1)
    src = smooth_field(rng, (B, C, h, w), 3, -1, 1)
    K, invK = kitti_K(h, w, B)
    prior = (2 + 20 * smooth_field(rng, (B, 1, h, w), 8, 0, 1)).astype(np.float32)
    pose = rand_pose(oracle_lib, rng, B, 0.3, 2.0)
    pose[1] = rand_pose(oracle_lib, rng, 1, 0.01, 0.05)[0]
    hyp = oracle_lib.schedule_depth_range(prior, D, 0.3, None, "inverse")
    gout = rng.standard_normal((B, D, G, h, w)).astype(np.float32)
    exp = oracle_lib.costvol_grouped(ref, src, K, invK, hyp, pose, G)
    exp_dref, exp_dsrc = oracle_lib.costvol_grouped_bwd(gout, ref, src, K, invK, hyp, pose)
    ops.enable_library_kernel_timing(True)
    try:
        r, s = feat_dev(ref, feat), feat_dev(src, feat)
        vol = ops.costvol_grouped(r, s, dev(K), dev(invK), dev(pose), G, prior=dev(prior), ndepth=D, scale_fac=0.3, layout="ndhwc")
        assert_close(host(vol), exp, what="volume")
        vol.backward(dev(gout))
        torch.cuda.synchronize()
        t = ops.library_kernel_times_us(["md_costvol_bwd"])
    finally:
        ops.enable_library_kernel_timing(False)
    assert t["md_costvol_bwd"]["launches"] == 1, t
    pol = ops.backward_policy()
    torch.cuda.synchronize()
    gathered, total, windows, segments = pol.last_census()
    assert abs(total - B * (h // 4) * (w // 16) * D) <= 2 * segments and segments >= B * (h // 4) * (w // 16), (gathered, total, windows, segments)   # every step of every 16 x 4 tile, once (counted in units of 4 per workgroup)
    if feat == "nhwc":
        assert 0.1 < pol.gathered_share() < 0.9, pol.gathered_share()   # two of three samples wild (D = 32: short slices fit more often)
        c = pol.costs()
        assert c is not None and len(c) == B * (h // 4) * (w // 16) and (c > 0).all()   # per-item cycle counts of the launch
    else:
        assert gathered == 0                                             # planar features never gather
    assert_close(host(r.grad), exp_dref, what="d_ref")
    assert_close(host(s.grad), exp_dsrc, what="d_src")


@pytest.mark.parametrize("case", ["wild", "moderate", "driving_2m"])
def test_costvol_backward_cell_table_vs_oracle(ops, oracle_lib, case):
    """MD_CV_GATHER_TABLE (md_costvol_bwd's `flags`, pinned here through the policy object): the backward instantiation whose gather mode merges a tile's d_src terms per source cell in LDS
    (csrc/costvol_cl.inc, TAB: cell-keyed slots in the idle d_src window, a taken slot falls back to the queue, the table leaves the CU
    every 32 steps) at config 2's launch shape, channels-last features: wild poses (every slice gathered, slots contended), moderate
    poses and the driving scene at 2 m per frame (windows and gathered ranges in one workgroup).  White-noise features and gradient.
    Reference: autograd of layers.py:791 (zeros padding, per-tap drop)."""
    from movedepth_amd.synthetic import driving_scene
    rng = np.random.default_rng(79)
    B, C, G, h, w, D = 6, 32, 16, 48, 160, 96
    ref = rng.standard_normal((B, C, h, w)).astype(np.float32)
    src = rng.standard_normal((B, C, h, w)).astype(np.float32)
    K, invK = kitti_K(h, w, B)
    if case == "driving_2m":
        prior, pose = driving_scene(B, h, w, speed=2.0)
    else:
        prior = (2 + 20 * smooth_field(rng, (B, 1, h, w), 12, 0, 1)).astype(np.float32)
        pose = rand_pose(oracle_lib, rng, B, 0.3, 2.0) if case == "wild" else rand_pose(oracle_lib, rng, B, 0.05, 0.3)
    hyp = oracle_lib.schedule_depth_range(prior, D, 0.3, None, "inverse")
    gout = rng.standard_normal((B, D, G, h, w)).astype(np.float32)
    exp_dref, exp_dsrc = oracle_lib.costvol_grouped_bwd(gout, ref, src, K, invK, hyp, pose)
    pol = ops.backward_policy()
    pol.reset()
    pol.force_table = True
    ops.enable_library_kernel_timing(True)
    try:
        r, s = feat_dev(ref, "nhwc"), feat_dev(src, "nhwc")
        vol = ops.costvol_grouped(r, s, dev(K), dev(invK), dev(pose), G, prior=dev(prior), ndepth=D, scale_fac=0.3, type="inverse",
                                  layout="ndhwc")
        vol.backward(dev(gout))
        torch.cuda.synchronize()
        t = ops.library_kernel_times_us(["md_costvol_bwd"])
    finally:
        ops.enable_library_kernel_timing(False)
        pol.force_table = None
    assert t["md_costvol_bwd"]["launches"] == 1, t
    assert_close(host(r.grad), exp_dref, what="d_ref")
    assert_close(host(s.grad), exp_dsrc, what="d_src")
    share = pol.gathered_share()
    assert (share > 0.45) == (case == "wild"), (case, share)   # the threshold of the automatic choice separates the three regimes


def test_costvol_gather_table_policy_follows_the_poses(ops, oracle_lib):
    """The trainer's automatic choice (ops.GatherTablePolicy; no environment variable, no host synchronisation): the backward of
    a wild-pose volume leaves a census above the threshold, so the NEXT backward runs the cell-table build; a sane-pose launch
    brings it back.  Gradients of every launch against the oracle (autograd of layers.py:784-792)."""
    rng = np.random.default_rng(83)
    B, C, G, h, w, D = 2, 32, 16, 48, 160, 96     # (whole-depth slices, as at config 2's shape: short ones fit their windows more often)
    ref = rng.standard_normal((B, C, h, w)).astype(np.float32)
    src = rng.standard_normal((B, C, h, w)).astype(np.float32)
    K, invK = kitti_K(h, w, B)
    prior = (2 + 20 * smooth_field(rng, (B, 1, h, w), 12, 0, 1)).astype(np.float32)
    hyp = oracle_lib.schedule_depth_range(prior, D, 0.3, None, "inverse")
    gout = rng.standard_normal((B, D, G, h, w)).astype(np.float32)
    poses = {"wild": rand_pose(oracle_lib, rng, B, 0.3, 2.0), "sane": rand_pose(oracle_lib, rng, B, 0.01, 0.05)}
    exp = {k: oracle_lib.costvol_grouped_bwd(gout, ref, src, K, invK, hyp, p) for k, p in poses.items()}
    pol = ops.backward_policy()
    pol.reset()
    pol.force_balance = False      # (the partition is test_costvol_backward_balanced_partition_vs_oracle's subject)
    keep_threshold, pol.threshold = pol.threshold, 0.15   # B = 2: three 32-step slices per item fit their windows more often than config 2's whole-depth slices
    seen = []
    for case in ("wild", "wild", "sane", "sane"):
        r, s = feat_dev(ref, "nhwc"), feat_dev(src, "nhwc")
        vol = ops.costvol_grouped(r, s, dev(K), dev(invK), dev(poses[case]), G, prior=dev(prior), ndepth=D, scale_fac=0.3, layout="ndhwc")
        before = pol.table_launches
        vol.backward(dev(gout))
        seen.append(pol.table_launches - before)
        torch.cuda.synchronize()          # (the test waits so that the census has landed; a trainer simply reads it a launch later)
        print("gather-table policy: %s poses, gathered share %.3f, table build used: %d" % (case, pol.gathered_share(), seen[-1]))
        assert_close(host(r.grad), exp[case][0], what="d_ref " + case)
        assert_close(host(s.grad), exp[case][1], what="d_src " + case)
    pol.threshold = keep_threshold
    pol.reset()
    assert seen == [0, 1, 1, 0], seen


def test_costvol_forward_fine_slices_follow_the_backward_census(ops, oracle_lib):
    """md_costvol_fwd's MD_CV_FINE_SLICES (ABI 17) and its automatic choice (ops.BackwardPolicy.forward_flags): the backward of a
    moderate-pose volume stages more than one window per segment, so the NEXT forward of that shape runs twice the slices per item;
    after a sane-pose backward it does not.  The volume of every launch against the oracle (layers.py:778-794), white-noise features."""
    rng = np.random.default_rng(87)
    B, C, G, h, w, D = 6, 32, 16, 48, 160, 96
    ref = rng.standard_normal((B, C, h, w)).astype(np.float32)
    src = rng.standard_normal((B, C, h, w)).astype(np.float32)
    K, invK = kitti_K(h, w, B)
    prior = (2 + 20 * smooth_field(rng, (B, 1, h, w), 12, 0, 1)).astype(np.float32)
    hyp = oracle_lib.schedule_depth_range(prior, D, 0.3, None, "inverse")
    poses = {"moderate": rand_pose(oracle_lib, rng, B, 0.05, 0.3), "sane": rand_pose(oracle_lib, rng, B, 0.01, 0.05)}
    exp = {k: oracle_lib.costvol_grouped(ref, src, K, invK, hyp, p, G) for k, p in poses.items()}
    pol = ops.backward_policy()
    pol.reset()
    pol.force_balance = False
    key = (B, C, G, h, w, D, True, True)
    try:
        seen = []
        for case in ("moderate", "moderate", "sane", "sane"):
            r, s = feat_dev(ref, "nhwc"), feat_dev(src, "nhwc")
            before = pol.fine_launches
            vol = ops.costvol_grouped(r, s, dev(K), dev(invK), dev(poses[case]), G, prior=dev(prior), ndepth=D, scale_fac=0.3, layout="ndhwc")
            seen.append(pol.fine_launches - before)
            rel, mx = relerr_chunked(host(vol), exp[case])
            assert rel <= 1e-4, (case, rel)
            vol.backward(torch.ones_like(vol))
            torch.cuda.synchronize()
            _, _, wnd, seg = pol.last_census()
            print("%s poses: forward on fine slices %d; backward staged %d windows for %d segments" % (case, seen[-1], wnd, seg))
            assert seg == B * (h // 4) * (w // 16) and (wnd == seg if case == "sane" else wnd != seg)
        assert seen == [0, 1, 1, 0], seen
        pol.force_fine = True                       # the flag itself, on the sane launch too
        r, s = feat_dev(ref, "nhwc"), feat_dev(src, "nhwc")
        vol = ops.costvol_grouped(r, s, dev(K), dev(invK), dev(poses["sane"]), G, prior=dev(prior), ndepth=D, scale_fac=0.3, layout="ndhwc")
        assert relerr_chunked(host(vol), exp["sane"])[0] <= 1e-4
    finally:
        pol.reset()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("case", ["driving_2m", "moderate", "wild"])
def test_costvol_backward_balanced_partition_vs_oracle(ops, oracle_lib, case, dtype):
    """md_costvol_bwd's `shares` / `cost` (ABI 17; ops.BackwardPolicy): the first launch records the shader cycles per work item, the
    second runs on the partition the policy computes from them -- equal COST per workgroup, shares that cut items at multiples of 8
    steps and span item boundaries, d_ref accumulated with atomics -- at config 2's launch shape on the cases whose tiles differ in
    cost (driving scene, moderate poses; wild poses: together with the cell table).  Checked: the partition covers every step exactly
    once, both launches' gradients against the oracle (autograd of layers.py:784-792), white-noise features and gradient."""
    from movedepth_amd.synthetic import driving_scene
    rng = np.random.default_rng(85)
    B, C, G, h, w, D = 6, 32, 16, 48, 160, 96
    half = dtype != torch.float32
    rnd = (lambda a: torch.from_numpy(a).to(dtype).float().numpy()) if half else (lambda a: a)
    ref = rnd(rng.standard_normal((B, C, h, w)).astype(np.float32))
    src = rnd(rng.standard_normal((B, C, h, w)).astype(np.float32))
    K, invK = kitti_K(h, w, B)
    if case == "driving_2m":
        prior, pose = driving_scene(B, h, w, speed=2.0)
    else:
        prior = (2 + 20 * smooth_field(rng, (B, 1, h, w), 12, 0, 1)).astype(np.float32)
        pose = rand_pose(oracle_lib, rng, B, 0.3, 2.0) if case == "wild" else rand_pose(oracle_lib, rng, B, 0.05, 0.3)
    hyp = oracle_lib.schedule_depth_range(prior, D, 0.3, None, "inverse")
    gout = rnd(rng.standard_normal((B, D, G, h, w)).astype(np.float32))
    exp_dref, exp_dsrc = oracle_lib.costvol_grouped_bwd(gout, ref, src, K, invK, hyp, pose)
    tol = 5e-4 if half else 1e-4
    pol = ops.backward_policy()
    pol.reset()
    key = (B, C, G, h, w, D, True, True)
    try:
        for launch in range(3):
            pol.force_balance = launch > 0
            r, s = feat_dev(ref, "nhwc", dtype=dtype), feat_dev(src, "nhwc", dtype=dtype)
            vol = ops.costvol_grouped(r, s, dev(K), dev(invK), dev(pose), G, prior=dev(prior), ndepth=D, scale_fac=0.3, type="inverse",
                                      layout="ndhwc")
            before = pol.balanced_launches
            vol.backward(torch.from_numpy(gout).to(dtype).cuda())
            torch.cuda.synchronize()
            assert pol.balanced_launches - before == int(launch > 0)
            e_r, e_s = relerr(host(r.grad), exp_dref), relerr(host(s.grad), exp_dsrc)
            cost = pol.costs(key).astype(np.float64)
            print("%s %s launch %d (%s): d_ref %.2e d_src %.2e; item cycles max / mean %.2f" % (
                case, dtype, launch, "balanced" if launch else "library partition", e_r, e_s, cost.max() / cost.mean()))
            assert e_r <= tol and e_s <= tol, (launch, e_r, e_s)
            assert (cost > 0).all()
            if launch > 0:
                sh = pol._shapes[key].shares_dev.cpu().numpy()
                assert sh[0, 0] == 0 and sh[-1, 1] == B * (h // 4) * (w // 16) * D
                assert (sh[1:, 0] == sh[:-1, 1]).all() and (sh[:, 1] >= sh[:, 0]).all()      # contiguous, ordered: every step exactly once
                assert ((sh % D) % 8 == 0).all()
                lens = sh[:, 1] - sh[:, 0]
                assert lens.max() > lens[lens > 0].min()                                       # it IS uneven in steps
    finally:
        pol.reset()


@pytest.mark.parametrize("feat", FEATS)
def test_costvol_full_size_properties(ops, feat):
    """BASELINE config 2 size (B=6, 48x160, D=96, C=32, G=16): size-independent properties, beside the oracle comparison of
    the same launch above.
    (1) identity pose => volume == group-mean(ref*src) for every hypothesis (KAT1);
    (2) linearity in ref and in src;  (3) <vol, g> == <ref, d_ref> == <src, d_src> (adjoint identity)."""
    torch.manual_seed(0)
    B, C, G, h, w, D = 6, 32, 16, 48, 160, 96
    Knp, invKnp = kitti_K(h, w, B)
    K, invK = dev(Knp), dev(invKnp)
    mf = torch.channels_last if feat == "nhwc" else torch.contiguous_format
    ref = torch.randn(B, C, h, w, device="cuda").contiguous(memory_format=mf)
    src = torch.randn(B, C, h, w, device="cuda").contiguous(memory_format=mf)
    prior = 2 + 20 * torch.rand(B, 1, h, w, device="cuda")
    eye = torch.eye(4, device="cuda").repeat(B, 1, 1)
    kw = dict(prior=prior, ndepth=D, scale_fac=0.3, type="inverse", layout="ndhwc")
    vol = ops.costvol_grouped(ref, src, K, invK, eye, G, **kw)
    exp = (ref * src).reshape(B, 2, G, h, w).mean(1)[:, None].expand(B, D, G, h, w)
    assert float((vol - exp).abs().max()) < 5e-4  # coordinate round trip is not bit exact (SURVEY KAT1)
    pose = eye.clone()
    pose[:, 0, 3] = 0.05
    pose[:, 2, 3] = 0.03
    ref2, src2 = torch.randn_like(ref), torch.randn_like(src)    # (randn_like keeps the memory format)
    v1 = ops.costvol_grouped(ref, src, K, invK, pose, G, **kw)
    v2 = ops.costvol_grouped(ref2, src, K, invK, pose, G, **kw)
    v12 = ops.costvol_grouped(ref + 2 * ref2, src, K, invK, pose, G, **kw)
    assert relerr(host(v12), host(v1 + 2 * v2)) < 1e-5
    v3 = ops.costvol_grouped(ref, src2, K, invK, pose, G, **kw)
    v13 = ops.costvol_grouped(ref, src - 3 * src2, K, invK, pose, G, **kw)
    assert relerr(host(v13), host(v1 - 3 * v3)) < 1e-5
    r, s = ref.clone().requires_grad_(True), src.clone().requires_grad_(True)   # (clone keeps the memory format)
    assert r.is_contiguous(memory_format=mf)
    v = ops.costvol_grouped(r, s, K, invK, pose, G, **kw)
    g = torch.randn_like(v)
    (v * g).sum().backward()
    lhs = float((v.detach().double() * g.double()).sum())
    assert abs(float((r.detach().double() * r.grad.double()).sum()) - lhs) < 1e-4 * abs(lhs)
    assert abs(float((s.detach().double() * s.grad.double()).sum()) - lhs) < 1e-4 * abs(lhs)


# ------------------------------------------------------------------ warp
@pytest.mark.parametrize("tag", ["small", "border"])
def test_warp_golden(ops, tag):
    g = load_golden("warp_" + tag)
    depth, T = dev(g["depth"], True), dev(g["T"], True)
    out, pix, mask = ops.warp_border(dev(g["img"]), depth, dev(g["K"]), dev(g["invK"]), T, want_pix=True, want_mask=True)
    # bit for bit against the REFERENCE's own sample grid and warped frame (md_common.hpp: the operation order the fixtures pin)
    assert np.array_equal(host(pix), g["pix_coords"]), "%d sample coordinates differ from the reference's" % int((host(pix) != g["pix_coords"]).sum())
    assert np.array_equal(host(out), g["warped"]), "%d warped values differ from the reference's" % int((host(out) != g["warped"]).sum())
    assert int((host(mask).astype(bool) != g["mvs_mask"]).sum()) == 0   # bit-exact: no pixel flips on the fixtures
    (out * dev(g["grad_out"])).sum().backward()
    print("warp golden", tag, "d_depth", relerr(host(depth.grad), g["d_depth"]), "d_T", relerr(host(T.grad), g["d_T"]))
    assert_close(host(depth.grad), g["d_depth"], rtol=1e-4, atol_scale=5e-3, what="d_depth")
    assert_close(host(T.grad), g["d_T"], rtol=1e-4, what="d_T")


def test_warp_fullres_reference_fixture(ops):
    """192 x 640 against the REFERENCE's own warp + autograd (tests/golden/warp_fullres.npz, tools/gen_golden.py gen_warp_fullres;
    inputs rebuilt from a seed by tests/golden_inputs.py, the fixture keeps the small outputs): stored samples of the sample grid and
    of the warped frame bit-equal, their row sums equal, the out-of-view count equal, d_T within 1e-4 (VERDICT r3 missing #4)."""
    from golden_inputs import warp_fullres_inputs
    g = load_golden("warp_fullres")
    img, depth, gout = warp_fullres_inputs()
    d, t = dev(depth, True), dev(g["T"], True)
    out, pix, mask = ops.warp_border(dev(img), d, dev(g["K"]), dev(g["invK"]), t, want_pix=True, want_mask=True)
    hp, ho = host(pix), host(out)
    assert np.array_equal(hp[:, ::16, ::16], g["pix_sample"]) and np.array_equal(ho[:, :, ::16, ::16], g["warped_sample"])
    assert np.allclose(hp.astype(np.float64).sum(2), g["pix_rowsum"], rtol=0, atol=1e-9)
    assert np.allclose(ho.astype(np.float64).sum(-1), g["warped_rowsum"], rtol=0, atol=1e-9)
    assert int(host(mask).astype(bool).sum()) == int(g["mask_count"])
    (out * dev(gout)).sum().backward()
    print("warp fullres vs reference fixture: d_T rel err", relerr(host(t.grad), g["d_T"]))
    assert_close(host(t.grad), g["d_T"], rtol=1e-4, what="d_T")
    dd = host(d.grad).reshape(depth.shape)
    assert_close(dd[:, :, ::16, ::16], g["d_depth_sample"], rtol=2e-4, atol_scale=5e-3, what="d_depth samples")
    err = np.abs(dd.astype(np.float64).sum(-1) - g["d_depth_rowsum"])
    assert float(err.max()) <= 2e-4 * float(g["d_depth_abs_rowsum"].max()), float(err.max())


def test_warp_vs_oracle_fullres(ops, oracle_lib):
    rng = np.random.default_rng(3)
    B, H, W = 2, 192, 640
    img = smooth_field(rng, (B, 3, H, W), 8)
    depth = (2 + 20 * smooth_field(rng, (B, 1, H, W), 16)).astype(np.float32)
    K, invK = kitti_K(H, W, B)
    T = rand_pose(oracle_lib, rng, B, 0.01, 0.1)
    # a smooth upstream gradient: with white noise d_T is a sqrt(N)-cancelling sum in which the few samples that
    # sit on a texel boundary (see assert_close_knife_edge) dominate the difference
    gout = smooth_field(rng, (B, 3, H, W), 8, 0.2, 1.0)
    exp, exp_pix = oracle_lib.warp(img, depth, K, invK, T)
    exp_dd, exp_dT = oracle_lib.warp_bwd(gout, img, depth, K, invK, T)
    d, t = dev(depth, True), dev(T, True)
    out, pix, _ = ops.warp_border(dev(img), d, dev(K), dev(invK), t, want_pix=True)
    # The kernel's backproject / project chain runs the oracle's operations one by one (no contraction, P formed as kt_rows
    # does): the sample positions -- hence every texel decision of the 245,760 samples -- and the warped frame are BIT-EQUAL.
    # (Until round 4 the positions agreed to 1e-5 only, a handful of samples landed in neighbouring texels, each moving its O(1)
    # term of a sum that cancels to ~1 % of its absolute value, and d_T was held to 1e-3.)
    nbad = int((host(pix) != exp_pix).sum())
    assert nbad == 0, "%d of %d sample coordinates differ from the oracle's bit patterns" % (nbad, exp_pix.size)
    assert np.array_equal(host(out), exp), "warped frame not bit-equal to the oracle's"
    (out * dev(gout)).sum().backward()
    assert_close_knife_edge(host(d.grad).reshape(exp_dd.shape), exp_dd, rtol=2e-4, what="d_depth")
    # d_T sums 122,880 per-pixel terms that cancel to ~1% of their absolute sum: north_star's 1e-4, now that the texels agree
    print("warp fullres d_T rel err", relerr(host(t.grad), exp_dT))
    assert_close(host(t.grad), exp_dT, rtol=1e-4, what="d_T")
    # the same sum in float64 at the kernel's own sample positions (kept from round 3)
    assert_close(host(t.grad), _warp_dT_float64(img, depth, K, invK, T, gout, host(pix)), rtol=1e-4, what="d_T at own positions")


def _warp_dT_float64(img, depth, K, invK, T, gout, pix):
    """dL/dT of grid_sample(img, pix, border, align_corners=True) with pix = Project3D(BackprojectDepth(depth)) (reference
    layers.py:556-621, autograd of trainer.py:519-529), in float64, taking the sampled texel of every pixel from `pix`."""
    B, Ci, H, W = img.shape
    img, gout, depth = img.astype(np.float64), gout.astype(np.float64), depth.astype(np.float64).reshape(B, H, W)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    out = np.zeros((B, 4, 4))
    for b in range(B):
        Kb, iKb, Tb = K[b].astype(np.float64), invK[b].astype(np.float64), T[b].astype(np.float64)
        ray = np.einsum("ij,jhw->ihw", iKb[:3, :3], np.stack([xs, ys, np.ones_like(xs)]))
        Xh = np.concatenate([ray * depth[b][None], np.ones((1, H, W))], 0)                      # (4,H,W)
        P = (Kb @ Tb)[:3]
        zz = np.einsum("j,jhw->hw", P[2], Xh) + 1e-7
        ix = (pix[b, ..., 0].astype(np.float64) + 1) / 2 * (W - 1)                                # the kernel's positions
        iy = (pix[b, ..., 1].astype(np.float64) + 1) / 2 * (H - 1)
        inx, iny = (ix >= 0) & (ix <= W - 1), (iy >= 0) & (iy <= H - 1)                           # border: clamped => zero grid gradient
        cx, cy = np.clip(ix, 0, W - 1), np.clip(iy, 0, H - 1)
        x0, y0 = np.minimum(np.floor(cx), W - 1).astype(int), np.minimum(np.floor(cy), H - 1).astype(int)
        wx1, wy1 = cx - x0, cy - y0
        x1, y1 = np.minimum(x0 + 1, W - 1), np.minimum(y0 + 1, H - 1)
        vx1, vy1 = (x0 + 1 < W), (y0 + 1 < H)
        gix, giy = np.zeros((H, W)), np.zeros((H, W))
        for c in range(Ci):
            im = img[b, c]
            nw, ne, sw, se = im[y0, x0], np.where(vx1, im[y0, x1], 0), np.where(vy1, im[y1, x0], 0), np.where(vx1 & vy1, im[y1, x1], 0)
            gix += gout[b, c] * ((ne - nw) * (1 - wy1) + (se - sw) * wy1)
            giy += gout[b, c] * ((sw - nw) * (1 - wx1) + (se - ne) * wx1)
        du, dv = gix * inx, giy * iny                      # d(ix)/du = 1: the [-1,1] normalise / un-normalise factors cancel
        dc = np.stack([du / zz, dv / zz, -(du * ix + dv * iy) / zz])                               # u = ix, v = iy where unclamped
        dP = np.einsum("ihw,jhw->ij", dc, Xh)                                                      # (3,4)
        out[b] = Kb[:3].T @ dP
    return out.astype(np.float32)


@pytest.mark.parametrize("hw", [(4, 8), (8, 16), (16, 32), (32, 64)])
def test_disp_to_depth_up(ops, oracle_lib, hw):
    rng = np.random.default_rng(5)
    B, H, W = 2, 32, 64
    h, w = hw
    disp = (0.01 + 0.5 * rng.random((B, 1, h, w))).astype(np.float32)
    up = oracle_lib.resize_bilinear(disp, H, W)
    _, exp = oracle_lib.disp_to_depth(up, 0.1, 100.0)
    d = dev(disp, True)
    depth = ops.disp_to_depth_up(d, H, W, 0.1, 100.0)
    assert_close(host(depth), exp, rtol=1e-5)
    g = rng.standard_normal((B, 1, H, W)).astype(np.float32)
    (depth * dev(g)).sum().backward()
    sd = 1.0 / exp
    g_up = -g * (1 / 0.1 - 1 / 100.0) * exp * exp
    exp_d = oracle_lib.resize_bilinear_bwd(g_up.astype(np.float32), h, w)
    assert_close(host(d.grad), exp_d, rtol=2e-5)


# ------------------------------------------------------------------ SSIM / reprojection loss
def test_ssim_golden(ops):
    g = load_golden("ssim")
    assert_close(host(ops.ssim_map(dev(g["pred"]), dev(g["target"]))), g["ssim"])
    pred = dev(g["pred"], True)
    rl = ops.reprojection_loss(pred, dev(g["target"]))
    assert_close(host(rl), g["reproj"])
    (rl * dev(g["grad_out"])).sum().backward()
    assert_close(host(pred.grad), g["d_pred"], rtol=2e-4)
    pred.grad = None
    rl0 = ops.reprojection_loss(pred, dev(g["target"]), ssim_w=0.0)
    assert_close(host(rl0), g["reproj_l1only"], rtol=1e-6)
    (rl0 * dev(g["grad_out"])).sum().backward()
    assert_close(host(pred.grad), g["d_pred_l1only"], rtol=1e-6)
    assert float(ops.ssim_map(dev(g["kat_x"]), dev(g["kat_x"])).abs().max()) < 1e-6  # KAT5
    z, o = torch.zeros(1, 3, 8, 8, device="cuda"), torch.ones(1, 3, 8, 8, device="cuda")
    assert abs(float(ops.ssim_map(z, o).mean()) - 0.49995) < 1e-5


@pytest.mark.parametrize("shape", [(2, 3, 192, 640), (1, 3, 37, 71), (2, 3, 3, 3)])
def test_reproj_loss_vs_oracle(ops, oracle_lib, shape):
    rng = np.random.default_rng(11)
    B, C, H, W = shape
    x = smooth_field(rng, shape, 2) if H > 8 else rng.random(shape, dtype=np.float32)
    y = smooth_field(rng, shape, 2) if H > 8 else rng.random(shape, dtype=np.float32)
    g = rng.standard_normal((B, 1, H, W)).astype(np.float32)
    p = dev(x, True)
    out = ops.reprojection_loss(p, dev(y))
    assert_close(host(out), oracle_lib.reproj_loss(x, y))
    (out * dev(g)).sum().backward()
    assert_close(host(p.grad), oracle_lib.reproj_loss_bwd(g, x, y), rtol=2e-4)
    assert_close(host(ops.ssim_map(dev(x), dev(y))), oracle_lib.ssim(x, y))


# ------------------------------------------------------------------ reductions
@pytest.mark.parametrize("mode", ["automask", "plain", "mvs", "ext"])
def test_masked_min_vs_oracle(ops, oracle_lib, mode):
    rng = np.random.default_rng(13)
    B, N, H, W = 2, 2, 48, 80
    rp = rng.random((B, N, H, W), dtype=np.float32)
    idn = rng.random((B, N, H, W), dtype=np.float32)
    noise = (rng.standard_normal((B, 1, H, W)) * 1e-5).astype(np.float32)
    ext = (rng.random((B, 1, H, W)) > 0.3).astype(np.float32)
    kw = dict(automask=dict(ident=idn, noise=noise), plain={}, mvs=dict(ident=idn, noise=noise, mvs_mode=True),
              ext=dict(ext_mask=ext))[mode]
    emn, emask, eloss = oracle_lib.masked_min(rp, **kw)
    r = dev(rp, True)
    loss, mn, mask = ops.masked_min_loss(r, **{k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
    assert abs(float(loss.detach()) - eloss) < 1e-5 * abs(eloss)
    assert np.array_equal(host(mask), emask)
    assert np.array_equal(host(mn), emn)
    (loss * 1.7).backward()
    assert_close(host(r.grad), oracle_lib.masked_min_bwd(1.7, rp, emask), rtol=1e-5)


@pytest.mark.parametrize("normalize", [True, False])
def test_smooth_vs_oracle_and_golden(ops, oracle_lib, normalize):
    g = load_golden("smooth")
    d = dev(g["disp"], True)
    loss = ops.smooth_loss(d, dev(g["img"]), normalize)
    exp = float(g["smooth_norm"] if normalize else g["smooth_raw"])
    assert abs(float(loss.detach()) - exp) < 1e-5 * exp
    if normalize:
        loss.backward()
        assert_close(host(d.grad), g["d_disp"])
    rng = np.random.default_rng(17)
    B, h, w = 3, 96, 320
    disp = (0.01 + rng.random((B, 1, h, w))).astype(np.float32)
    img = smooth_field(rng, (B, 3, h, w), 4)
    dd = dev(disp, True)
    l2 = ops.smooth_loss(dd, dev(img), normalize)
    e2 = oracle_lib.smooth_loss(disp, img, normalize)
    assert abs(float(l2.detach()) - e2) < 2e-5 * e2
    (l2 * 2.0).backward()
    assert_close(host(dd.grad), oracle_lib.smooth_loss_bwd(2.0, disp, img, normalize))


def test_smooth_all_levels_in_one_launch_vs_oracle(ops, oracle_lib):
    """md_smooth_multi_*: the four pyramid levels of one compute_losses call (trainer.py:712-714) in one launch per pass,
    each level against the oracle's get_smooth_loss (layers.py:630-643) and its gradient; one level left without an upstream
    gradient (its d_disp must be zero)."""
    rng = np.random.default_rng(19)
    B, H, W = 2, 96, 160
    disps = [(0.01 + rng.random((B, 1, H >> s, W >> s))).astype(np.float32) for s in range(4)]
    imgs = [smooth_field(rng, (B, 3, H >> s, W >> s), 4) for s in range(4)]
    td = [dev(d, True) for d in disps]
    losses = ops.smooth_losses(td, [dev(i) for i in imgs])
    gl = [1.0, 0.5, None, 2.0]
    for s in range(4):
        e = oracle_lib.smooth_loss(disps[s], imgs[s], True)
        assert abs(float(losses[s].detach()) - e) < 2e-5 * e, (s, float(losses[s]), e)
    sum(g * l for g, l in zip(gl, losses) if g is not None).backward()
    for s in range(4):
        if gl[s] is None:
            assert float(td[s].grad.abs().max()) == 0.0
        else:
            assert_close(host(td[s].grad), oracle_lib.smooth_loss_bwd(gl[s], disps[s], imgs[s], True), what="d_disp[%d]" % s)


# ------------------------------------------------------------------ post-volume
def test_postvol_golden(ops):
    g = load_golden("postvol")
    hyp = g["hyp"]
    logits = dev(g["logits"], True)
    depth, ent, prob = ops.softmax_entropy_localmax(logits, dev(1 / hyp[:, -1]), dev(1 / hyp[:, 0]), 1, want_prob=True)
    assert_close(host(depth), g["depth_r1"], rtol=1e-5)
    assert_close(host(ent), g["entropy"], rtol=1e-5)
    assert_close(host(prob), torch.softmax(torch.from_numpy(g["logits"]), 1).numpy(), rtol=1e-5)
    ((depth * dev(g["grad_depth"])).sum() + (ent * dev(g["grad_entropy"])).sum()).backward()
    assert_close(host(logits.grad), g["d_logits"], rtol=2e-4)
    d2, _, _ = ops.softmax_entropy_localmax(dev(g["logits"]), dev(1 / hyp[:, -1]), dev(1 / hyp[:, 0]), 2)
    assert_close(host(d2), g["depth_r2"], rtol=1e-5)


@pytest.mark.parametrize("shape,radius", [((2, 13, 7, 9), 1), ((1, 96, 48, 160), 1), ((2, 5, 3, 70), 2), ((1, 130, 6, 11), 1),
                                          ((1, 128, 4, 65), 3)])
def test_postvol_vs_oracle_and_torch(ops, oracle_lib, shape, radius):
    """D not a multiple of 4, ragged pixel counts, the BASELINE size, and D > 128 (one-thread-per-pixel fallback):
    forward against the oracle; backward against torch autograd of the reference's own formulation
    (layers.localmax / layers.entropy are line-by-line torch restatements)."""
    from movedepth_amd import layers
    rng = np.random.default_rng(41)
    B, D, h, w = shape
    logits = (rng.standard_normal(shape) * 2).astype(np.float32)
    a = (0.02 + 0.05 * rng.random((B, h, w))).astype(np.float32)
    b = (a + 0.1 + 0.2 * rng.random((B, h, w))).astype(np.float32)
    prob = oracle_lib.softmax_d(logits)
    lg = dev(logits, True)
    depth, ent, pr = ops.softmax_entropy_localmax(lg, dev(a), dev(b), radius, want_prob=True)
    assert_close(host(pr), prob, rtol=1e-5, what="prob")
    assert_close(host(ent), oracle_lib.entropy(prob), rtol=1e-5, what="entropy")
    assert_close(host(depth), oracle_lib.localmax(prob, radius, a, b), rtol=1e-5, what="depth")
    gd, ge = dev(rng.standard_normal((B, h, w)).astype(np.float32)), dev(rng.standard_normal((B, 1, h, w)).astype(np.float32))
    ((depth * gd).sum() + (ent * ge).sum()).backward()
    lt = dev(logits, True)
    pt = torch.softmax(lt, 1)
    dt = layers.localmax(pt, radius, D, dev(a), dev(b))
    et = layers.entropy(pt, dim=1, keepdim=True)
    ((dt * gd).sum() + (et * ge).sum()).backward()
    assert_close(host(lg.grad), host(lt.grad), rtol=2e-4, what="d_logits")


def test_postvol_launch_shape_reference_fixture(ops):
    """softmax -> entropy / localmax and the convex up-sampling, forward and backward, at BASELINE config 2's launch shape (B=6, D=96,
    48x160 -> 192x640) against the REFERENCE's own functions (tests/golden/postvol_launch.npz, tools/gen_golden.py gen_postvol_launch;
    inputs rebuilt from a seed): row / plane sums within 1e-4 of the absolute sums, lattices within 1e-5 / 2e-4."""
    from golden_inputs import postvol_launch_inputs
    g = load_golden("postvol_launch")
    logits, prior, g_depth, g_ent, up_depth, up_mask, g_up = postvol_launch_inputs()
    hyp = host(ops.schedule_depth_range(dev(prior), 96, 0.3, None, "inverse"))
    lat = lambda x: x[..., ::8, ::16]
    assert_close(lat(1 / hyp[:, -1]), g["inv_hi_lattice"], rtol=1e-6)
    lg = dev(logits, True)
    depth, ent, _ = ops.softmax_entropy_localmax(lg, dev(1 / hyp[:, -1]), dev(1 / hyp[:, 0]), 1)
    rs = lambda x: x.astype(np.float64).sum(-1)
    assert_close(lat(host(depth)), g["depth_lattice"], rtol=1e-5, what="depth")
    assert_close(lat(host(ent)), g["entropy_lattice"], rtol=1e-5, what="entropy")
    assert np.abs(rs(host(depth)) - g["depth_rowsum"]).max() <= 1e-5 * np.abs(g["depth_rowsum"]).max()
    assert np.abs(rs(host(ent)) - g["entropy_rowsum"]).max() <= 1e-5 * np.abs(g["entropy_rowsum"]).max()
    ((depth * dev(g_depth)).sum() + (ent * dev(g_ent)).sum()).backward()
    dl = host(lg.grad)
    assert np.abs(dl.astype(np.float64).sum((-1, -2)) - g["d_logits_planesum"]).max() <= 1e-4 * g["d_logits_abs_planesum"].max()
    assert_close(lat(dl)[:, ::8], g["d_logits_lattice"], rtol=2e-4, what="d_logits")
    ud, um = dev(up_depth, True), dev(up_mask, True)
    up = ops.convex_upsample(ud, um, 2)
    assert_close(host(up)[..., ::16, ::32], g["up_lattice"], rtol=1e-5, what="up-sampled depth")
    assert np.abs(rs(host(up)) - g["up_rowsum"]).max() <= 1e-5 * np.abs(g["up_rowsum"]).max()
    (up * dev(g_up)).sum().backward()
    gd, gm = host(ud.grad), host(um.grad)
    assert np.abs(rs(gd) - g["d_up_depth_rowsum"]).max() <= 1e-4 * g["d_up_depth_abs_rowsum"].max()
    assert_close(lat(gd), g["d_up_depth_lattice"], rtol=2e-5, what="d_up_depth")
    assert np.abs(rs(gm)[:, ::9] - g["d_up_mask_rowsum"]).max() <= 1e-4 * g["d_up_mask_abs_rowsum"].max()
    assert_close(lat(gm)[:, ::12], g["d_up_mask_lattice"], rtol=2e-5, what="d_up_mask")


def test_convex_upsample_golden(ops):
    g = load_golden("postvol")
    depth, mask = dev(g["up_depth"], True), dev(g["up_mask"], True)
    up = ops.convex_upsample(depth, mask, 2)
    assert_close(host(up), g["up_out"], rtol=1e-5)
    (up * dev(g["up_grad"])).sum().backward()
    assert_close(host(depth.grad), g["d_up_depth"], rtol=2e-5)
    assert_close(host(mask.grad), g["d_up_mask"], rtol=2e-5)


def test_convex_upsample_vs_oracle(ops, oracle_lib):
    rng = np.random.default_rng(23)
    B, h, w = 2, 48, 160
    depth = (2 + 20 * rng.random((B, h, w))).astype(np.float32)
    mask = rng.standard_normal((B, 144, h, w)).astype(np.float32)
    assert_close(host(ops.convex_upsample(dev(depth), dev(mask), 2)), oracle_lib.convex_upsample(depth, mask, 2), rtol=1e-5)


@pytest.mark.parametrize("scale,h,w", [(0, 9, 21), (1, 13, 37), (2, 12, 50), (3, 6, 11), (2, 48, 160)])
def test_convex_upsample_gradients_all_scales(ops, scale, h, w):
    """Both gradients against autograd of the reference's expression (layers.py:200-214: softmax over 9 taps of a zero-padded 3x3
    unfold) for every up-sampling factor, ragged sizes included: the backward reduces the depth gradient per coarse cell inside
    the fine-pixel kernel, with a different block shape per factor."""
    torch.manual_seed(scale * 7 + h)
    B, s = 2, 2 ** scale
    depth = (2 + 20 * torch.rand(B, h, w, device="cuda")).requires_grad_(True)
    mask = torch.randn(B, 9 * s * s, h, w, device="cuda", requires_grad=True)
    g = torch.randn(B, s * h, s * w, device="cuda")
    up = ops.convex_upsample(depth, mask, scale)
    (up * g).sum().backward()
    d2, m2 = depth.detach().clone().double().requires_grad_(True), mask.detach().clone().double().requires_grad_(True)
    p = torch.softmax(m2.view(B, 1, 9, s, s, h, w), 2)
    taps = torch.nn.functional.unfold(d2[:, None], [3, 3], padding=1).view(B, 1, 9, 1, 1, h, w)
    ref = (p * taps).sum(2).permute(0, 1, 4, 2, 5, 3).reshape(B, s * h, s * w)
    (ref * g.double()).sum().backward()
    assert_close(host(up), ref.detach().cpu().numpy(), rtol=1e-5)
    assert_close(host(depth.grad), d2.grad.cpu().numpy(), rtol=2e-5, what="d_depth")
    assert_close(host(mask.grad), m2.grad.cpu().numpy(), rtol=2e-5, what="d_mask")


def test_standalone_geometry_modules_golden(ops):
    """BackprojectDepth / Project3D keep the reference's call signature (layers.py:556-621)."""
    from movedepth_amd.layers import BackprojectDepth, Project3D

    g = load_golden("geometry")
    B, _, h, w = g["depth"].shape
    bp, pj = BackprojectDepth(B, h, w).cuda(), Project3D(B, h, w).cuda()
    with torch.no_grad():
        pts = bp(dev(g["depth"]), dev(g["invK"]))
        pix = pj(pts, dev(g["K"]), dev(g["T"]))
    assert_close(host(pts), g["cam_points"])
    assert_close(host(pix), g["pix_coords"], rtol=1e-5)
    # the differentiable (torch) fallback of the same modules gives the same values
    d = dev(g["depth"], True)
    pix2 = pj(bp(d, dev(g["invK"])), dev(g["K"]), dev(g["T"]))
    assert_close(host(pix2), g["pix_coords"], rtol=1e-5)
    pix2.sum().backward()
    assert torch.isfinite(d.grad).all()


# ------------------------------------------------------------------ ragged / edge shapes of the plane sweep
@pytest.mark.parametrize("case", [
    dict(B=1, C=32, G=16, h=5, w=7, D=3),       # smaller than one tile, odd sizes
    dict(B=3, C=32, G=16, h=37, w=53, D=5),     # nothing divides the tile; w % 4 != 0 (no 16-byte stores)
    dict(B=1, C=32, G=16, h=9, w=12, D=300),    # D larger than the LDS interval table: several slices per item
    dict(B=1, C=32, G=16, h=8, w=16, D=2),      # the minimum the fused schedule accepts
    dict(B=2, C=16, G=16, h=10, w=20, D=6),     # one channel per group, 16 channels
    dict(B=1, C=8, G=2, h=10, w=20, D=6),       # 4 channels per group, 2 groups (no group-quad map)
])
@pytest.mark.parametrize("layout,feat", [("bdg", "nchw"), ("ndhwc", "nchw"), ("ndhwc", "nhwc")])
@pytest.mark.parametrize("sched", ["inverse", "linear", "log"])
def test_costvol_ragged_shapes(ops, oracle_lib, case, layout, sched, feat):
    rng = np.random.default_rng(29)
    B, C, G, h, w, D = (case[k] for k in "BCGhwD")
    ref = smooth_field(rng, (B, C, h, w), 2, -1, 1)
    src = smooth_field(rng, (B, C, h, w), 2, -1, 1)
    K, invK = kitti_K(h, w, B)
    prior = (2 + 20 * rng.random((B, 1, h, w))).astype(np.float32)
    pose = rand_pose(oracle_lib, rng, B, 0.02, 0.1)
    z = 30.0 * pose[:, 2, 3]
    hyp = oracle_lib.schedule_depth_range(prior, D, 0.3, z, sched)
    # the standalone schedule kernel and the schedule fused into the volume kernel agree with the oracle
    assert_close(host(ops.schedule_depth_range(dev(prior), D, 0.3, dev(z), sched)), hyp, rtol=1e-5)
    exp = oracle_lib.costvol_grouped(ref, src, K, invK, hyp, pose, G)
    gout = rng.standard_normal(exp.shape).astype(np.float32)
    exp_dref, exp_dsrc = oracle_lib.costvol_grouped_bwd(gout, ref, src, K, invK, hyp, pose)
    r, s = feat_dev(ref, feat), feat_dev(src, feat)
    vol = ops.costvol_grouped(r, s, dev(K), dev(invK), dev(pose), G, prior=dev(prior), ndepth=D, scale_fac=0.3,
                              z_trans=dev(z), type=sched, layout=layout)
    assert vol.shape == exp.shape
    assert_close(host(vol), exp, what="volume")
    (vol * dev(gout)).sum().backward()
    assert_close(host(r.grad), exp_dref, what="d_ref")
    assert_close(host(s.grad), exp_dsrc, what="d_src")


def test_costvol_single_hypothesis_and_bad_range(ops, oracle_lib):
    """D = 1 (hypotheses passed explicitly) and the reference's unguarded 1 + f*z <= 0 range:
    negative / non-finite hypotheses must not crash and must match wherever the oracle is finite."""
    rng = np.random.default_rng(31)
    B, C, G, h, w = 2, 32, 16, 8, 16
    ref = smooth_field(rng, (B, C, h, w), 2, -1, 1)
    src = smooth_field(rng, (B, C, h, w), 2, -1, 1)
    K, invK = kitti_K(h, w, B)
    pose = rand_pose(oracle_lib, rng, B, 0.02, 0.1)
    hyp1 = (2 + 20 * rng.random((B, 1, h, w))).astype(np.float32)
    out = ops.costvol_grouped(dev(ref), dev(src), dev(K), dev(invK), dev(pose), G, depth_priors=dev(hyp1), layout="bdg")
    assert_close(host(out), oracle_lib.costvol_grouped(ref, src, K, invK, hyp1, pose, G))
    zbad = np.array([-3.5, -10.0 / 3.0], np.float32)
    with np.errstate(all="ignore"):
        hyp = oracle_lib.schedule_depth_range(hyp1, 6, 0.3, zbad, "inverse")
        exp = oracle_lib.costvol_grouped(ref, src, K, invK, hyp, pose, G)
    got = host(ops.costvol_grouped(dev(ref), dev(src), dev(K), dev(invK), dev(pose), G, depth_priors=dev(hyp), layout="bdg"))
    fin = np.isfinite(exp) & np.isfinite(got)
    assert fin.mean() > 0.4
    assert_close(got[fin], exp[fin])


def test_unsupported_grouping_fails_loudly(ops):
    from movedepth_amd._lib import MovedepthHipError

    x = torch.zeros(1, 6, 8, 8, device="cuda")
    K = torch.eye(4, device="cuda")[None]
    with pytest.raises(MovedepthHipError):
        ops.costvol_grouped(x, x, K, K, K, 2, depth_priors=torch.ones(1, 2, 8, 8, device="cuda"))  # C/G = 3


# ------------------------------------------------------------------ reg3d's last layer (C -> 1 convolution)
def _cl3d(t):
    return t.contiguous(memory_format=torch.channels_last_3d)


@pytest.mark.parametrize("weight_cl", [False, True])
@pytest.mark.parametrize("tag", ["c16", "c8"])
def test_prob_conv_golden(ops, tag, weight_cl):
    g = load_golden("prob_conv_" + tag)
    x = _cl3d(dev(g["x"])).requires_grad_(True)
    w = dev(g["weight"])
    w = (_cl3d(w) if weight_cl else w).requires_grad_(True)
    y = ops.conv3d_c1(x, w)
    assert y.shape == (x.shape[0], 1) + tuple(x.shape[2:])
    assert_close(host(y)[:, 0], g["y"], what="prob y")
    (y[:, 0] * dev(g["grad_out"])).sum().backward()
    assert x.grad.is_contiguous(memory_format=torch.channels_last_3d) and w.grad.stride() == w.stride()
    assert_close(host(x.grad), g["d_x"], what="prob d_x")
    assert_close(host(w.grad), g["d_weight"], what="prob d_weight")


@pytest.mark.parametrize("shape", [
    (2, 16, 19, 13, 45),    # ragged everywhere; D=19 -> 2 slices of 10 and 9 planes
    (1, 16, 40, 8, 32),     # exactly one tile, 4 slices: every slice boundary inside the volume
    (1, 8, 17, 20, 70),     # 8 channels (2 quads per voxel), 3x3 tiles
    (3, 16, 1, 5, 3),       # a single plane, smaller than the halo
    (1, 16, 2, 1, 1),       # one voxel column
])
def test_prob_conv_vs_oracle(ops, oracle_lib, shape):
    rng = np.random.default_rng(21)
    B, C, D, H, W = shape
    x = rng.standard_normal(shape).astype(np.float32)
    wt = (rng.standard_normal((1, C, 3, 3, 3)) * 0.1).astype(np.float32)
    gy = rng.standard_normal((B, 1, D, H, W)).astype(np.float32)
    exp = oracle_lib.conv3d_c1(x, wt)
    exp_dx, exp_dw = oracle_lib.conv3d_c1_bwd(gy, x, wt)
    xt, wtt = _cl3d(dev(x)).requires_grad_(True), dev(wt, True)
    y = ops.conv3d_c1(xt, wtt)
    assert_close(host(y), exp, what="y")
    (y * dev(gy)).sum().backward()
    assert_close(host(xt.grad), exp_dx, what="d_x")
    assert_close(host(wtt.grad), exp_dw, what="d_weight")


def test_prob_conv_full_size_vs_library(ops):
    """BASELINE config 2 size (6x16x96x48x160): against the library convolution on the same device (the CPU oracle
    would take minutes), plus linearity in x (a size-independent property) and determinism of the weight gradient."""
    torch.manual_seed(5)
    B, C, D, H, W = 6, 16, 96, 48, 160
    x = _cl3d(torch.randn(B, C, D, H, W, device="cuda")).requires_grad_(True)
    w = (torch.randn(1, C, 3, 3, 3, device="cuda") * 0.1).requires_grad_(True)
    gy = torch.randn(B, 1, D, H, W, device="cuda")
    y = ops.conv3d_c1(x, w)
    dx, dw = torch.autograd.grad(y, (x, w), gy)
    x2, w2 = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    y_ref = torch.nn.functional.conv3d(x2, w2, padding=1)
    dx_ref, dw_ref = torch.autograd.grad(y_ref, (x2, w2), gy)
    assert_close(host(y), host(y_ref), what="y")
    assert_close(host(dx), host(dx_ref), what="d_x")
    assert_close(host(dw), host(dw_ref), what="d_weight")
    with torch.no_grad():
        x3 = _cl3d(torch.randn_like(x))
        lin = ops.conv3d_c1(x.detach() + 0.5 * x3, w.detach())
        assert_close(host(lin), host(y.detach() + 0.5 * ops.conv3d_c1(x3, w.detach())), what="linearity")
    dw_again = torch.autograd.grad(ops.conv3d_c1(x, w), w, gy)[0]
    assert torch.equal(dw, dw_again), "weight gradient must be bit-reproducible (fixed-order reduction)"


def test_prob_conv_rejects_unsupported(ops):
    from movedepth_amd._lib import MovedepthHipError
    x = _cl3d(torch.randn(1, 12, 4, 4, 4, device="cuda"))
    with pytest.raises(MovedepthHipError):
        ops.conv3d_c1(x, torch.randn(1, 12, 3, 3, 3, device="cuda"))
    with pytest.raises(MovedepthHipError):
        ops.conv3d_c1(torch.randn(1, 16, 4, 4, 4), torch.randn(1, 16, 3, 3, 3))


def test_reg3d_prob_paths_agree(ops):
    """reg3d with the hand-written last layer against the same module with the library convolution."""
    from movedepth_amd import networks
    torch.manual_seed(3)
    net = networks.reg3d(16, 16, 3).cuda().to(memory_format=torch.channels_last_3d)
    vol = torch.randn(2, 16, 16, 24, 32, device="cuda")  # (B,D,G,h,w)
    outs = []
    for hip in (True, False):
        net.hip_prob = hip
        net.zero_grad()
        v = vol.clone().requires_grad_(True)
        o = net(v)
        o.square().mean().backward()
        outs.append((host(o), host(v.grad), host(net.prob.weight.grad)))
    for a, b, what in zip(outs[0], outs[1], ("logits", "d_volume", "d_prob_weight")):
        assert_close(a, b, what=what)


# ------------------------------------------------------------------ reg3d's first layer (16 -> 16): weight gradient
@pytest.mark.parametrize("lib_fwd_dgrad", [False, True])
@pytest.mark.parametrize("weight_cl", [False, True])
def test_conv0_golden(ops, weight_cl, lib_fwd_dgrad):
    g = load_golden("conv0_c16")
    x = _cl3d(dev(g["x"])).requires_grad_(True)
    w = dev(g["weight"])
    w = (_cl3d(w) if weight_cl else w).requires_grad_(True)
    y = ops.conv3d_16(x, w, lib_fwd_dgrad)
    assert_close(host(y), g["y"], what="conv0 y")
    (y * dev(g["grad_out"])).sum().backward()
    assert w.grad.stride() == w.stride() and x.grad.is_contiguous(memory_format=torch.channels_last_3d)
    assert_close(host(w.grad), g["d_weight"], what="conv0 d_weight")
    assert_close(host(x.grad), g["d_x"], what="conv0 d_x")


@pytest.mark.parametrize("shape", [
    (2, 19, 13, 45),   # ragged everywhere, 2 D slices
    (1, 40, 8, 32),    # one tile, 4 slices
    (1, 17, 20, 70),   # 3x3 tiles
    (3, 1, 5, 3),      # a single plane smaller than the halo
    (1, 2, 1, 1),      # one voxel column
])
@pytest.mark.parametrize("planar", [False, True])
def test_conv0_vs_oracle(ops, oracle_lib, shape, planar):
    """planar: x is the cost volume's `bgd` storage (contiguous [B,16,D,H,W]), read in place; dx comes back planar."""
    rng = np.random.default_rng(31)
    B, D, H, W = shape
    x = rng.standard_normal((B, 16, D, H, W)).astype(np.float32)
    wt = (rng.standard_normal((16, 16, 3, 3, 3)) * 0.1).astype(np.float32)
    gy = rng.standard_normal((B, 16, D, H, W)).astype(np.float32)
    exp_y, exp_dx, exp_dw = oracle_lib.conv3d(x, wt, gy)
    xt = (dev(x) if planar else _cl3d(dev(x))).requires_grad_(True)
    wtt = dev(wt, True)
    y = ops.conv3d_16(xt, wtt)
    assert y.is_contiguous(memory_format=torch.channels_last_3d)
    assert_close(host(y), exp_y, what="y")
    dx, dw = torch.autograd.grad(y, (xt, wtt), _cl3d(dev(gy)))
    assert dx.stride() == xt.stride(), "the data gradient must come back in x's layout"
    assert_close(host(dx), exp_dx, what="d_x")
    assert_close(host(dw), exp_dw, what="d_weight")


@pytest.mark.parametrize("weight_cl", [False, True])
@pytest.mark.parametrize("chans,shape", [
    ((32, 32), (2, 5, 9, 33)),     # ragged tiles, two blocks each way
    ((48, 32), (1, 3, 6, 31)),     # three input blocks: two accumulating launches per output block
    ((32, 64), (1, 4, 4, 40)),
    ((16, 32), (2, 2, 5, 17)),
    ((64, 64), (1, 9, 7, 12)),
])
def test_conv3d_channel_blocks_vs_oracle(ops, oracle_lib, chans, shape, weight_cl):
    """reg3d's interior stride-1 layers (networks/resnet_encoder.py:233-239, applied :260-262: ConvBnReLU3D(32,32), (64,64), (128,128)) as sums over
    16 x 16 channel blocks on the 16 -> 16 layer's bf16 x 3 kernels (md_conv3d_cb_*: record pitch / block offset / accumulate in the
    epilogue): forward, data gradient and weight gradient against the oracle's direct convolution."""
    rng = np.random.default_rng(41)
    Ci, Co = chans
    B, D, H, W = shape
    x = rng.standard_normal((B, Ci, D, H, W)).astype(np.float32)
    wt = (rng.standard_normal((Co, Ci, 3, 3, 3)) * 0.1).astype(np.float32)
    gy = rng.standard_normal((B, Co, D, H, W)).astype(np.float32)
    exp_y, exp_dx, exp_dw = oracle_lib.conv3d(x, wt, gy)
    xt = _cl3d(dev(x)).requires_grad_(True)
    wtt = dev(wt)
    if weight_cl:
        wtt = wtt.contiguous(memory_format=torch.channels_last_3d)
    wtt.requires_grad_(True)
    y = ops.conv3d_cb(xt, wtt)
    assert y.is_contiguous(memory_format=torch.channels_last_3d)
    assert_close(host(y), exp_y, what="y")
    dx, dw = torch.autograd.grad(y, (xt, wtt), _cl3d(dev(gy)))
    assert dw.stride() == wtt.stride()
    assert_close(host(dx), exp_dx, what="d_x")
    assert_close(host(dw), exp_dw, what="d_weight")


def test_conv3d_channel_blocks_layer_size_vs_library(ops):
    """reg3d.conv2 at BASELINE config 2 (6 x 32 x 48 x 24 x 80): all three directions against the library's fp32 convolution, the
    forward / data-gradient adjoint identity, and the weight gradient bit-reproducible."""
    torch.manual_seed(8)
    B, C, D, H, W = 6, 32, 48, 24, 80
    x = _cl3d(torch.randn(B, C, D, H, W, device="cuda")).requires_grad_(True)
    w = (torch.randn(C, C, 3, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    gy = _cl3d(torch.randn(B, C, D, H, W, device="cuda"))
    y = ops.conv3d_cb(x, w)
    dx, dw = torch.autograd.grad(y, (x, w), gy, retain_graph=True)
    y_ref = torch.nn.functional.conv3d(x.detach(), w.detach(), padding=1)
    dx_ref, dw_ref, _ = torch.ops.aten.convolution_backward(gy, x.detach(), w.detach(), None, [1] * 3, [1] * 3, [1] * 3, False,
                                                            [0] * 3, 1, [True, True, False])
    assert_close(host(y), host(y_ref), what="y vs library")
    assert_close(host(dx), host(dx_ref), what="d_x vs library")
    assert_close(host(dw), host(dw_ref), what="d_weight vs library")
    lhs, rhs = (y.detach().double() * gy.double()).sum().item(), (x.detach().double() * dx.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0), "forward and data gradient are not adjoint: %r %r" % (lhs, rhs)
    dw2 = torch.autograd.grad(y, w, gy)[0]
    assert torch.equal(dw, dw2), "the weight gradient must be bit-reproducible"


def test_conv0_full_size_vs_library(ops):
    """BASELINE config 2 size: all three directions against the library; weight gradient bit-reproducible and linear
    in gy; forward / data gradient adjoint to each other (<y, gy> == <x, dx>, a size-independent property)."""
    torch.manual_seed(6)
    B, D, H, W = 6, 96, 48, 160
    x = _cl3d(torch.randn(B, 16, D, H, W, device="cuda")).requires_grad_(True)
    w = (torch.randn(16, 16, 3, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    gy = _cl3d(torch.randn(B, 16, D, H, W, device="cuda"))
    y = ops.conv3d_16(x, w)
    dx, dw = torch.autograd.grad(y, (x, w), gy, retain_graph=True)
    y_ref = torch.nn.functional.conv3d(x.detach(), w.detach(), padding=1)
    dx_ref, dw_ref, _ = torch.ops.aten.convolution_backward(gy, x.detach(), w.detach(), None, [1] * 3, [1] * 3, [1] * 3, False,
                                                            [0] * 3, 1, [True, True, False])
    assert_close(host(y), host(y_ref), what="y vs library")
    assert_close(host(dx), host(dx_ref), what="d_x vs library")
    assert_close(host(dw), host(dw_ref), what="d_weight vs library")
    lhs, rhs = (y.detach().double() * gy.double()).sum().item(), (x.detach().double() * dx.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0), "forward and data gradient are not adjoint: %r %r" % (lhs, rhs)
    # the same volume in planar storage must give the same three results (different kernels: LDS layout, operand roles)
    xp = x.detach().contiguous().requires_grad_(True)
    yp = ops.conv3d_16(xp, w)
    dxp, dwp = torch.autograd.grad(yp, (xp, w), gy)
    assert dxp.is_contiguous()
    assert_close(host(yp), host(y_ref), what="y (planar x) vs library")
    assert_close(host(dxp), host(dx_ref), what="d_x (planar) vs library")
    assert_close(host(dwp), host(dw_ref), what="d_weight (planar x) vs library")
    (dw2,) = torch.autograd.grad(y, w, gy, retain_graph=True)
    assert torch.equal(dw, dw2), "weight gradient must be bit-reproducible"
    gy2 = _cl3d(torch.randn_like(gy))
    (dw_lin,) = torch.autograd.grad(y, w, gy + 0.5 * gy2, retain_graph=True)
    (dw_b,) = torch.autograd.grad(y, w, gy2)
    assert_close(host(dw_lin), host(dw + 0.5 * dw_b), what="linearity in gy")


def _reg3d_run(net, vol):
    """forward + backward of reg3d; returns (logits, d_volume, d_conv0_weight) and the ReLU masks (sign of every
    BatchNorm output)."""
    from movedepth_amd import networks
    masks, hooks = [], []

    def fused_mask(mod, i, o):  # FusedBNReLU3d returns the activation (+ skip): recompute the pre-activation's sign
        with torch.no_grad():
            z = torch.nn.functional.batch_norm(i[0].detach(), None, None, mod.weight, mod.bias, True, 0.0, mod.eps)
        masks.append((z > 0).cpu())

    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            hooks.append(m.register_forward_hook(lambda mod, i, o: masks.append((o.detach() > 0).cpu())))
        elif isinstance(m, networks.FusedBNReLU3d):
            hooks.append(m.register_forward_hook(fused_mask))
    net.zero_grad()
    v = vol.detach().requires_grad_(True)  # keeps the strides
    o = net(v)
    for hk in hooks:
        hk.remove()
    o.square().mean().backward()
    return (host(o), host(v.grad), host(net.conv0.conv.weight.grad)), masks


@pytest.mark.parametrize("vol_layout", ["bdg", "bgd", "ndhwc"])
def test_reg3d_conv0_paths_agree(ops, vol_layout):
    """reg3d fed with the volume in each storage the cost-volume kernel can write: same logits and gradients whether
    its first convolution runs on the MFMA kernels (in place on that storage) or on the library.

    Gradients through 11 ReLUs are only comparable at 1e-4 when both runs take the same side of every ReLU: one
    pre-activation within fp32 rounding of zero (expected about once per run at these sizes; measured with seed 4 /
    layout bgd: one element of 393,216 at 2e-6 against a layer rms of 1, flipped in the LIBRARY run relative to an
    fp64 copy of the network, tools/diag_reg3d_paths.py) changes the whole volume gradient by ~5e-4 norm-wise.  So:
    logits always at 1e-4; gradients at 1e-4 when the masks agree, at 5e-3 and with at most 3 flipped elements
    otherwise."""
    from movedepth_amd import networks
    torch.manual_seed(4)
    net = networks.reg3d(16, 16, 3).cuda().to(memory_format=torch.channels_last_3d)
    B, D, G, h, w = 2, 16, 16, 24, 32
    if vol_layout == "bdg":
        vol = torch.randn(B, D, G, h, w, device="cuda")
    elif vol_layout == "bgd":
        vol = torch.randn(B, G, D, h, w, device="cuda").permute(0, 2, 1, 3, 4)
    else:
        vol = torch.randn(B, D, h, w, G, device="cuda").permute(0, 1, 4, 2, 3)
    runs = []
    for hip, lib_fd in ((True, False), (True, True), (False, False)):
        net.hip_conv0_wgrad, net.lib_conv0_fwd_dgrad = hip, lib_fd
        runs.append(_reg3d_run(net, vol))
    ref_out, ref_masks = runs[2]
    for out, masks in runs[:2]:
        flips = sum(int((a != b).sum()) for a, b in zip(masks, ref_masks))
        assert flips <= 3, "%d ReLU decisions differ between the two conv0 implementations" % flips
        assert_close(out[0], ref_out[0], what="logits")
        for a, b, what in zip(out[1:], ref_out[1:], ("d_volume", "d_conv0_weight")):
            assert_close(a, b, rtol=1e-4 if flips == 0 else 5e-3, what="%s (%d ReLU flips)" % (what, flips))




def test_reg3d_conv2_paths_agree(ops):
    """reg3d with its 32 -> 32 layer (conv2) on the channel-block kernels (ConvBnReLU3D.hip_cb, the trainer's --hip_conv2) and on the
    library: same logits and gradients, as test_reg3d_conv0_paths_agree; and the hand-written entry points did run."""
    from movedepth_amd import networks
    torch.manual_seed(5)
    net = networks.reg3d(16, 16, 3).cuda().to(memory_format=torch.channels_last_3d)
    assert net.conv2.conv.in_channels == 32 and net.conv2.conv.out_channels == 32
    B, D, G, h, w = 2, 16, 16, 24, 32
    vol = torch.randn(B, D, h, w, G, device="cuda").permute(0, 1, 4, 2, 3)
    runs = []
    for on in (True, False):
        net.conv2.hip_cb = on
        if on:
            ops.enable_kernel_timing(["md_conv3d_cb_fwd", "md_conv3d_cb_bwd_data", "md_conv3d_cb_bwd_weight"])
        out, masks = _reg3d_run(net, vol)
        if on:
            t = ops.kernel_times_us()
            ops.enable_kernel_timing([])
            assert all(t.get(k, {}).get("launches", 0) >= 1 for k in ("md_conv3d_cb_fwd", "md_conv3d_cb_bwd_data", "md_conv3d_cb_bwd_weight")), t
        runs.append((out, masks, host(net.conv2.conv.weight.grad)))
        net.zero_grad(set_to_none=True)
    (out, masks, dw2), (ref_out, ref_masks, ref_dw2) = runs
    flips = sum(int((a != b).sum()) for a, b in zip(masks, ref_masks))
    assert flips <= 3, "%d ReLU decisions differ between the two conv2 implementations" % flips
    assert_close(out[0], ref_out[0], what="logits")
    for a, b, what in zip(out[1:] + (dw2,), ref_out[1:] + (ref_dw2,), ("d_volume", "d_conv0_weight", "d_conv2_weight")):
        assert_close(a, b, rtol=1e-4 if flips == 0 else 5e-3, what="%s (%d ReLU flips)" % (what, flips))


# ------------------------------------------------------------------ pose parameters -> 4x4
@pytest.mark.parametrize("invert", [False, True])
def test_pose_matrix_golden(ops, invert):
    g = load_golden("pose_grad")
    k = "_inv" if invert else ""
    aa, tr = dev(g["axisangle"], True), dev(g["translation"], True)
    T = ops.pose_matrix(aa, tr, invert)
    assert_close(host(T), g["T" + k], rtol=1e-6)
    (T * dev(g["grad_T" + k])).sum().backward()
    assert aa.grad.shape == aa.shape and tr.grad.shape == tr.shape
    # sample 5 (|v| = 5.9e-5) is ill-conditioned in fp32 for the reference as well (see test_oracle_golden.py): compare the
    # well-conditioned samples with the reference, the tiny-angle one with the fp64 oracle at a looser bound
    assert_close(host(aa.grad)[:5], g["d_axisangle" + k][:5], rtol=1e-5, what="d_axisangle")
    import oracle
    oracle.build()
    da64, _ = oracle.transformation_from_parameters_bwd(g["grad_T" + k], g["axisangle"], g["translation"], invert)
    assert_close(host(aa.grad).reshape(-1, 3)[5:], da64[5:], rtol=1e-3, what="d_axisangle (tiny angle) vs fp64 oracle")
    assert_close(host(tr.grad), g["d_translation" + k], rtol=1e-5, what="d_translation")


def test_pose_matrix_matches_torch_form_and_layers_dispatch(ops):
    """layers.transformation_from_parameters uses the kernel on the GPU and torch ops on the CPU: same values/gradients;
    zero rotation gives the identity rotation and a finite (zero) angle gradient."""
    from movedepth_amd import layers
    torch.manual_seed(9)
    aa = torch.randn(16, 1, 3) * 0.2
    tr = torch.randn(16, 1, 3)
    aa[0] = 0.0
    for inv in (False, True):
        a_c, t_c = aa.clone().requires_grad_(True), tr.clone().requires_grad_(True)
        a_g, t_g = aa.cuda().requires_grad_(True), tr.cuda().requires_grad_(True)
        Tc, Tg = layers.transformation_from_parameters(a_c, t_c, inv), layers.transformation_from_parameters(a_g, t_g, inv)
        W = torch.randn(16, 4, 4)
        (Tc * W).sum().backward()
        (Tg * W.cuda()).sum().backward()
        assert_close(host(Tg), Tc.detach().numpy(), rtol=1e-6)
        assert torch.isfinite(a_g.grad).all()
        assert_close(host(a_g.grad)[1:], a_c.grad.numpy()[1:], rtol=1e-5, what="d_axisangle")  # [0]: torch gives nan/0 at v = 0
        assert_close(host(t_g.grad), t_c.grad.numpy(), rtol=1e-5, what="d_translation")
    np.testing.assert_allclose(host(Tg)[0, :3, :3] if inv else host(Tg)[0, :3, :3], np.eye(3), atol=1e-7)


# ------------------------------------------------------------------ fused BatchNorm + ReLU (+ residual), 16 channels
@pytest.mark.parametrize("shape,with_res", [((2, 16, 5, 7, 9), False), ((2, 16, 5, 7, 9), True), ((1, 16, 1, 1, 3), True),
                                             ((2, 32, 6, 5, 11), True), ((3, 64, 3, 4, 7), False), ((6, 32, 48, 24, 80), True),
                                             ((6, 16, 96, 48, 160), True)])
def test_bn_relu_vs_torch(ops, shape, with_res):
    """relu(batch_norm(x)) [+ res] in training mode against the torch ops the reference module uses (BatchNorm3d + ReLU,
    resnet_encoder.py:231 / :249-252, skip add :264): output, all gradients, running statistics.  The last shape is
    BASELINE config 2."""
    torch.manual_seed(13)
    C = shape[1]
    x = _cl3d(torch.randn(*shape, device="cuda") * 1.5 + 0.3)
    res = _cl3d(torch.randn(*shape, device="cuda")) if with_res else None
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.2
    gy = _cl3d(torch.randn(*shape, device="cuda"))
    rm_a, rv_a = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    rm_b, rv_b = rm_a.clone(), rv_a.clone()
    xa, ga, ba = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ra = res.clone().requires_grad_(True) if with_res else None
    ya = ops.bn_relu_3d(xa, ga, ba, ra, rm_a, rv_a, 0.1, 1e-5)
    ya.backward(gy)
    # reference in fp64 (same torch ops): at 70 M elements a ReLU pre-activation within fp32 rounding of zero is expected
    # (one such element moves the norm-wise error of d_x to 1.7e-4), so d_x allows a 1e-6 fraction of outlier elements
    xb, gb, bb = (t.double().clone().requires_grad_(True) for t in (x, gamma, beta))
    rb = res.double().clone().requires_grad_(True) if with_res else None
    rm_b, rv_b = rm_b.double(), rv_b.double()
    yb = torch.relu(torch.nn.functional.batch_norm(xb, rm_b, rv_b, gb, bb, True, 0.1, 1e-5))
    if with_res:
        yb = yb + rb
    yb.backward(gy.double())
    n = x.numel() // C
    if n > 1:
        assert_close(host(ya), host(yb), what="y")
        assert_close_knife_edge(host(xa.grad), host(xb.grad), rtol=2e-4 if n < 100 else 1e-4, max_outlier_frac=1e-6 if n > 1e5 else 0.0,
                                what="d_x")
        # d_gamma / d_beta are sums of n random-sign terms here (gy is white noise), i.e. ~sqrt(n) in size, so the handful
        # of ReLU decisions that differ from fp64 at 70 M elements (expected ~6: 70e6 * density(0) * 2e-7) show up at
        # ~6^0.5 * 0.25 / 6000 = 1e-4 of the norm; measured 1.15e-4.  Small shapes have no such elements.
        rt = 5e-4 if n > 1e5 else 1e-4
        assert_close(host(ga.grad), host(gb.grad), rtol=rt, what="d_gamma")
        assert_close(host(ba.grad), host(bb.grad), rtol=rt, what="d_beta")
        assert_close(host(rm_a), host(rm_b), rtol=1e-5, what="running_mean")
        assert_close(host(rv_a), host(rv_b), rtol=1e-5, what="running_var")
    if with_res:
        assert torch.equal(ra.grad, gy)


def test_reg3d_fused_bn_paths_agree(ops):
    """reg3d with the fused BatchNorm+ReLU(+skip) kernels against the same module on torch ops: same state_dict keys,
    logits at 1e-4, gradients at 1e-4 when every ReLU decision agrees (see test_reg3d_conv0_paths_agree)."""
    from movedepth_amd import networks
    torch.manual_seed(4)
    net = networks.reg3d(16, 16, 3, fused_bn=True).cuda().to(memory_format=torch.channels_last_3d)
    assert set(net.state_dict().keys()) == set(networks.reg3d(16, 16, 3).state_dict().keys())
    keys = set(net.state_dict().keys())
    assert {"conv0.bn.weight", "conv0.bn.running_var", "conv0.bn.num_batches_tracked", "conv11.1.weight", "conv11.1.bias",
            "conv11.1.running_mean"} <= keys and not any(k.startswith("conv11.2") for k in keys)
    vol = torch.randn(2, 16, 24, 32, 16, device="cuda").permute(0, 1, 4, 2, 3)   # channels-last volume
    runs = []
    for fused in (True, False):
        for m in net.modules():
            if isinstance(m, networks.FusedBNReLU3d):
                m.fused = fused
        runs.append(_reg3d_run(net, vol))
    (out, _), (ref_out, _) = runs
    assert_close(out[0], ref_out[0], what="logits")
    for a, b, what in zip(out[1:], ref_out[1:], ("d_volume", "d_conv0_weight")):
        assert_close(a, b, rtol=5e-3, what=what)   # masks of the fused layers are not hookable: loose bound, see docstring


# ------------------------------------------------------------------ cost volume with 2-byte feature maps / volume
@pytest.mark.parametrize("dtype,tol_rounded,tol_exact", [(torch.bfloat16, 1.5e-3, 4e-3), (torch.float16, 2e-4, 5e-4)])
@pytest.mark.parametrize("fused,layout,feat", [(False, "bgd", "nchw"), (True, "bgd", "nchw"), (True, "ndhwc", "nchw"), (True, "ndhwc", "nhwc")])
@pytest.mark.parametrize("case", [dict(B=2, C=32, G=16, h=24, w=40, D=12), dict(B=1, C=32, G=16, h=48, w=160, D=16),
                                  dict(B=1, C=32, G=16, h=24, w=40, D=13),   # odd slice: the unpaired tail of the 16-byte store path
                                  dict(B=1, C=16, G=8, h=24, w=40, D=11),   # two lanes per pixel
                                  # wild poses (an untrained pose network): with channels-last features the 2-byte kernels walk these
                                  # sub-slices in gather mode (16-byte loads from L2 forward, float atomics backward)
                                  dict(B=2, C=32, G=16, h=32, w=96, D=16, rot=0.3, trans=2.0),
                                  dict(B=2, C=64, G=16, h=32, w=96, D=16, rot=0.1, trans=1.0)])
def test_costvol_half_io_vs_oracle(ops, oracle_lib, case, fused, layout, feat, dtype, tol_rounded, tol_exact):
    """BASELINE configs 4 / 5 precision: bf16 or fp16 feature maps and volume, fp32 arithmetic in between.  The oracle gets
    the *rounded* features as floats; the kernel's output must equal the oracle's fp32 volume rounded to the format (a
    result within fp32 noise of a rounding boundary may land on the neighbouring value: norm-wise bound `tol_rounded`,
    about one format rounding step spread over few elements) and be within the format's own rounding of the exact one."""
    rng = np.random.default_rng(17)
    B, C, G, h, w, D = (case[k] for k in "BCGhwD")
    ref_t = torch.from_numpy(smooth_field(rng, (B, C, h, w), 3, -1, 1)).to(dtype)
    src_t = torch.from_numpy(smooth_field(rng, (B, C, h, w), 3, -1, 1)).to(dtype)
    ref, src = ref_t.float().numpy(), src_t.float().numpy()
    K, invK = kitti_K(h, w, B)
    prior = (2 + 20 * rng.random((B, 1, h, w))).astype(np.float32)
    pose = rand_pose(oracle_lib, rng, B, case.get("rot", 0.01), case.get("trans", 0.05))
    hyp = oracle_lib.schedule_depth_range(prior, D, 0.3, None, "inverse")
    gout_t = torch.from_numpy(rng.standard_normal((B, D, G, h, w)).astype(np.float32)).to(dtype)
    exp = oracle_lib.costvol_grouped(ref, src, K, invK, hyp, pose, G)
    exp_dref, exp_dsrc = oracle_lib.costvol_grouped_bwd(gout_t.float().numpy(), ref, src, K, invK, hyp, pose)
    r, s = feat_dev(ref, feat, dtype=dtype), feat_dev(src, feat, dtype=dtype)
    if fused:
        vol = ops.costvol_grouped(r, s, dev(K), dev(invK), dev(pose), G, prior=dev(prior), ndepth=D, scale_fac=0.3, layout=layout)
    else:
        vol = ops.costvol_grouped(r, s, dev(K), dev(invK), dev(pose), G, depth_priors=dev(hyp), layout=layout)
    assert vol.dtype == dtype and vol.shape == exp.shape
    got = vol.detach().float().cpu().numpy()
    assert relerr(got, torch.from_numpy(exp).to(dtype).float().numpy()) <= tol_rounded
    assert relerr(got, exp) <= tol_exact
    (vol.float() * gout_t.cuda().float()).sum().backward()
    assert r.grad.dtype == dtype and s.grad.dtype == dtype
    assert relerr(r.grad.float().cpu().numpy(), exp_dref) <= tol_exact
    assert relerr(s.grad.float().cpu().numpy(), exp_dsrc) <= tol_exact


def test_fuse_accepts_half_volumes(ops):
    """Config 5 (three lookup frames, fp16): the fusion kernels are fp32; 2-byte volumes must be widened, not misread."""
    torch.manual_seed(2)
    vols32 = [torch.randn(1, 8, 16, 6, 10, device="cuda") for _ in range(3)]
    vols16 = [v.half() for v in vols32]
    ref, _ = ops.fuse_volumes([v.half().float() for v in vols32], layout="bdg")
    got, _ = ops.fuse_volumes(vols16, layout="bdg")
    assert got.dtype == torch.float32
    assert_close(host(got), host(ref), rtol=1e-6)
