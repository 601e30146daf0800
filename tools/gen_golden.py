#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Runs ONLY in the build container (needs /root/reference, imported read-only via
tools/refload.py).  Every fixture is data: seeded synthetic inputs plus the
outputs / gradients the reference's own PyTorch-CPU code (torch 2.10 CPU, fp32)
produces for them.  No reference source text is stored.

    python tools/gen_golden.py            # rewrites tests/golden/*.npz

Reference functions driven (file:line in /root/reference/movedepth):
  layers.py   BackprojectDepth 556-586, Project3D 589-621, generate_costvol 778-794,
              schedule_depth_rangev2 256-284, schedule_depth_range_zv2 370-398,
              SSIM 646-677, get_smooth_loss 630-643, localmax 796-812, entropy 862-863,
              convex_upsample 200-214, disp_to_depth 400-409,
              transformation_from_parameters 412-429
  trainer.py  generate_images_pred 491-532, compute_reprojection_loss 535-550,
              compute_losses 614-724, compute_fuse_losses 569-612;
              the group-mean / confidence-fusion lines 358-363 are inline code in
              process_batch, so they are driven by executing the same torch ops on
              the reference's generate_costvol output (marked "inline" below).
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from refload import load_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


# ----------------------------------------------------------------------------- helpers
def kitti_K(h, w, B):
    """Normalised KITTI intrinsics scaled to (h, w); inv_K = pinv(K) (dataset contract a0)."""
    K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    K[0, :] *= w
    K[1, :] *= h
    invK = np.linalg.pinv(K)
    K = torch.from_numpy(np.repeat(K[None], B, 0).copy())
    invK = torch.from_numpy(np.repeat(invK[None], B, 0).astype(np.float32).copy())
    return K, invK


def smooth_noise(shape, g, coarse=4, lo=0.0, hi=1.0):
    """Low-pass noise: coarse uniform noise, bicubic-free bilinear upsample (well-conditioned taps)."""
    *lead, H, W = shape
    n = int(np.prod(lead))
    c = torch.rand(n, 1, max(2, H // coarse), max(2, W // coarse), generator=g)
    x = F.interpolate(c, size=(H, W), mode="bilinear", align_corners=True)
    return (lo + (hi - lo) * x).reshape(*shape).contiguous()


def npify(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    return out


def save(name, d):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **npify(d))
    print("%-28s %8.1f KB  (%d arrays)" % (name + ".npz", os.path.getsize(path) / 1024, len(d)))


def poses(L, B, g, rot=0.01, trans=0.05, tz=None, invert=False):
    aa = torch.randn(B, 1, 3, generator=g) * rot
    t = torch.randn(B, 1, 3, generator=g) * trans
    if tz is not None:
        t[:, 0, 2] = torch.as_tensor(tz, dtype=torch.float32)
    T = L.transformation_from_parameters(aa, t, invert=invert)
    return aa, t, T


# ----------------------------------------------------------------------------- fixtures
def gen_geometry(L):
    g = torch.Generator().manual_seed(101)
    B, h, w = 2, 6, 10
    K, invK = kitti_K(h, w, B)
    depth = 2 + 20 * torch.rand(B, 1, h, w, generator=g)
    aa, t, T = poses(L, B, g, rot=0.05, trans=0.3)
    _, _, Tinv = aa, t, L.transformation_from_parameters(aa, t, invert=True)
    bp, pj = L.BackprojectDepth(B, h, w), L.Project3D(B, h, w)
    pts = bp(depth, invK)
    pix = pj(pts, K, T)
    # KAT6: identity pose returns the align_corners grid (up to fp32 rounding)
    pix_id = pj(bp(depth, invK), K, torch.eye(4)[None].repeat(B, 1, 1))
    disp = torch.rand(B, 1, h, w, generator=g)
    sdisp, d2 = L.disp_to_depth(disp, 0.1, 100.0)
    save("geometry", dict(K=K, invK=invK, depth=depth, axisangle=aa, translation=t, T=T, T_invert=Tinv,
                          cam_points=pts, pix_coords=pix, pix_coords_identity=pix_id,
                          disp=disp, scaled_disp=sdisp, depth_from_disp=d2))


def gen_schedule(L):
    g = torch.Generator().manual_seed(102)
    B, h, w, D = 2, 5, 7, 8
    prior = 2 + 20 * torch.rand(B, 1, h, w, generator=g)
    out = dict(prior=prior, ndepth=D, scale_fac=0.3)
    z = torch.tensor([0.9, -1.2]).reshape(B, 1, 1, 1)  # z_scale * T[2,3]; second flips the ordering
    out["z_trans"] = z
    for ty in ("inverse", "linear", "log"):
        out["v2_" + ty] = L.schedule_depth_rangev2(prior, D, 0.3, type=ty)
        out["zv2_" + ty] = L.schedule_depth_range_zv2(prior, D, 0.3, z, type=ty)
    # no guard on 1 + f*z <= 0: negative / inf hypotheses are reproduced
    zbad = torch.tensor([-3.5, -10.0 / 3.0]).reshape(B, 1, 1, 1)
    out["z_trans_bad"] = zbad
    with np.errstate(all="ignore"):
        out["zv2_inverse_bad"] = L.schedule_depth_range_zv2(prior, D, 0.3, zbad, type="inverse")
    save("schedule", out)


def _fuse_inline(cvs, G):
    """trainer.py:349-363 executed on reference generate_costvol outputs ("inline")."""
    cor_weight_sum = 1e-8
    cor_feats = 0
    ws = []
    for cv in cvs:
        B, D, C, H, W = cv.shape
        cg = cv.reshape(B, D, -1, G, H, W).mean(2)
        wgt = torch.softmax(cg.mean(1), dim=1).max(1)[0]
        ws.append(wgt)
        cor_weight_sum = cor_weight_sum + wgt
        cor_feats = cor_feats + wgt.unsqueeze(1).unsqueeze(1) * cg
    cor_feats = cor_feats / cor_weight_sum.unsqueeze(1).unsqueeze(1)
    return cor_feats, ws


def gen_costvol(L):
    def case(tag, seed, B, C, G, h, w, D, rot, trans, tz, nframes=1, white=False, sched="v2", ztr=None):
        g = torch.Generator().manual_seed(seed)
        K, invK = kitti_K(h, w, B)
        if white:
            ref = torch.randn(B, C, h, w, generator=g)
            srcs = [torch.randn(B, C, h, w, generator=g) for _ in range(nframes)]
        else:
            ref = smooth_noise((B, C, h, w), g, coarse=3, lo=-1, hi=1)
            srcs = [smooth_noise((B, C, h, w), g, coarse=3, lo=-1, hi=1) for _ in range(nframes)]
        prior = 2 + 20 * torch.rand(B, 1, h, w, generator=g)
        Ts = [poses(L, B, g, rot=rot, trans=trans, tz=tz)[2] for _ in range(nframes)]
        pose = torch.stack(Ts, 1)  # B N 4 4
        if sched == "v2":
            hyp = L.schedule_depth_rangev2(prior, D, 0.3, type="inverse")
        else:
            hyp = L.schedule_depth_range_zv2(prior, D, 0.3, 30.0 * pose[:, :1, 2:3, -1:], type="inverse")
        ref.requires_grad_(True)
        for s in srcs:
            s.requires_grad_(True)
        bp, pj = L.BackprojectDepth(D, h, w), L.Project3D(D, h, w)
        cvs = [L.generate_costvol(ref, srcs[f], K, invK, hyp, pose[:, f:f + 1], D, bp, pj) for f in range(nframes)]
        cor, ws = _fuse_inline(cvs, G)
        Wt = torch.randn(cor.shape, generator=g)
        (cor * Wt).sum().backward()
        d = dict(K=K, invK=invK, ref=ref, prior=prior, pose=pose, hyp=hyp, G=G,
                 cor_feats=cor, grad_out=Wt, d_ref=ref.grad)
        for f in range(nframes):
            d["src%d" % f] = srcs[f]
            d["d_src%d" % f] = srcs[f].grad
            d["cor_weight%d" % f] = ws[f]
            # grouped per-frame volume (B,D,G,h,w); the ungrouped one only for the first, small, case
            d["grouped%d" % f] = cvs[f].reshape(B, D, -1, G, h, w).mean(2)
        if tag == "small":
            d["cost_vol_full0"] = cvs[0]
        save("costvol_" + tag, d)

    case("small", 201, B=2, C=32, G=16, h=8, w=16, D=6, rot=0.01, trans=0.05, tz=None)
    case("white", 202, B=1, C=32, G=16, h=8, w=16, D=6, rot=0.01, trans=0.05, tz=None, white=True)
    # strong motion: many taps leave the image; one sample has c_z <= 0 for near hypotheses
    case("oob", 203, B=2, C=32, G=16, h=8, w=16, D=6, rot=0.2, trans=1.5, tz=[-3.0, 2.0])
    case("zv2", 204, B=2, C=32, G=16, h=8, w=16, D=6, rot=0.01, trans=0.05, tz=[0.04, -0.03], sched="zv2")
    case("twoframe", 205, B=2, C=32, G=16, h=8, w=16, D=6, rot=0.02, trans=0.1, tz=None, nframes=2)
    case("c64g8", 206, B=1, C=64, G=8, h=6, w=10, D=5, rot=0.02, trans=0.1, tz=None)


def gen_costvol_launch(L):
    """generate_costvol + the group mean at the launch shape bench.py and the trainer run (BASELINE config 2: B=6, C=32, G=16, 48x160,
    D=96), with autograd, from the reference's own code.  Large tensors are rebuilt from a seed (tests/golden_inputs.py); the
    fixture keeps K, inv_K, the poses and small outputs: per-(b, d, g) plane sums and absolute sums of the grouped volume (float64),
    a lattice of its values, per-(b, c) sums / absolute sums and a lattice of d_ref and d_src."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    from golden_inputs import costvol_launch_inputs
    B, C, G, h, w, D = 6, 32, 16, 48, 160, 96
    ref, src, prior, gout = (torch.from_numpy(a) for a in costvol_launch_inputs(12, B, C, G, h, w, D))
    K, invK = kitti_K(h, w, B)
    g = torch.Generator().manual_seed(312)
    pose = poses(L, B, g, rot=0.01, trans=0.05)[2].unsqueeze(1)   # B 1 4 4
    hyp = L.schedule_depth_rangev2(prior, D, 0.3, type="inverse")
    ref.requires_grad_(True)
    src.requires_grad_(True)
    bp, pj = L.BackprojectDepth(D, h, w), L.Project3D(D, h, w)
    cv = L.generate_costvol(ref, src, K, invK, hyp, pose, D, bp, pj)          # B D C h w
    cor = cv.reshape(B, D, -1, G, h, w).mean(2)                                # trainer.py:358-359, one lookup frame
    (cor * gout).sum().backward()
    lat = lambda t: t[..., ::8, ::16]
    save("costvol_launch", dict(K=K, invK=invK, pose=pose, G=G,
                                vol_sum=cor.double().sum((-1, -2)), vol_abs_sum=cor.double().abs().sum((-1, -2)), vol_lattice=lat(cor)[:, ::8, ::2],
                                d_ref_sum=ref.grad.double().sum((-1, -2)), d_ref_abs_sum=ref.grad.double().abs().sum((-1, -2)), d_ref_lattice=lat(ref.grad),
                                d_src_sum=src.grad.double().sum((-1, -2)), d_src_abs_sum=src.grad.double().abs().sum((-1, -2)), d_src_lattice=lat(src.grad)))


def gen_warp(L):
    def case(tag, seed, rot, trans):
        g = torch.Generator().manual_seed(seed)
        B, H, W = 2, 16, 32
        K, invK = kitti_K(H, W, B)
        img = smooth_noise((B, 3, H, W), g, coarse=4)
        depth = (2 + 20 * smooth_noise((B, 1, H, W), g, coarse=4)).requires_grad_(True)
        aa, t, T = poses(L, B, g, rot=rot, trans=trans)
        T = T.clone().requires_grad_(True)
        bp, pj = L.BackprojectDepth(B, H, W), L.Project3D(B, H, W)
        pix = pj(bp(depth, invK), K, T)
        warped = F.grid_sample(img, pix, padding_mode="border", align_corners=True)
        mvs_mask = ((pix < -1) | (pix > 1)).sum(-1) > 0  # trainer.py:503
        Wt = torch.randn(warped.shape, generator=g)
        (warped * Wt).sum().backward()
        save("warp_" + tag, dict(K=K, invK=invK, img=img, depth=depth, T=T, pix_coords=pix, warped=warped,
                                 mvs_mask=mvs_mask, grad_out=Wt, d_depth=depth.grad, d_T=T.grad))

    case("small", 301, rot=0.01, trans=0.1)
    case("border", 302, rot=0.1, trans=2.0)  # many samples clamp to the border (zero grid-grad there)


def gen_warp_fullres(L):
    """The warp and its autograd at 192 x 640 (VERDICT r3 missing #4: the small warp fixtures leave the pose gradient at full
    resolution to the oracle alone).  The large tensors are rebuilt from a seed by tests/golden_inputs.py (numpy only); the fixture
    keeps K, inv_K, T and the SMALL outputs of the reference's own BackprojectDepth / Project3D / grid_sample / autograd: d_T, the
    scalar sum(warped * gout), per-row sums of d_depth, of the warped frame and of the sample grid, the out-of-view count."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    from golden_inputs import warp_fullres_inputs
    B, H, W = 2, 192, 640
    img, depth, gout = (torch.from_numpy(a) for a in warp_fullres_inputs(11, B, H, W))
    K, invK = kitti_K(H, W, B)
    g = torch.Generator().manual_seed(311)
    aa, t, T = poses(L, B, g, rot=0.01, trans=0.1)
    depth = depth.clone().requires_grad_(True)
    T = T.clone().requires_grad_(True)
    bp, pj = L.BackprojectDepth(B, H, W), L.Project3D(B, H, W)
    pix = pj(bp(depth, invK), K, T)
    warped = F.grid_sample(img, pix, padding_mode="border", align_corners=True)
    mvs_mask = ((pix < -1) | (pix > 1)).sum(-1) > 0
    loss = (warped * gout).sum()
    loss.backward()
    save("warp_fullres", dict(K=K, invK=invK, T=T, d_T=T.grad, loss=loss.double(), mask_count=mvs_mask.sum(),
                              d_depth_rowsum=depth.grad.double().sum(-1), d_depth_abs_rowsum=depth.grad.double().abs().sum(-1),
                              warped_rowsum=warped.double().sum(-1), pix_rowsum=pix.double().sum(2),
                              d_depth_sample=depth.grad[:, :, ::16, ::16], warped_sample=warped[:, :, ::16, ::16], pix_sample=pix[:, ::16, ::16]))


def gen_ssim(L, Trainer):
    g = torch.Generator().manual_seed(401)
    B, H, W = 2, 16, 32
    x = smooth_noise((B, 3, H, W), g, coarse=2).requires_grad_(True)
    y = smooth_noise((B, 3, H, W), g, coarse=2)
    ssim = L.SSIM()
    s = ssim(x, y)
    t = Trainer.__new__(Trainer)
    t.opt = types.SimpleNamespace(no_ssim=False, ssim_lw=0.85)
    t.ssim = ssim
    rl = t.compute_reprojection_loss(x, y)
    Wt = torch.randn(rl.shape, generator=g)
    (rl * Wt).sum().backward()
    d_pred = x.grad.clone()
    x.grad = None
    rl0 = t.compute_reprojection_loss(x, y, ssim_lw=0)
    (rl0 * Wt).sum().backward()
    # KAT5
    ones, zeros = torch.ones(1, 3, 8, 8), torch.zeros(1, 3, 8, 8)
    xr = torch.rand(1, 3, 8, 8, generator=g)
    save("ssim", dict(pred=x, target=y, ssim=s, reproj=rl, grad_out=Wt, d_pred=d_pred,
                      reproj_l1only=rl0, d_pred_l1only=x.grad,
                      kat_same=ssim(xr, xr), kat_const=ssim(zeros, ones), kat_x=xr))


def _make_trainer(L, Trainer, B, H, W, D=8, **flags):
    t = Trainer.__new__(Trainer)
    opt = dict(no_ssim=False, ssim_lw=0.85, scales=[0, 1, 2, 3], frame_ids=[0, -1, 1], height=H, width=W,
               min_depth=0.1, max_depth=100.0, disable_automasking=False, disparity_smoothness=1e-3,
               mask_mvs_auto=False, mask_mvs_conf=False, mask_mvs_dist=False, mask_mvs_geo=False,
               mvs_smooth_loss=False)
    opt.update(flags)
    t.opt = types.SimpleNamespace(**opt)
    t.device = torch.device("cpu")
    t.num_scales = 4
    t.ssim = L.SSIM()
    t.backproject_depth = {s: L.BackprojectDepth(B, H // 2 ** s, W // 2 ** s) for s in range(4)}
    t.project_3d = {s: L.Project3D(B, H // 2 ** s, W // 2 ** s) for s in range(4)}
    return t


def _inputs(B, H, W, g):
    inputs = {}
    for f in (0, -1, 1):
        base = smooth_noise((B, 3, H, W), g, coarse=4)
        for s in range(4):
            inputs[("color", f, s)] = F.interpolate(base, size=(H // 2 ** s, W // 2 ** s), mode="bilinear",
                                                    align_corners=False)
            inputs[("color_aug", f, s)] = inputs[("color", f, s)]
    for s in range(4):
        inputs[("K", s)], inputs[("inv_K", s)] = kitti_K(H // 2 ** s, W // 2 ** s, B)
    return inputs


def gen_losses(L, Trainer):
    B, H, W = 2, 32, 64
    g = torch.Generator().manual_seed(501)
    inputs = _inputs(B, H, W, g)
    fx = {}
    for (k, v) in inputs.items():
        if k[0] in ("color", "K", "inv_K"):
            fx["in_" + "_".join(str(x) for x in k)] = v
    # ---- mono branch: generate_images_pred + compute_losses (trainer.py:510-532, 675-724)
    t = _make_trainer(L, Trainer, B, H, W)
    disps = {s: (0.004 + 0.1 * smooth_noise((B, 1, H // 2 ** s, W // 2 ** s), g, coarse=4)).requires_grad_(True)
             for s in range(4)}
    aa = {f: (torch.randn(B, 1, 3, generator=g) * 0.01).requires_grad_(True) for f in (-1, 1)}
    tr = {f: (torch.randn(B, 1, 3, generator=g) * 0.05).requires_grad_(True) for f in (-1, 1)}
    outputs = {("disp", s): disps[s] for s in range(4)}
    for f in (-1, 1):
        outputs[("cam_T_cam", 0, f)] = L.transformation_from_parameters(aa[f], tr[f], invert=(f < 0))
    t.generate_images_pred(inputs, outputs)
    torch.manual_seed(777)  # automask tie-break noise, trainer.py:698 (4 draws of (B,1,H,W), CPU generator)
    losses = t.compute_losses(inputs, outputs)
    losses["loss"].backward()
    d = dict(fx)
    d["noise_seed"] = 777
    for s in range(4):
        d["disp_%d" % s] = disps[s]
        d["d_disp_%d" % s] = disps[s].grad
        d["loss_%d" % s] = losses["loss/%d" % s]
        d["smooth_%d" % s] = losses["mono_smooth_loss/%d" % s]
        d["depth_0_%d" % s] = outputs[("depth", 0, s)]
    for f in (-1, 1):
        n = "m1" if f < 0 else "p1"
        d["axisangle_" + n], d["translation_" + n] = aa[f], tr[f]
        d["d_axisangle_" + n], d["d_translation_" + n] = aa[f].grad, tr[f].grad
        d["T_" + n] = outputs[("cam_T_cam", 0, f)]
        d["sample_%s_0" % n] = outputs[("sample", f, 0)]
        d["color_%s_0" % n] = outputs[("color", f, 0)]
        d["color_%s_3" % n] = outputs[("color", f, 3)]
    d["loss"] = losses["loss"]
    d["mono_reproj_loss"] = outputs["mono_reproj_loss"]
    # The reference's OWN float32-vs-float64 distance of the pose / disparity gradients (same code, same seeds, tensors and
    # geometry modules in double): dL/dT sums ~4000 signed per-pixel terms that largely cancel, so float32 gradients sit
    # this far from the exact ones whoever computes them; the GPU test bounds its error by 3x these figures.
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        t64 = _make_trainer(L, Trainer, B, H, W)
        for m in [t64.ssim] + list(t64.backproject_depth.values()) + list(t64.project_3d.values()):
            m.double()
        in64 = {k: v.double() for k, v in inputs.items()}
        disps64 = {s: disps[s].detach().double().requires_grad_(True) for s in range(4)}
        aa64 = {f: aa[f].detach().double().requires_grad_(True) for f in (-1, 1)}
        tr64 = {f: tr[f].detach().double().requires_grad_(True) for f in (-1, 1)}
        out64 = {("disp", s): disps64[s] for s in range(4)}
        for f in (-1, 1):
            out64[("cam_T_cam", 0, f)] = L.transformation_from_parameters(aa64[f], tr64[f], invert=(f < 0))
        t64.generate_images_pred(in64, out64)
        torch.manual_seed(777)
        orig_randn = torch.randn
        torch.randn = lambda *a_, **k_: orig_randn(*a_, dtype=torch.float32, **k_).double()
        try:
            l64 = t64.compute_losses(in64, out64)
        finally:
            torch.randn = orig_randn
        l64["loss"].backward()
    finally:
        torch.set_default_dtype(prev)

    def _rel(a_, b_):
        return float((a_.double() - b_).norm() / b_.norm())

    for f in (-1, 1):
        n = "m1" if f < 0 else "p1"
        d["noise_d_axisangle_" + n] = _rel(aa[f].grad, aa64[f].grad)
        d["noise_d_translation_" + n] = _rel(tr[f].grad, tr64[f].grad)
    for s in range(4):
        d["noise_d_disp_%d" % s] = _rel(disps[s].grad, disps64[s].grad)
    d["noise_loss"] = abs(float(losses["loss"]) - float(l64["loss"])) / float(l64["loss"])
    print("reference float32-vs-float64:", {k: "%.1e" % v for k, v in d.items() if k.startswith("noise_")})
    save("losses_mono", d)

    # ---- mono branch without automasking
    t2 = _make_trainer(L, Trainer, B, H, W, disable_automasking=True)
    outputs2 = {("disp", s): disps[s].detach() for s in range(4)}
    for f in (-1, 1):
        outputs2[("cam_T_cam", 0, f)] = outputs[("cam_T_cam", 0, f)].detach()
    t2.generate_images_pred(inputs, outputs2)
    l2 = t2.compute_losses(inputs, outputs2)
    save("losses_mono_noautomask", dict(loss=l2["loss"], **{"loss_%d" % s: l2["loss/%d" % s] for s in range(4)}))

    # ---- MVS + fuse branches (trainer.py:495-508, 621-673, 569-612)
    for tag, flags in (("default", {}), ("auto_smooth", dict(mask_mvs_auto=True, mvs_smooth_loss=True))):
        g2 = torch.Generator().manual_seed(502)
        t3 = _make_trainer(L, Trainer, B, H, W, **flags)
        depth_mvs = (2 + 20 * smooth_noise((B, H, W), g2, coarse=4)).requires_grad_(True)
        mono_depth = 2 + 20 * smooth_noise((B, 1, H, W), g2, coarse=4)
        trust = smooth_noise((B, 1, H, W), g2, coarse=4).requires_grad_(True)
        out3 = {"depth_mvs": depth_mvs}
        for f in (-1, 1):
            out3[("cam_T_cam", 0, f)] = outputs[("cam_T_cam", 0, f)].detach()
        fused = (1 - trust) * depth_mvs[:, None].detach() + trust * mono_depth  # trainer.py:413
        out3["fused_depth"] = fused
        torch.manual_seed(778)
        fuse_losses = t3.compute_fuse_losses(inputs, out3)
        t3.generate_images_pred(inputs, out3, is_mvs=True)
        mvs_losses = t3.compute_losses(inputs, out3, is_mvs=True)
        (mvs_losses["loss"] + fuse_losses["loss"]).backward()
        d3 = dict(noise_seed=778, depth_mvs=depth_mvs, mono_depth=mono_depth, trust_mono_mask=trust,
                  T_m1=out3[("cam_T_cam", 0, -1)], T_p1=out3[("cam_T_cam", 0, 1)],
                  mvs_loss=mvs_losses["loss"], fuse_loss=fuse_losses["loss"],
                  fuse_reproj_loss=fuse_losses["fuse_reproj_loss"],
                  mvs_reprojection_loss=out3["mvs_reprojection_loss"], mvs_reproj_loss=out3["mvs_reproj_loss"],
                  mvs_color_m1=out3[("mvs_color", -1)], mvs_mask_m1=out3[("mvs_mask", -1)],
                  mvs_color_fuse_p1=out3[("mvs_color_fuse", 1)],
                  d_depth_mvs=depth_mvs.grad, d_trust=trust.grad)
        if "mvs_smooth_loss/0" in mvs_losses:
            d3["mvs_smooth_loss"] = mvs_losses["mvs_smooth_loss/0"]
        save("losses_mvs_" + tag, d3)


def gen_losses_fullres(L, Trainer):
    """generate_images_pred + compute_losses + backward of the reference's own Trainer at 192 x 640 (the size bench.py runs), B = 2,
    mono branch with auto-masking.  Images and disparities are rebuilt from a seed (tests/golden_inputs.py); the fixture keeps the
    pose parameters and small outputs: losses, pose gradients, row sums / absolute row sums and a lattice of the disparity
    gradients, lattices of depth, sample grid, warped frames and of the per-pixel minimum loss."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    from golden_inputs import losses_fullres_inputs
    B, H, W = 2, 192, 640
    colors, disps_np = losses_fullres_inputs(13, B, H, W)
    inputs = {}
    for (f, s), v in colors.items():
        inputs[("color", f, s)] = inputs[("color_aug", f, s)] = torch.from_numpy(v)
    for s in range(4):
        inputs[("K", s)], inputs[("inv_K", s)] = kitti_K(H // 2 ** s, W // 2 ** s, B)
    g = torch.Generator().manual_seed(521)
    t = _make_trainer(L, Trainer, B, H, W)
    disps = {s: torch.from_numpy(disps_np[s]).requires_grad_(True) for s in range(4)}
    aa = {f: (torch.randn(B, 1, 3, generator=g) * 0.01).requires_grad_(True) for f in (-1, 1)}
    tr = {f: (torch.randn(B, 1, 3, generator=g) * 0.05).requires_grad_(True) for f in (-1, 1)}
    outputs = {("disp", s): disps[s] for s in range(4)}
    for f in (-1, 1):
        outputs[("cam_T_cam", 0, f)] = L.transformation_from_parameters(aa[f], tr[f], invert=(f < 0))
    t.generate_images_pred(inputs, outputs)
    torch.manual_seed(778)
    losses = t.compute_losses(inputs, outputs)
    losses["loss"].backward()
    lat = lambda x: x[..., ::8, ::16]
    d = dict(noise_seed=778, loss=losses["loss"], mono_reproj_lattice=lat(outputs["mono_reproj_loss"]),
             mono_reproj_sum=outputs["mono_reproj_loss"].double().sum())
    for s in range(4):
        d["K_%d" % s], d["inv_K_%d" % s] = inputs[("K", s)], inputs[("inv_K", s)]
        d["loss_%d" % s], d["smooth_%d" % s] = losses["loss/%d" % s], losses["mono_smooth_loss/%d" % s]
        d["d_disp_rowsum_%d" % s] = disps[s].grad.double().sum(-1)
        d["d_disp_abs_rowsum_%d" % s] = disps[s].grad.double().abs().sum(-1)
        d["d_disp_lattice_%d" % s] = disps[s].grad[..., ::4, ::8]
        d["depth_lattice_%d" % s] = lat(outputs[("depth", 0, s)])
    for f in (-1, 1):
        n = "m1" if f < 0 else "p1"
        d["axisangle_" + n], d["translation_" + n] = aa[f], tr[f]
        d["d_axisangle_" + n], d["d_translation_" + n] = aa[f].grad, tr[f].grad
        d["T_" + n] = outputs[("cam_T_cam", 0, f)]
        for s in (0, 3):
            d["sample_lattice_%s_%d" % (n, s)] = outputs[("sample", f, s)][:, ::8, ::16]
            d["color_lattice_%s_%d" % (n, s)] = lat(outputs[("color", f, s)])
    save("losses_mono_fullres", d)

    # ---- MVS + fused-depth branches at the same size (trainer.py:495-508, 621-673, 569-612), masks and MVS smoothness on
    from golden_inputs import mvs_fullres_inputs
    t3 = _make_trainer(L, Trainer, B, H, W, mask_mvs_auto=True, mvs_smooth_loss=True)
    dm, md, tm = mvs_fullres_inputs(14, B, H, W)
    depth_mvs, mono_depth, trust = torch.from_numpy(dm).requires_grad_(True), torch.from_numpy(md), torch.from_numpy(tm).requires_grad_(True)
    out3 = {"depth_mvs": depth_mvs}
    for f in (-1, 1):
        out3[("cam_T_cam", 0, f)] = outputs[("cam_T_cam", 0, f)].detach()
    out3["fused_depth"] = (1 - trust) * depth_mvs[:, None].detach() + trust * mono_depth  # trainer.py:413
    torch.manual_seed(779)
    fuse_losses = t3.compute_fuse_losses(inputs, out3)
    t3.generate_images_pred(inputs, out3, is_mvs=True)
    mvs_losses = t3.compute_losses(inputs, out3, is_mvs=True)
    (mvs_losses["loss"] + fuse_losses["loss"]).backward()
    rs = lambda x: x.double().sum(-1)
    save("losses_mvs_fullres", dict(noise_seed=779, T_m1=out3[("cam_T_cam", 0, -1)], T_p1=out3[("cam_T_cam", 0, 1)],
                                    K_0=inputs[("K", 0)], inv_K_0=inputs[("inv_K", 0)],
                                    mvs_loss=mvs_losses["loss"], fuse_loss=fuse_losses["loss"], fuse_reproj_loss=fuse_losses["fuse_reproj_loss"],
                                    mvs_reproj_loss=out3["mvs_reproj_loss"], mvs_smooth_loss=mvs_losses["mvs_smooth_loss/0"],
                                    mvs_reprojection_lattice=lat(out3["mvs_reprojection_loss"]), mvs_color_lattice_m1=lat(out3[("mvs_color", -1)]),
                                    mvs_color_fuse_lattice_p1=lat(out3[("mvs_color_fuse", 1)]), mvs_mask_count_m1=out3[("mvs_mask", -1)].sum(),
                                    d_depth_mvs_rowsum=rs(depth_mvs.grad), d_depth_mvs_abs_rowsum=rs(depth_mvs.grad.abs()),
                                    d_depth_mvs_lattice=depth_mvs.grad[..., ::4, ::8],
                                    d_trust_rowsum=rs(trust.grad), d_trust_abs_rowsum=rs(trust.grad.abs()), d_trust_lattice=trust.grad[..., ::4, ::8]))


def gen_smooth(L):
    g = torch.Generator().manual_seed(601)
    B, H, W = 2, 12, 20
    disp = (0.05 + 0.9 * torch.rand(B, 1, H, W, generator=g)).requires_grad_(True)
    img = smooth_noise((B, 3, H, W), g, coarse=2)
    mean_disp = disp.mean(2, True).mean(3, True)  # trainer.py:712-714
    norm = disp / (mean_disp + 1e-7)
    sl = L.get_smooth_loss(norm, img)
    sl.backward()
    raw = L.get_smooth_loss(disp.detach(), img)
    save("smooth", dict(disp=disp, img=img, smooth_norm=sl, d_disp=disp.grad, smooth_raw=raw))


def gen_postvol(L):
    g = torch.Generator().manual_seed(701)
    B, D, h, w = 2, 8, 6, 10
    logits = (torch.randn(B, D, h, w, generator=g) * 2).requires_grad_(True)
    prob = F.softmax(logits, 1)
    prior = 2 + 20 * torch.rand(B, 1, h, w, generator=g)
    hyp = L.schedule_depth_rangev2(prior, D, 0.3)
    ent = L.entropy(prob, dim=1, keepdim=True)
    dm = L.localmax(prob, 1, D, 1 / hyp[:, -1], 1 / hyp[:, 0])  # swapped endpoints, trainer.py:371 (KAT4)
    Wd = torch.randn(dm.shape, generator=g)
    We = torch.randn(ent.shape, generator=g)
    ((dm * Wd).sum() + (ent * We).sum()).backward()
    dm2 = L.localmax(prob.detach(), 2, D, 1 / hyp[:, -1], 1 / hyp[:, 0])
    # KAT4: one-hot at d decodes to hypothesis D-1-d
    onehot = torch.zeros(1, D, 1, D)
    for d_ in range(D):
        onehot[0, d_, 0, d_] = 1
    hyp1 = hyp[:1, :, :1, :1].repeat(1, 1, 1, D)
    kat = L.localmax(onehot, 1, D, 1 / hyp1[:, -1], 1 / hyp1[:, 0])
    # convex upsample (layers.py:200-214), scale 2 -> 4x
    depth = 2 + 20 * torch.rand(B, h, w, generator=g)
    depth.requires_grad_(True)
    mask = torch.randn(B, 16 * 9, h, w, generator=g).requires_grad_(True)
    up = L.convex_upsample(depth, mask, 2)
    Wu = torch.randn(up.shape, generator=g)
    (up * Wu).sum().backward()
    save("postvol", dict(logits=logits, hyp=hyp, entropy=ent, depth_r1=dm, depth_r2=dm2, grad_depth=Wd,
                         grad_entropy=We, d_logits=logits.grad, kat_onehot_depth=kat, kat_hyp=hyp1,
                         up_depth=depth, up_mask=mask, up_out=up, up_grad=Wu, d_up_depth=depth.grad,
                         d_up_mask=mask.grad))


def gen_postvol_launch(L):
    """softmax -> entropy / localmax (trainer.py:366-371) and convex_upsample (layers.py:200-214) with autograd at BASELINE config 2's
    launch shape (B=6, D=96, 48x160 -> 192x640), from the reference's own functions; inputs rebuilt from a seed
    (tests/golden_inputs.py), the fixture keeps row sums / absolute row sums and lattices."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    from golden_inputs import postvol_launch_inputs
    B, D, h, w = 6, 96, 48, 160
    logits, prior, g_depth, g_ent, up_depth, up_mask, g_up = (torch.from_numpy(a) for a in postvol_launch_inputs(15, B, D, h, w))
    logits.requires_grad_(True)
    prob = F.softmax(logits, 1)
    hyp = L.schedule_depth_rangev2(prior, D, 0.3)
    ent = L.entropy(prob, dim=1, keepdim=True)
    dm = L.localmax(prob, 1, D, 1 / hyp[:, -1], 1 / hyp[:, 0])
    ((dm * g_depth).sum() + (ent * g_ent).sum()).backward()
    up_depth.requires_grad_(True)
    up_mask.requires_grad_(True)
    up = L.convex_upsample(up_depth, up_mask, 2)
    (up * g_up).sum().backward()
    rs = lambda x: x.double().sum(-1)
    lat = lambda x: x[..., ::8, ::16]
    save("postvol_launch", dict(inv_hi_lattice=lat(1 / hyp[:, -1]), inv_lo_lattice=lat(1 / hyp[:, 0]),
                                depth_rowsum=rs(dm), depth_lattice=lat(dm), entropy_rowsum=rs(ent), entropy_lattice=lat(ent),
                                d_logits_planesum=logits.grad.double().sum((-1, -2)), d_logits_abs_planesum=logits.grad.double().abs().sum((-1, -2)), d_logits_lattice=lat(logits.grad)[:, ::8],
                                up_rowsum=rs(up), up_lattice=up[..., ::16, ::32],
                                d_up_depth_rowsum=rs(up_depth.grad), d_up_depth_abs_rowsum=rs(up_depth.grad.abs()), d_up_depth_lattice=lat(up_depth.grad),
                                d_up_mask_rowsum=rs(up_mask.grad)[:, ::9], d_up_mask_abs_rowsum=rs(up_mask.grad.abs())[:, ::9],
                                d_up_mask_lattice=lat(up_mask.grad)[:, ::12]))


def gen_prob_conv(networks):
    """reg3d's last layer through the reference's own module (networks/resnet_encoder.py:254, applied :277):
    prob(x).squeeze(1) on the tensor the U-Net hands it, with gradients to that tensor and to the weight.
    Ragged sizes (not multiples of the kernel's 8x32 tile, D not a multiple of its slices)."""
    g = torch.Generator().manual_seed(811)
    for tag, (B, C, D, H, W) in (("c16", (2, 16, 11, 10, 37)), ("c8", (1, 8, 5, 9, 33))):
        torch.manual_seed(812)
        net = networks.reg3d(C, C, down_size=1)
        x = torch.randn(B, C, D, H, W, generator=g).requires_grad_(True)
        y = net.prob(x).squeeze(1)
        Wy = torch.randn(y.shape, generator=g)
        (y * Wy).sum().backward()
        save("prob_conv_" + tag, dict(x=x, weight=net.prob.weight, y=y, grad_out=Wy, d_x=x.grad,
                                      d_weight=net.prob.weight.grad))


def gen_pose_grad(L):
    """transformation_from_parameters through the reference with gradients (both invert modes), incl. a large angle and
    a nearly-zero rotation."""
    g = torch.Generator().manual_seed(831)
    B = 6
    out = {}
    aa = torch.randn(B, 1, 3, generator=g) * torch.tensor([0.01, 0.05, 0.3, 1.0, 2.5, 1e-4]).view(B, 1, 1)
    tr = torch.randn(B, 1, 3, generator=g)
    out["axisangle"], out["translation"] = aa, tr
    for inv in (False, True):
        a, t = aa.clone().requires_grad_(True), tr.clone().requires_grad_(True)
        T = L.transformation_from_parameters(a, t, invert=inv)
        W = torch.randn(T.shape, generator=g)
        (T * W).sum().backward()
        k = "_inv" if inv else ""
        out["T" + k], out["grad_T" + k], out["d_axisangle" + k], out["d_translation" + k] = T, W, a.grad, t.grad
    save("pose_grad", out)


def gen_conv0(networks):
    """reg3d's first convolution through the reference's own module (conv0.conv, networks/resnet_encoder.py:231,
    applied :258 on the permuted volume): output and both gradients.  Ragged size."""
    g = torch.Generator().manual_seed(821)
    torch.manual_seed(822)
    net = networks.reg3d(16, 16, down_size=1)
    x = torch.randn(1, 16, 6, 9, 35, generator=g).requires_grad_(True)
    y = net.conv0.conv(x)
    Wy = torch.randn(y.shape, generator=g)
    (y * Wy).sum().backward()
    save("conv0_c16", dict(x=x, weight=net.conv0.conv.weight, y=y, grad_out=Wy, d_x=x.grad,
                           d_weight=net.conv0.conv.weight.grad))


def main():
    torch.set_num_threads(1)
    L, Trainer, networks = load_reference(with_trainer=True)
    gen_prob_conv(networks)
    gen_conv0(networks)
    gen_pose_grad(L)
    gen_geometry(L)
    gen_schedule(L)
    gen_costvol(L)
    gen_costvol_launch(L)
    gen_warp(L)
    gen_warp_fullres(L)
    gen_ssim(L, Trainer)
    gen_losses(L, Trainer)
    gen_losses_fullres(L, Trainer)
    gen_smooth(L)
    gen_postvol(L)
    gen_postvol_launch(L)


if __name__ == "__main__":
    main()
