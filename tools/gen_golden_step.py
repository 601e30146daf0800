#!/usr/bin/env python3
"""Golden fixtures for the whole training step: the reference's own Trainer.process_batch + backward
(/root/reference/movedepth/trainer.py:297-442) and its evaluation forward (evaluate_depth.py:208-250), run on the
CPU in THIS container (torch 2.10, fp32), for tests/test_step_golden.py.

    python tools/gen_golden_step.py        # rewrites tests/golden/step_*.npz, eval_*.npz, ckpt_manifest.json

How the reference Trainer is built: stub modules for cv2 / tensorboardX / pykitti / skimage, a
torchvision stand-in whose ResNet classes are assembled from this repo's own ResNet blocks (tools/refload.py), then
`Trainer.__new__` + the attributes `process_batch` reads.  The sub-models are the REFERENCE's own `networks.*` classes;
their weights come from this repo's `build_models` under a fixed seed (identical state_dict keys and shapes -- asserted
below -- so the 113 MB of weights need not be committed: the test rebuilds them from the same seed on the CPU and the
fixture carries per-tensor checksums to prove it got the same numbers).

Stored per case: every loss-dict entry, the output maps the hot path produces (depth_mvs, masked_depth, fused_depth,
trust_mono_mask, photo_conf_map, disparities, poses, erase rectangle), per-parameter gradient norms of every sub-model
and full gradients of the small tensors.  Inputs (frames, intrinsics) are stored once in step_inputs.npz.
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from refload import load_reference  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

from step_fixture import (B, H, W, D, STEP_SEED, WEIGHT_SEED, BASE_ARGS, CASES, build_weights, make_frames, checksums,  # noqa: E402
                          formula_state)


def save(name, d):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(path, **out)
    print("%-28s %8.1f KB  (%d arrays)" % (name + ".npz", os.path.getsize(path) / 1024, len(out)))


def ref_trainer(L, Trainer, networks, ref_opt, build_models_dict):
    t = Trainer.__new__(Trainer)
    t.opt = ref_opt
    t.device = torch.device("cpu")
    t.num_scales = len(ref_opt.scales)
    t.matching_ids = ref_opt.matching_ids
    t.ssim = L.SSIM()
    fh, fw = H // 4, W // 4
    t.backprojector, t.projector = L.BackprojectDepth(D, fh, fw), L.Project3D(D, fh, fw)
    t.backproject_depth = {s: L.BackprojectDepth(B, H // 2 ** s, W // 2 ** s) for s in ref_opt.scales}
    t.project_3d = {s: L.Project3D(B, H // 2 ** s, W // 2 ** s) for s in ref_opt.scales}
    m = {}
    m["mono_encoder"] = networks.ResnetEncoder(ref_opt.res_arch, False)
    m["mono_depth"] = networks.DepthDecoder(m["mono_encoder"].num_ch_enc, ref_opt.scales, match_conv=False, ddv=False,
                                            discret=None, mono_conf=False, mono_bins=False)
    m["pose_encoder"] = networks.ResnetEncoder(ref_opt.res_arch, False, num_input_images=2)
    m["pose"] = networks.PoseDecoder(m["pose_encoder"].num_ch_enc, num_input_features=1, num_frames_to_predict_for=2)
    m["mask_cnn"] = networks.UncertNet()
    m["mvs_encoder"] = networks.FPN4(base_channels=8, scale=ref_opt.prior_scale, dcn=False)
    m["reg3d"] = networks.reg3d(in_channels=ref_opt.reg3d_c, base_channels=ref_opt.reg3d_c, down_size=3)
    m["up"] = L.convex_upsample_layer(feature_dim=8 * 2 ** ref_opt.prior_scale, scale=ref_opt.prior_scale)
    for k in m:
        want, have = build_models_dict[k].state_dict(), m[k].state_dict()
        assert list(want.keys()) == list(have.keys()), "state_dict keys of %s differ from the reference's" % k
        for kk in want:
            assert want[kk].shape == have[kk].shape, (k, kk)
        m[k].load_state_dict(want, strict=True)
        m[k].train()
    t.models = m
    return t


def parse_ref_options(extra):
    from movedepth.options import MonodepthOptions

    argv = sys.argv
    sys.argv = ["x", "--no_cuda"] + BASE_ARGS + list(extra)
    try:
        return MonodepthOptions().parse()
    finally:
        sys.argv = argv


FULL_GRADS = (("mask_cnn", "head_convs.weight"), ("mask_cnn", "conv1.0.weight"), ("up", "upsample_mask.2.weight"),
              ("pose", "net.3.weight"), ("pose", "net.3.bias"), ("reg3d", "prob.weight"), ("reg3d", "conv0.conv.weight"),
              ("mvs_encoder", "out.weight"), ("mono_depth", "decoder.10.conv.weight"),
              ("mono_encoder", "encoder.conv1.weight"), ("pose_encoder", "encoder.conv1.weight"))


def _run_reference(epoch, extra, L, Trainer, networks, frames, dtype):
    """One reference process_batch + backward in `dtype`.  float64 is the reference's own arithmetic carried in double
    (same seeds, the float32 tie-break noise widened) -- used only to measure how far its float32 gradients sit from
    the exact ones, which bounds what any other float32 implementation can be asked to reproduce."""
    _, models = build_weights(extra)
    ref_opt = parse_ref_options(extra)
    prev, orig_randn = torch.get_default_dtype(), torch.randn
    torch.set_default_dtype(dtype)
    try:
        t = ref_trainer(L, Trainer, networks, ref_opt, models)
        t.epoch = epoch
        if dtype == torch.float64:
            torch.randn = lambda *a, **k: orig_randn(*a, dtype=torch.float32, **k).double()
            for m in list(t.models.values()) + [t.backprojector, t.projector, t.ssim] + list(t.backproject_depth.values()) + \
                    list(t.project_3d.values()):
                m.double()
        inputs = {k: v.clone().to(dtype) for k, v in frames.items()}
        torch.manual_seed(STEP_SEED)
        np.random.seed(STEP_SEED)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            outputs, losses = t.process_batch(inputs, is_train=True)
            losses["loss"].backward()
    finally:
        torch.set_default_dtype(prev)
        torch.randn = orig_randn
    return t, inputs, outputs, losses


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def run_case(tag, epoch, extra, L, Trainer, networks, frames):
    t, inputs, outputs, losses = _run_reference(epoch, extra, L, Trainer, networks, frames, torch.float32)
    t64, _, _, _ = _run_reference(epoch, extra, L, Trainer, networks, frames, torch.float64)
    d = {"epoch": epoch, "flags": " ".join(extra), "step_seed": STEP_SEED, "weight_seed": WEIGHT_SEED}
    for k, v in losses.items():
        d["loss:" + k] = v
    for k in ("depth_mvs", "masked_depth", "fused_depth", "trust_mono_mask", "mono_reproj_loss", "mvs_reprojection_loss",
              "mvs_reproj_loss", "reprojection_loss_mask"):
        d["out:" + k] = outputs[k]
    for k in ("photo_conf_map", "dist_mask"):
        if k in outputs:
            d["out:" + k] = outputs[k]
    aug = outputs["masked_aug"][0, 0]                       # the erase rectangle (np.random, layers.py:64-65)
    ys, xs = np.where(aug.numpy() == 0)
    d["erase_rect"] = np.array([ys.min(), ys.max() + 1, xs.min(), xs.max() + 1])
    for s in range(4):
        d["out:disp_%d" % s] = outputs[("disp", s)]
    for f in (-1, 1):
        n = "m1" if f < 0 else "p1"
        d["out:cam_T_cam_" + n] = outputs[("cam_T_cam", 0, f)]
        d["out:axisangle_" + n] = outputs[("axisangle", 0, f)]
        d["out:translation_" + n] = outputs[("translation", 0, f)]
        d["out:mvs_color_" + n] = outputs[("mvs_color", f)]
        d["out:mvs_mask_" + n] = outputs[("mvs_mask", f)]
    d["out:color_m1_0"] = outputs[("color", -1, 0)]
    d["out:mvs_color_fuse_p1"] = outputs[("mvs_color_fuse", 1)]
    d["in:relative_pose_m1"] = inputs[("relative_pose", -1)]
    # gradients: per-parameter L2 norms for every sub-model; full tensors for the small ones
    for name in sorted(t.models):
        norms = []
        for pn, p in t.models[name].named_parameters():
            norms.append(0.0 if p.grad is None else float(p.grad.double().norm()))
        d["gradnorm:" + name] = np.array(norms, np.float64)
    for name, pn in FULL_GRADS:
        p = dict(t.models[name].named_parameters())[pn]
        d["grad:%s:%s" % (name, pn)] = p.grad
        d["noise:grad:%s:%s" % (name, pn)] = _rel(p.grad, dict(t64.models[name].named_parameters())[pn].grad)
    # the reference's own float32-vs-float64 distance for every sub-model's vector of per-parameter gradient norms
    for name in sorted(t.models):
        n32 = torch.tensor([0.0 if p.grad is None else float(p.grad.double().norm()) for _, p in t.models[name].named_parameters()])
        n64 = torch.tensor([0.0 if p.grad is None else float(p.grad.double().norm()) for _, p in t64.models[name].named_parameters()])
        d["noise:gradnorm:" + name] = _rel(n32, n64)
    # BatchNorm running statistics after the step (momentum update with the batch statistics)
    d["bn:mvs_encoder.conv0.0.bn.running_mean"] = t.models["mvs_encoder"].state_dict()["conv0.0.bn.running_mean"]
    d["bn:reg3d.conv0.bn.running_var"] = t.models["reg3d"].state_dict()["conv0.bn.running_var"]
    save("step_" + tag, d)


def gen_eval(L, networks, frames):
    """evaluate_depth.py:181-256, composed from the imported reference's own layers / networks ("inline": the script's
    body is module-level code with hard .cuda() calls, so its lines are executed here on the CPU with the same calls)."""
    _, models = build_weights()
    opt = parse_ref_options([])
    ref = ref_trainer(L, _TrainerShell, networks, opt, models)
    for m in ref.models.values():
        m.eval()
    fh, fw = H // 4, W // 4
    bp, pj = L.BackprojectDepth(D, fh, fw), L.Project3D(D, fh, fw)
    for tag, frames_to_load in (("n1", [0, -1]), ("n2", [0, -1, 1])):
        with torch.no_grad():
            data = {k: v.clone() for k, v in frames.items()}
            # evaluate_depth.py:186-200: poses of every lookup frame from the pose network (matching_ids order)
            pose_feats = {f: data[("color", f, 0)] for f in frames_to_load}
            for f in frames_to_load[1:]:
                pair = [pose_feats[f], pose_feats[0]] if f < 0 else [pose_feats[0], pose_feats[f]]
                aa, tr = ref.models["pose"]([ref.models["pose_encoder"](torch.cat(pair, 1))])
                data[("relative_pose", f)] = L.transformation_from_parameters(aa[:, 0], tr[:, 0], invert=(f < 0))
            relative_poses = torch.stack([data[("relative_pose", f)] for f in frames_to_load[1:]], 1)
            # :203-206 features
            ref_feat, ref_ctx = ref.models["mvs_encoder"](data[("color", 0, 0)])
            src_feats = [ref.models["mvs_encoder"](data[("color_aug", f, 0)])[0] for f in frames_to_load[1:]]   # :207
            # :208-215 mono prior
            out = ref.models["mono_depth"](ref.models["mono_encoder"](data[("color", 0, 0)]))
            disp_prior = out[("disp", opt.prior_scale)]
            depth_prior = 1 / (1 / opt.max_depth + disp_prior * (1 / opt.min_depth - 1 / opt.max_depth))
            # :216-222 velocity-guided range from batch element 0, lookup frame 0 (scalar z)
            z = opt.z_scale * relative_poses[0, 0, 2, -1]
            hyp = L.schedule_depth_range_zv2(depth_prior, ndepth=D, scale_fac=opt.depth_bin_fac, z_trans=z)
            # :224-240 cost volume with the evaluation-time confidence weight (softmax over D of the group mean)
            cor_weight_sum, cor_feats, ws = 1e-8, 0, []
            for fi in range(len(frames_to_load) - 1):
                cv = L.generate_costvol(ref_feat, src_feats[fi], data[("K", 2)], data[("inv_K", 2)], hyp,
                                        relative_poses[:, fi:fi + 1], D, bp, pj)
                Bq, Dq, Cq, Hq, Wq = cv.shape
                cv = cv.reshape(Bq, Dq, -1, opt.reg3d_c, Hq, Wq).mean(2)
                cw = torch.softmax(cv.mean(2), dim=1).max(1)[0]            # evaluate_depth.py:236
                ws.append(cw)
                cor_weight_sum = cor_weight_sum + cw
                cor_feats = cor_feats + cw.unsqueeze(1).unsqueeze(1) * cv
            cor_feats = cor_feats / cor_weight_sum.unsqueeze(1).unsqueeze(1)
            # :242-250 regularise, regress, upsample
            prob = F.softmax(ref.models["reg3d"](cor_feats), 1)
            depth = L.localmax(prob, opt.norm_radius, D, 1 / hyp[:, -1], 1 / hyp[:, 0])
            up = ref.models["up"](depth, ref_ctx)
        d = dict(frames_to_load=np.array(frames_to_load), z_trans=z, hyp=hyp, cor_feats=cor_feats, depth_lowres=depth,
                 pred_depth=up, disp_prior=disp_prior)
        for i, w_ in enumerate(ws):
            d["cor_weight%d" % i] = w_
            d["relative_pose%d" % i] = relative_poses[:, i]
        save("eval_" + tag, d)


class _TrainerShell:
    """bare attribute holder for gen_eval (the evaluation script does not use the Trainer class)"""


def gen_ckpt_manifest(networks, L):
    """state_dict key -> shape of every sub-model as the REFERENCE's classes define them (ResNet-18 and ResNet-50
    encoders), + the list of per-model file names save_model writes (trainer.py:807-831)."""
    man = {}
    for arch in (18, 50):
        opt = parse_ref_options(["--res_arch", str(arch)])
        m = {}
        m["mono_encoder"] = networks.ResnetEncoder(arch, False)
        m["mono_depth"] = networks.DepthDecoder(m["mono_encoder"].num_ch_enc, opt.scales, match_conv=False, ddv=False,
                                                discret=None, mono_conf=False, mono_bins=False)
        m["pose_encoder"] = networks.ResnetEncoder(arch, False, num_input_images=2)
        m["pose"] = networks.PoseDecoder(m["pose_encoder"].num_ch_enc, num_input_features=1, num_frames_to_predict_for=2)
        m["mask_cnn"] = networks.UncertNet()
        m["mvs_encoder"] = networks.FPN4(base_channels=8, scale=opt.prior_scale, dcn=False)
        m["reg3d"] = networks.reg3d(in_channels=opt.reg3d_c, base_channels=opt.reg3d_c, down_size=3)
        m["up"] = L.convex_upsample_layer(feature_dim=8 * 2 ** opt.prior_scale, scale=opt.prior_scale)
        # per sub-model: the state_dict entries IN ORDER, [key, dtype, dim0, dim1, ...]
        man["res%d" % arch] = {k: [[kk, str(v.dtype).replace("torch.", "")] + list(v.shape) for kk, v in mod.state_dict().items()]
                               for k, mod in m.items()}
    man["files"] = ["mono_encoder", "mono_depth", "pose_encoder", "pose", "mask_cnn", "mvs_encoder", "reg3d", "up", "adam"]
    path = os.path.join(OUT, "ckpt_manifest.json")
    json.dump(man, open(path, "w"), separators=(",", ":"))
    print("%-28s %8.1f KB" % ("ckpt_manifest.json", os.path.getsize(path) / 1024))


def gen_networks_forward(networks, L):
    """Every sub-model class of the reference (networks/*.py, layers.convex_upsample_layer) with formula weights
    (step_fixture.formula_state), forward in training mode on seeded inputs: the outputs this repo's own network
    definitions must reproduce from the same formula (tests/test_networks_cpu.py)."""
    g = torch.Generator().manual_seed(31)
    img = torch.rand(2, 3, 64, 128, generator=g)
    pair = torch.rand(2, 6, 64, 128, generator=g)
    d = {"img": img, "pair": pair}

    def load(m):
        m.load_state_dict(formula_state(m.state_dict()), strict=True)
        return m.train()

    def summarise(prefix, t):
        d[prefix] = t if t.numel() <= 20000 else t.flatten()[:: max(1, t.numel() // 4096)][:4096].clone()
        d[prefix + ":sums"] = torch.tensor([float(t.double().sum()), float(t.double().abs().sum())], dtype=torch.float64)

    for arch in (18, 50):
        enc = load(networks.ResnetEncoder(arch, False))
        feats = enc(img)
        for i, f in enumerate(feats):
            summarise("enc%d_f%d" % (arch, i), f)
        dec = load(networks.DepthDecoder(enc.num_ch_enc, [0, 1, 2, 3], match_conv=False, ddv=False, discret=None,
                                         mono_conf=False, mono_bins=False))
        out = dec(feats, no_match=False)
        for s_ in range(4):
            summarise("disp%d_s%d" % (arch, s_), out[("disp", s_)])
        penc = load(networks.ResnetEncoder(arch, False, num_input_images=2))
        pf = penc(pair)
        summarise("penc%d_f4" % arch, pf[-1])
        pdec = load(networks.PoseDecoder(penc.num_ch_enc, num_input_features=1, num_frames_to_predict_for=2))
        aa, tr = pdec([pf])
        summarise("pose%d_aa" % arch, aa)
        summarise("pose%d_tr" % arch, tr)
    fpn = load(networks.FPN4(base_channels=8, scale=2, dcn=False))
    mf, cf = fpn(img)
    summarise("fpn_match", mf)
    summarise("fpn_context", cf)
    unc = load(networks.UncertNet())
    ent = torch.rand(2, 1, 16, 32, generator=g)
    d["entropy"] = ent
    summarise("uncert", unc(ent))
    r3 = load(networks.reg3d(in_channels=16, base_channels=16, down_size=3))
    vol = torch.randn(1, 16, 16, 16, 32, generator=g) * 0.1   # B D G h w
    d["vol"] = vol
    summarise("reg3d", r3(vol))
    r2 = load(networks.reg2d(input_channel=16, base_channel=8))
    vol2 = torch.randn(1, 4, 16, 16, 32, generator=g) * 0.1
    d["vol2"] = vol2
    summarise("reg2d", r2(vol2))
    up = load(L.convex_upsample_layer(feature_dim=32, scale=2))
    dep = 2 + torch.rand(2, 16, 32, generator=g)
    d["up_depth"] = dep
    summarise("up", up(dep, cf))
    save("networks_forward", d)


def main():
    torch.set_num_threads(8)
    L, Trainer, networks = load_reference(with_trainer=True, working_resnet=True)
    torch.set_num_threads(8)
    frames = make_frames()
    fx = {}
    for f in (0, -1, 1):
        fx["color_%d" % f] = frames[("color", f, 0)]
        fx["color_aug_%d" % f] = frames[("color_aug", f, 0)]
    for s in range(4):
        fx["K_%d" % s], fx["inv_K_%d" % s] = frames[("K", s)], frames[("inv_K", s)]
    fx["dims"] = np.array([B, H, W, D])
    for tag, (epoch, extra) in CASES.items():
        run_case(tag, epoch, extra, L, Trainer, networks, frames)
    _, fresh = build_weights()
    for name, cs in checksums(fresh).items():
        fx["wsum:" + name] = cs
    save("step_inputs", fx)
    gen_eval(L, networks, frames)
    gen_ckpt_manifest(networks, L)
    gen_networks_forward(networks, L)


if __name__ == "__main__":
    main()
