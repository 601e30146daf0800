"""Inference path of the reference's evaluate_depth.py on the same HIP kernels, forward only.

`predict_depth` restates the inline forward of evaluate_depth.py:181-256 (mono prior -> velocity-guided hypotheses
-> plane-sweep volume -> reg3d -> softmax + localmax -> convex upsample); `compute_errors` is :22-40.  Reference
quirks kept: the hypothesis range uses the z-translation of batch element 0 for the whole batch (:218); with more
than one lookup frame the confidence weight is softmax over D of the mean over G (:236), not the training variant
(App. B-7).  The KITTI reader / ground-truth files are out of scope here; `evaluate` takes any iterable of input
dicts (movedepth_amd.synthetic.SyntheticLoader by default) and an optional ground-truth callback.
"""
import numpy as np
import torch

from . import ops
from .layers import disp_to_depth, transformation_from_parameters


def compute_errors(gt, pred):
    """abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3 (reference evaluate_depth.py:22-40)"""
    thresh = np.maximum((gt / pred), (pred / gt))
    a1, a2, a3 = (thresh < 1.25).mean(), (thresh < 1.25 ** 2).mean(), (thresh < 1.25 ** 3).mean()
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


@torch.no_grad()
def predict_depth(models, data, opt, vol_layout="ndhwc", details=False):
    """-> dict(depth_mvs (B,H,W), disp_mono (B,H,W), relative_poses (B,N,4,4)).  `models`: the trainer's dict.
    details=True adds the intermediates the parity tests compare: hyp (B,D,h,w), cor_feats (B,D,G,h,w), cor_weights
    (one (B,h,w) per lookup frame; empty for a single frame), depth_lowres (B,h,w), disp_prior."""
    dev = next(models["mono_encoder"].parameters()).device
    data = {k: v.to(dev) for k, v in data.items()}
    color = data[("color", 0, 0)]
    output = models["mono_depth"](models["mono_encoder"](color))
    frames = list(opt.matching_ids)   # frames_to_load = opt.matching_ids upstream (evaluate_depth.py:92)
    for fi in frames[1:]:
        pair = [data["color", fi, 0], data["color", 0, 0]] if fi < 0 else [data["color", 0, 0], data["color", fi, 0]]
        axisangle, translation = models["pose"]([models["pose_encoder"](torch.cat(pair, 1))])
        data[("relative_pose", fi)] = transformation_from_parameters(axisangle[:, 0], translation[:, 0], invert=fi < 0)
    relative_poses = torch.stack([data[("relative_pose", idx)] for idx in opt.matching_ids[1:]], 1)
    ref_feat, ref_context = models["mvs_encoder"](color)
    src_feats = [models["mvs_encoder"](data["color_aug", f_i, 0])[0] for f_i in opt.matching_ids[1:]]

    disp_prior = output[("disp", opt.prior_scale)]
    depth_prior = 1 / (1 / opt.max_depth + disp_prior * (1 / opt.min_depth - 1 / opt.max_depth))
    B = color.shape[0]
    z_trans = (opt.z_scale * relative_poses[0, 0, 2, -1]).reshape(1).repeat(B).contiguous()  # sample 0's z for all
    hyp = ops.schedule_depth_range(depth_prior, opt.num_depth_bins, opt.depth_bin_fac, z_trans, "inverse")
    vols = [ops.costvol_grouped(ref_feat, src_feats[f], data[("K", 2)], data[("inv_K", 2)], relative_poses[:, f],
                                opt.reg3d_c, prior=depth_prior, ndepth=opt.num_depth_bins, scale_fac=opt.depth_bin_fac,
                                z_trans=z_trans, type="inverse", layout=vol_layout) for f in range(len(src_feats))]
    weights = []
    # one lookup frame: w/(1e-8 + w) is the identity to 1.6e-7; more: the evaluation-time confidence weights (soft-max over D of
    # the group mean, evaluate_depth.py:236) and the weighted sum in one kernel
    cor, wts = ops.fuse_volumes_eval(vols, layout=vol_layout)
    weights = [] if wts is None else [wts[i] for i in range(len(vols))]
    logits = models["reg3d"](cor)
    depth_low, _, _ = ops.softmax_entropy_localmax(logits, 1 / hyp[:, -1], 1 / hyp[:, 0], opt.norm_radius)
    depth_mvs = models["up"](depth_low, ref_context) if opt.convex_up else depth_low
    disp_mono, _ = disp_to_depth(output[("disp", 0)], opt.min_depth, opt.max_depth)
    out = {"depth_mvs": depth_mvs, "disp_mono": disp_mono[:, 0], "relative_poses": relative_poses}
    if details:
        out.update(hyp=hyp, cor_feats=cor, cor_weights=weights, depth_lowres=depth_low, disp_prior=disp_prior)
    return out


def evaluate(trainer, loader, gt_fn=None, min_depth=1e-3, max_depth=80.0, median_scaling=True):
    """Runs predict_depth over `loader`; with gt_fn(batch_index, inputs) -> (B,H,W) ground-truth depth (numpy)
    returns the mean of the 7 metrics after the reference's median scaling / clamping (:262-331), else None."""
    trainer.set_eval()
    errors = []
    preds = []
    for i, data in enumerate(loader):
        out = predict_depth(trainer.models, data, trainer.opt, getattr(trainer, "vol_layout", "ndhwc"))
        pred = out["depth_mvs"].cpu().numpy()
        preds.append(pred)
        if gt_fn is None:
            continue
        gt = gt_fn(i, data)
        for b in range(pred.shape[0]):
            mask = np.logical_and(gt[b] > min_depth, gt[b] < max_depth)
            p, g = pred[b][mask], gt[b][mask]
            if median_scaling:
                p = p * np.median(g) / np.median(p)
            errors.append(compute_errors(g, np.clip(p, min_depth, max_depth)))
    trainer.set_train()
    return (np.array(errors).mean(0) if errors else None), np.concatenate(preds)
