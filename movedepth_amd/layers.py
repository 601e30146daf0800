"""Operator-level mirror of the reference's movedepth/layers.py for the hot path (same names, arguments and
return shapes), re-hosted on the HIP kernels of libmovedepth_hip.so through movedepth_amd.ops.

What differs by design, not by accident:
  * the geometry modules (BackprojectDepth / Project3D) are fused into the kernels that consume them
    (cost volume, photometric warp); the standalone modules below exist for call compatibility and run small
    HIP kernels of their own when called without autograd (as generate_costvol does upstream, layers.py:784);
  * generate_costvol returns the reference's (B,D,C,h,w) volume (G == C in the fused kernel); the trainer uses
    generate_costvol_grouped, which never materialises the C axis (SURVEY hard part 2).
The pose tail (Rodrigues 4x4 from 6 numbers) is one HIP kernel each way on the GPU; its torch-op form is kept
for CPU host use.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


# --------------------------------------------------------------------------- small geometry helpers
def disp_to_depth(disp, min_depth, max_depth):
    """reference layers.py:400-409"""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    depth = 1 / scaled_disp
    return scaled_disp, depth


def rot_from_axisangle(vec):
    """Axis-angle (B,1,3) -> 4x4 rotation by Rodrigues' formula R = cos(a) I + sin(a) [k]x + (1 - cos(a)) k k^T with
    k = v / (|v| + 1e-7) (the reference's guard, layers.py:485-486): same matrix as layers.py:479-518, written from the
    formula rather than entry by entry."""
    v = vec.reshape(-1, 3)
    angle = v.norm(dim=1, keepdim=True)
    k = v / (angle + 1e-7)
    c, s_ = torch.cos(angle)[:, :, None], torch.sin(angle)[:, :, None]
    zero = torch.zeros_like(k[:, 0])
    skew = torch.stack([zero, -k[:, 2], k[:, 1], k[:, 2], zero, -k[:, 0], -k[:, 1], k[:, 0], zero], 1).reshape(-1, 3, 3)
    outer = k[:, :, None] * k[:, None, :]
    # (1 - c) k k^T + c I: the reference's diagonal is x*x*(1-c) + c, i.e. NOT normalised by |k|^2 -- keep the identity term plain
    R3 = (1 - c) * outer + c * torch.eye(3, device=v.device, dtype=v.dtype) + s_ * skew
    R = torch.zeros(v.shape[0], 4, 4, device=v.device, dtype=v.dtype)
    R[:, :3, :3] = R3
    R[:, 3, 3] = 1
    return R


def get_translation_matrix(translation_vector):
    """reference layers.py:464-477"""
    t = translation_vector.contiguous().view(-1, 3, 1)
    T = torch.eye(4, device=t.device, dtype=t.dtype).repeat(t.shape[0], 1, 1)
    return torch.cat([T[:, :, :3], torch.cat([t, T[:, 3:, 3:]], 1)], 2)


def transformation_from_parameters(axisangle, translation, invert=False):
    """(axisangle, translation) -> 4x4; invert => R^T @ T(-t) else T(t) @ R.  reference layers.py:412-429.
    On the GPU one kernel forward, one backward (ops.pose_matrix); the torch-op form below is the CPU host path the
    tests use and costs ~75 launches each way."""
    if axisangle.is_cuda:
        return ops.pose_matrix(axisangle, translation, invert)
    R = rot_from_axisangle(axisangle)
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t = t * -1
    T = get_translation_matrix(t)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


class BackprojectDepth(nn.Module):
    """Depth image -> homogeneous camera points (Bs,4,h*w).  reference layers.py:556-586.
    Rigid in batch size like the reference (depth.view(batch_size, 1, -1))."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width
        ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float32), torch.arange(width, dtype=torch.float32),
                                indexing="ij")
        pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(height * width)], 0)
        self.register_buffer("pix_coords", pix.unsqueeze(0).repeat(batch_size, 1, 1), persistent=False)
        self.register_buffer("ones", torch.ones(batch_size, 1, height * width), persistent=False)

    def forward(self, depth, inv_K):
        if depth.is_cuda and not (torch.is_grad_enabled() and depth.requires_grad):
            return ops.backproject(depth, inv_K, self.batch_size, self.height, self.width)
        # differentiable use outside the fused kernels: plain torch ops, same arithmetic
        cam_points = torch.matmul(inv_K[:, :3, :3], self.pix_coords)
        cam_points = depth.view(self.batch_size, 1, -1) * cam_points
        return torch.cat([cam_points, self.ones], 1)


class Project3D(nn.Module):
    """Camera points -> normalised sampling grid (Bs,h,w,2) for align_corners=True.  reference layers.py:589-621."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points, K, T):
        if points.is_cuda and not (torch.is_grad_enabled() and (points.requires_grad or T.requires_grad)):
            return ops.project3d(points, K, T, self.batch_size, self.height, self.width, self.eps)
        P = torch.matmul(K, T)[:, :3, :]
        cam_points = torch.matmul(P, points)
        pix = cam_points[:, :2, :] / (cam_points[:, 2, :].unsqueeze(1) + self.eps)
        pix = pix.view(self.batch_size, 2, self.height, self.width).permute(0, 2, 3, 1)
        scale = torch.tensor([self.width - 1, self.height - 1], device=pix.device, dtype=pix.dtype)
        return (pix / scale - 0.5) * 2


# --------------------------------------------------------------------------- depth-range sampling
def schedule_depth_rangev2(prior_depth, ndepth, scale_fac, type="inverse"):
    """Depth hypotheses around the mono prior, (B,1,h,w) -> (B,D,h,w).  reference layers.py:256-284"""
    return ops.schedule_depth_range(prior_depth, ndepth, scale_fac, None, type)


def schedule_depth_range_zv2(prior_depth, ndepth, scale_fac, z_trans, type="inverse"):
    """Velocity-guided variant: range scaled by z_trans = z_scale * T[2,3] (B,1,1,1).  reference layers.py:370-398"""
    return ops.schedule_depth_range(prior_depth, ndepth, scale_fac, z_trans, type)


# --------------------------------------------------------------------------- plane-sweep cost volume
def _check_modules(backprojector, projector, D, h, w):
    for m in (backprojector, projector):
        if m is not None and (m.batch_size, m.height, m.width) != (D, h, w):
            raise RuntimeError("backprojector/projector built for %s, called with (D,h,w)=%s" %
                               ((m.batch_size, m.height, m.width), (D, h, w)))


def generate_costvol(ref, src, K, invK, depth_priors, pose, num_depth_bins, backprojector=None, projector=None):
    """reference layers.py:778-794 -> (B,D,C,h,w).  pose: (B,1,4,4).  The geometry modules are accepted for
    signature compatibility; the warp is computed inside the kernel."""
    B, C, h, w = ref.shape
    _check_modules(backprojector, projector, num_depth_bins, h, w)
    return ops.costvol_grouped(ref, src, K, invK, pose[:, 0], C, depth_priors=depth_priors, layout="bdg")


def generate_costvol_grouped(ref, src, K, invK, depth_priors, pose, num_depth_bins, groups, backprojector=None,
                             projector=None, layout="bgd"):
    """generate_costvol(...).reshape(B,D,-1,G,h,w).mean(2) (reference trainer.py:352-359) in one kernel.
    Logical shape (B,D,G,h,w); storage (B,G,D,h,w) with layout='bgd' (what reg3d permutes to)."""
    B, C, h, w = ref.shape
    _check_modules(backprojector, projector, num_depth_bins, h, w)
    return ops.costvol_grouped(ref, src, K, invK, pose[:, 0], groups, depth_priors=depth_priors, layout=layout)


# --------------------------------------------------------------------------- losses
class SSIM(nn.Module):
    """reference layers.py:646-677: reflect-pad 1, 3x3 box means, clamp((1 - n/d)/2, 0, 1)."""

    def forward(self, x, y):
        if torch.is_grad_enabled() and (x.requires_grad or y.requires_grad):
            raise RuntimeError("SSIM.forward is forward-only here; use ops.reprojection_loss for the differentiable "
                               "SSIM+L1 loss (gradient w.r.t. the prediction)")
        return ops.ssim_map(x, y)


def get_smooth_loss(disp, img):
    """Edge-aware smoothness (reference layers.py:630-643); the caller normalises disp as in the reference."""
    return ops.smooth_loss(disp, img, normalize=False)


# --------------------------------------------------------------------------- post-volume ops
def entropy(volume, dim, keepdim=False):
    """reference layers.py:862-863 (torch ops; the trainer uses the fused ops.softmax_entropy_localmax)."""
    return torch.sum(-volume * volume.clamp(1e-9, 1.).log(), dim=dim, keepdim=keepdim)


def localmax(cost_prob, radius, casbin, min_depth_inverse, max_depth_inverse):
    """reference layers.py:796-812 (torch ops; the trainer uses the fused ops.softmax_entropy_localmax)."""
    pred_idx = torch.argmax(cost_prob, 1, keepdim=True).float()
    offs = torch.arange(-radius, radius + 1, device=cost_prob.device).reshape(1, -1, 1, 1).float()
    idx = torch.clamp(pred_idx + offs, 0, casbin - 1).long()
    p = torch.gather(cost_prob, 1, idx)
    regress = (idx * p).sum(1, keepdim=True) / (1e-6 + p.sum(1, keepdim=True))
    norm = regress / (casbin - 1)
    return 1 / (min_depth_inverse + norm[:, 0] * (max_depth_inverse - min_depth_inverse))


def convex_upsample(depth, mask, scale=2):
    """RAFT-style convex upsampling (reference layers.py:200-214): depth (B,h,w) -> (B, 2**scale*h, 2**scale*w)."""
    if depth.is_cuda:
        return ops.convex_upsample(depth, mask, scale)
    if depth.dim() == 3:
        depth = depth.unsqueeze(1)
    B, _, H, W = depth.shape
    s = 2 ** scale
    mask = torch.softmax(mask.view(B, 9, s, s, H, W), dim=1)
    up = F.unfold(depth, [3, 3], padding=1).view(B, 9, 1, 1, H, W)
    up = torch.sum(mask * up, dim=1).permute(0, 3, 1, 4, 2)
    return up.reshape(B, s * H, s * W)


def random_image_mask(img, filter_size):
    """Erase one random rectangle, shared over the batch (reference layers.py:52-69; host RNG as upstream)."""
    fh, fw = filter_size
    _, _, h, w = img.size()
    if fh == h and fw == w:
        return img, None
    x = np.random.randint(0, w - fw)
    y = np.random.randint(0, h - fh)
    filter_mask = torch.ones_like(img)
    filter_mask[:, :, y:y + fh, x:x + fw] = 0.0
    return img * filter_mask, filter_mask
