"""Synthetic producer of the trainer's input-dict contract (reference
datasets/mono_dataset.py:134-237 and kitti_dataset.py:26-29): no KITTI in this environment.

keys: ("color", f, s), ("color_aug", f, s) float32 [0,1] (B,3,H/2^s,W/2^s) for f in frame_ids, s in 0..3;
      ("K", s), ("inv_K", s) (B,4,4): normalised KITTI intrinsics scaled per level, inv_K = pinv(K).
"""
import numpy as np
import torch
import torch.nn.functional as F

KITTI_K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)


def intrinsics(height, width, scale, batch):
    K = KITTI_K.copy()
    K[0, :] *= width // (2 ** scale)
    K[1, :] *= height // (2 ** scale)
    inv_K = np.linalg.pinv(K)
    return (torch.from_numpy(K).unsqueeze(0).repeat(batch, 1, 1),
            torch.from_numpy(inv_K.astype(np.float32)).unsqueeze(0).repeat(batch, 1, 1))


def make_inputs(batch, height, width, frame_ids=(0, -1, 1), num_scales=4, seed=0, device="cpu", smooth=8):
    """Seeded frames: low-pass noise (so bilinear taps are well conditioned); neighbouring frames are the
    reference frame shifted by a few pixels plus noise, so the photometric loss has something to fit."""
    g = torch.Generator().manual_seed(seed)
    coarse = torch.rand(batch, 3, max(2, height // smooth), max(2, width // smooth), generator=g)
    base = F.interpolate(coarse, size=(height, width), mode="bilinear", align_corners=True)
    inputs = {}
    for f in frame_ids:
        img = torch.roll(base, shifts=int(3 * f), dims=3) + 0.02 * torch.rand(batch, 3, height, width, generator=g)
        img = img.clamp(0, 1)
        for s in range(num_scales):
            h, w = height // 2 ** s, width // 2 ** s
            im = img if s == 0 else F.interpolate(img, size=(h, w), mode="bilinear", align_corners=False)
            inputs[("color", f, s)] = im.contiguous()
            inputs[("color_aug", f, s)] = im.contiguous()
    for s in range(num_scales):
        inputs[("K", s)], inputs[("inv_K", s)] = intrinsics(height, width, s, batch)
    return {k: v.to(device) for k, v in inputs.items()}


class SyntheticLoader:
    """Iterable standing in for the DataLoader: `steps` batches, rank-strided seeds like DistributedSampler
    (reference trainer.py:171; utils.py:73-87) so that ranks see different samples."""

    def __init__(self, batch, height, width, frame_ids, steps, rank=0, world_size=1, device="cpu"):
        self.args = (batch, height, width, tuple(frame_ids))
        self.steps, self.rank, self.world_size, self.device = steps, rank, world_size, device
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.steps

    def __iter__(self):
        b, h, w, fids = self.args
        for i in range(self.steps):
            seed = (self.epoch * self.steps + i) * self.world_size + self.rank
            yield make_inputs(b, h, w, fids, seed=seed, device=self.device)
