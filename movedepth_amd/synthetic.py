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


def _rodrigues(aa):
    """(3,) axis-angle -> 3x3 rotation (numpy, float64)"""
    th = float(np.linalg.norm(aa))
    if th < 1e-12:
        return np.eye(3)
    k = aa / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def driving_scene(batch, h, w, speed=1.0, seed=11):
    """Plane-sweep inputs with the parallax of KITTI odometry, which the reference's own synthetic case (prior U[2,22) m,
    translation 3-5 cm) does not have.  Returns (prior (B,1,h,w), pose (B,4,4)) as float32 numpy arrays at
    FEATURE resolution h x w:
      * depth in metres: a ground plane under the horizon (camera 1.65 m above the road, fy = 1.92 h as in kitti_dataset.py:26-29:
        6 m at the bottom row, 80 m at the horizon) and smooth 'facades' of 8-80 m above it;
      * source-from-reference transforms of a car driving straight: +-`speed` m along the optical axis (previous / next frame
        alternate over the batch), yaw within +-0.01 rad, pitch / roll within +-0.002, a few centimetres sideways and up.
    With speed = 1 (10 Hz at 36 km/h), t_z / depth spans 0.0125-0.2.  Used by tools/bench_costvol.py, bench.py
    (`roofline.parallax_cases`) and the oracle parity test at this launch shape."""
    rng = np.random.default_rng(seed)
    ys = np.arange(h, dtype=np.float64).reshape(1, 1, h, 1)
    ground = np.clip(1.65 * 1.92 * h / np.maximum(ys - 0.45 * h, 1e-3), 5.0, 80.0) * np.ones((batch, 1, h, w))
    cw = max(2, w // 16)
    coarse = rng.random((batch, 3, cw))
    # bilinear up-sampling of the coarse grid (align_corners=True), separable
    yi = np.linspace(0, 2, h)
    xi = np.linspace(0, cw - 1, w)
    y0 = np.clip(np.floor(yi).astype(int), 0, 1)
    x0 = np.clip(np.floor(xi).astype(int), 0, cw - 2)
    fy, fx = (yi - y0).reshape(1, h, 1), (xi - x0).reshape(1, 1, w)
    c = coarse
    top = c[:, y0][:, :, x0] * (1 - fx) + c[:, y0][:, :, x0 + 1] * fx
    bot = c[:, y0 + 1][:, :, x0] * (1 - fx) + c[:, y0 + 1][:, :, x0 + 1] * fx
    facade = 8.0 + 72.0 * (top * (1 - fy) + bot * fy).reshape(batch, 1, h, w)
    prior = np.minimum(ground, facade).astype(np.float32)
    pose = np.tile(np.eye(4), (batch, 1, 1))
    for b in range(batch):
        aa = np.array([(rng.random() * 2 - 1) * 0.002, (rng.random() * 2 - 1) * 0.01, (rng.random() * 2 - 1) * 0.002])
        pose[b, :3, :3] = _rodrigues(aa)
        pose[b, :2, 3] = (rng.random(2) * 2 - 1) * 0.05
        pose[b, 2, 3] = speed * (1.0 if b % 2 == 0 else -1.0)
    return prior, pose.astype(np.float32)
