"""KITTI-layout producer of the trainer's input dictionary, without torchvision / cv2 / skimage.

What the reference's `MonoDataset.__getitem__` + `KITTIRAWDataset` hand to `Trainer.process_batch`
(datasets/mono_dataset.py:134-237, datasets/kitti_dataset.py:19-90), reproduced from its contract:

  ("color", f, s), ("color_aug", f, s)   float32 in [0, 1], (3, H // 2**s, W // 2**s), f in frame_idxs, s in 0..num_scales-1;
                                          scale s is a LANCZOS resize of scale s-1 (s = 0: of the frame as stored on disk);
  ("K", s), ("inv_K", s)                  (4, 4) float32: KITTI's normalised intrinsics (kitti_dataset.py:26-29) with row 0 times
                                          W // 2**s and row 1 times H // 2**s (integer division, mono_dataset.py:212-213),
                                          inv_K = pinv(K);
  training only, each decided once per item with probability 1/2: horizontal flip of every frame; colour jitter (brightness,
  contrast, saturation in [0.8, 1.2], hue in [-0.1, 0.1], in a random order) on "color_aug" only -- with fresh factors for
  every image, as torchvision 0.8.2's ColorJitter.forward draws them (see ColorJitter below);
  an all-black frame keeps color_aug = color; a neighbouring frame missing on disk is replaced by the frame next to it.

File layout: <data_path>/<folder>/image_0{2,3}/data/<frame:010d><ext>, split lines "<folder> <frame> <l|r>".
Not produced: "depth_gt" (needs the velodyne projection of kitti_utils.py, out of scope: it is only read by the monitoring
metrics) and ('relative_pose', f) (DVSO poses, `--load_pose`).  Random draws: the reference's generators in the reference's
order (python `random`, numpy global, torch global); the logic half of the item -- key set, frame substitution at sequence
ends, intrinsics, the two coins and the position of all three random streams after an item -- is pinned to the reference's own
`MonoDataset.__getitem__` by tests/golden/loader.json (tools/gen_golden_loader.py); resized / jittered pixel values are not
(they are torchvision's in the reference, which this image does not have).
"""
import os
import random

import numpy as np
import torch
from PIL import Image, ImageEnhance

KITTI_K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
_SIDE = {"2": 2, "3": 3, "l": 2, "r": 3}


def read_split(path):
    with open(path) as f:
        return f.read().splitlines()


def _to_tensor(img):
    a = np.asarray(img, dtype=np.uint8)
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div_(255.0)


def _shift_hue(img, h):
    """hue shift by h in [-0.5, 0.5] turns of the colour wheel, as PIL-image colour jitter does it (8-bit HSV)."""
    hsv = np.array(img.convert("HSV"), dtype=np.uint8)
    hsv[..., 0] = (hsv[..., 0].astype(np.int16) + int(round(h * 255))) % 256
    return Image.fromarray(hsv, "HSV").convert("RGB")


class ColorJitter:
    """Colour jitter with the reference's random stream: `transforms.ColorJitter(...)` of torchvision 0.8.2 (the version the
    reference pins, environment.yml:16) draws in forward(), i.e. AT EVERY CALL -- torch.randperm(4) for the order, then one
    torch.tensor(1.0).uniform_(lo, hi) per operation as it comes up -- so although mono_dataset.py:104-109 says the same
    augmentation is applied to every image of an item, each image (every frame, every scale) gets its own factors; only
    the decision to augment at all is shared.  The same calls are made here, in the same order, on torch's global generator
    (which DataLoader seeds per worker and epoch): same stream.  The pixel arithmetic below is PIL's (torchvision's PIL path is
    PIL's ImageEnhance + an 8-bit HSV shift as well), not pinned by a fixture: torchvision is not in this image."""

    def __init__(self, brightness=(0.8, 1.2), contrast=(0.8, 1.2), saturation=(0.8, 1.2), hue=(-0.1, 0.1)):
        self.ranges = (brightness, contrast, saturation, hue)

    def __call__(self, img):
        for op in torch.randperm(4).tolist():
            lo, hi = self.ranges[op]
            factor = torch.tensor(1.0).uniform_(lo, hi).item()
            if op == 0:
                img = ImageEnhance.Brightness(img).enhance(factor)
            elif op == 1:
                img = ImageEnhance.Contrast(img).enhance(factor)
            elif op == 2:
                img = ImageEnhance.Color(img).enhance(factor)
            else:
                img = _shift_hue(img, factor)
        return img


class KITTIRAWDataset(torch.utils.data.Dataset):
    def __init__(self, data_path, filenames, height, width, frame_idxs, num_scales, is_train=False, img_ext=".jpg", seed=0):
        self.data_path, self.filenames = data_path, list(filenames)
        self.height, self.width = int(height), int(width)
        self.frame_idxs, self.num_scales = list(frame_idxs), int(num_scales)
        self.is_train, self.img_ext = bool(is_train), img_ext
        self.K = KITTI_K
        self.seed = int(seed)   # folded into the loader's generator (make_loader); the draws themselves come from the process-global generators, see __getitem__

    def __len__(self):
        return len(self.filenames)

    def image_path(self, folder, frame_index, side):
        return os.path.join(self.data_path, folder, "image_0%d/data" % _SIDE[side], "%010d%s" % (frame_index, self.img_ext))

    def _load(self, folder, frame_index, side, flip):
        with open(self.image_path(folder, frame_index, side), "rb") as f:
            img = Image.open(f).convert("RGB")
        return img.transpose(Image.FLIP_LEFT_RIGHT) if flip else img

    def __getitem__(self, index):
        parts = self.filenames[index].split()
        folder = parts[0]
        frame_index, side = (int(parts[1]), parts[2]) if len(parts) == 3 else (0, None)
        # the reference's draws, generator for generator and in its order (mono_dataset.py:160-166): python `random` for the two
        # coins (DataLoader re-seeds it in every worker, every epoch), then numpy's global generator for the robust-training
        # frame offsets, drawn whether or not they are used
        do_aug = self.is_train and random.random() > 0.5
        do_flip = self.is_train and random.random() > 0.5
        np.random.choice([-3, -2, -1, 1, 2, 3], 4, False)
        native = {}
        for i in self.frame_idxs:
            try:
                native[i] = self._load(folder, frame_index + i, side, do_flip)
            except FileNotFoundError:
                if i == 0:
                    raise FileNotFoundError("cannot find frame %s: check --data_path / --png" %
                                            self.image_path(folder, frame_index, side))
                native[i] = native[i - 1 if i > 0 else i + 1]  # sequence end: repeat the neighbour (mono_dataset.py:196-199)
        jitter = ColorJitter() if do_aug else (lambda im: im)
        # Resize every frame down the pyramid (scale s from scale s-1), then augment in the reference's ORDER of calls
        # (mono_dataset.py:111-125 walks the dictionary in insertion order): the native-resolution frames first -- their augmented
        # copies are deleted again at :226-228, but each call draws from torch's generator -- then frame by frame, scale by scale.
        # An all-black image is not augmented (and draws nothing).
        pyramid = {}
        for i in self.frame_idxs:
            img = native[i]
            for s in range(self.num_scales):
                img = img.resize((self.width // 2 ** s, self.height // 2 ** s), Image.LANCZOS)  # from the previous scale
                pyramid[(i, s)] = img
        if do_aug:
            for i in self.frame_idxs:
                if np.asarray(native[i]).any():
                    jitter(native[i])
        inputs = {}
        for i in self.frame_idxs:
            for s in range(self.num_scales):
                img = pyramid[(i, s)]
                t = _to_tensor(img)
                inputs[("color", i, s)] = t
                inputs[("color_aug", i, s)] = t if float(t.sum()) == 0 else _to_tensor(jitter(img))
        for s in range(self.num_scales):
            K = self.K.copy()
            K[0, :] *= self.width // (2 ** s)
            K[1, :] *= self.height // (2 ** s)
            inputs[("K", s)] = torch.from_numpy(K)
            inputs[("inv_K", s)] = torch.from_numpy(np.linalg.pinv(K))
        return inputs


def _seed_worker(_worker_id):
    """worker_init_fn (module level: picklable under the spawn / forkserver start methods): DataLoader has seeded torch's generator
    of this worker from the loader's base seed; carry it to python's and numpy's, which the augmentation draws from."""
    import random

    base = torch.initial_seed() % (2 ** 32)
    random.seed(base)
    np.random.seed(base)


def make_loader(dataset, batch_size, rank=0, world_size=1, shuffle=True, num_workers=0, seed=0, drop_last=True):
    """DataLoader over `dataset` with the reference's sharding (one DistributedSampler-style rank-strided shard per process,
    trainer.py:171-179; drop_last as there).  Returns (loader, sampler-or-None): call sampler.set_epoch(e) per epoch."""
    sampler = None
    if world_size > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world_size, rank=rank, shuffle=shuffle,
                                                                  seed=seed)
    # Augmentation draws come from the process-global generators, as in the reference (see MonoDataset.__getitem__), and
    # train.py seeds every rank alike: without more, all ranks (and worker i of every rank) would draw the same flip / jitter
    # stream.  The loader's own generator -- from which DataLoader derives each worker's base seed -- therefore folds in the
    # rank and the dataset's seed, and worker_init_fn carries that base seed to python's and numpy's generators (torch's is
    # seeded by DataLoader itself).  With num_workers = 0 the draws are the main process's, seeded alike on every rank by train.py exactly as upstream (train.py:8-19).
    # One process without workers is the reference's own situation (train.py:8-19 seeds, trainer.py:172-179 builds the loader
    # without a generator): the shuffle order then comes from the globally seeded torch generator, exactly as upstream -- no
    # private generator there.
    gen = None
    if world_size > 1 or num_workers > 0:
        gen = torch.Generator()
        gen.manual_seed(int(seed) + 7919 * int(getattr(dataset, "seed", 0)) + 1000003 * int(rank))
    loader = torch.utils.data.DataLoader(dataset, batch_size, shuffle=(shuffle and sampler is None), sampler=sampler,
                                         num_workers=num_workers, pin_memory=True, drop_last=drop_last, generator=gen,
                                         worker_init_fn=_seed_worker if num_workers > 0 else None)
    return loader, sampler
