#!/usr/bin/env python3
"""tests/golden/loader.json: the LOGIC half of the reference's data loader, recorded from the reference's own
`KITTIRAWDataset.__getitem__` (datasets/mono_dataset.py:134-237, datasets/kitti_dataset.py:19-90) imported in this container.

torchvision is not in this image, and the pixel half of the loader (LANCZOS resize through torchvision.transforms.Resize,
ColorJitter's pixel arithmetic) is torchvision's: a stand-in for it would pin nothing, so NO resized / jittered pixel value is
recorded.  What is recorded is everything that does not depend on that arithmetic, using a synthetic KITTI-layout tree whose
frames are constant-colour images (value = 10 * frame number + 5, blank for frame 4), so that any resize returns the same
constant and the colour identifies the file that was read:

  * the key set of an item, tensor shapes and dtypes;
  * which stored frame each ("color", f, s) came from: sequence start / end substitution (mono_dataset.py:196-205), side l / r;
  * K and inv_K per scale (mono_dataset.py:209-218);
  * the blank-frame rule (color_aug is color, mono_dataset.py:121-125);
  * in training mode, from fixed seeds: the two coins of every item (flip, colour augmentation), and the position of the three
    random streams (python `random`, numpy global, torch global) after each item -- the order and number of draws.

The stand-in modules (no reference file is edited or copied): torchvision.transforms with ToTensor = uint8 -> float / 255 CHW,
Resize = PIL resize, ColorJitter = a marker that performs torchvision 0.8.2's random CALLS (torch.randperm(4), four
torch.tensor(1.0).uniform_) and inverts the image, so "was augmented" is visible and the torch stream advances as it would.
"""
import json
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "loader.json")
H, W, FRAMES, NSCALES = 64, 96, [0, -1, 1], 4
NFRAMES, BLANK = 6, 4
LINES = ["seq/drive 2 l", "seq/drive 0 l", "seq/drive 5 l", "seq/drive 3 r", "seq/drive 4 l", "seq/drive 1 r"]


def write_tree(root, ext=".png"):
    """<root>/seq/drive/image_0{2,3}/data/%010d.png: constant colour 10*n+5 (+100 on the right camera), frame 4 blank."""
    for cam, off in ((2, 0), (3, 100)):
        d = os.path.join(root, "seq/drive", "image_0%d/data" % cam)
        os.makedirs(d, exist_ok=True)
        for n in range(NFRAMES):
            v = 0 if n == BLANK else 10 * n + 5 + off
            Image.fromarray(np.full((80, 120, 3), v, np.uint8)).save(os.path.join(d, "%010d%s" % (n, ext)))


def install_stubs():
    sys.dont_write_bytecode = True

    def stub(name, **kw):
        m = types.ModuleType(name)
        for k, v in kw.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    stub("cv2", setNumThreads=lambda n: None)
    stub("pykitti")
    sk = stub("skimage")
    sk.transform = stub("skimage.transform")
    if not hasattr(Image, "ANTIALIAS"):
        Image.ANTIALIAS = Image.LANCZOS        # removed in Pillow 10; mono_dataset.py:56 names it

    class ToTensor:
        def __call__(self, img):
            a = np.asarray(img, dtype=np.uint8)
            return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255.0)

    class Resize:
        def __init__(self, size, interpolation=None):
            self.size, self.interp = size, interpolation

        def __call__(self, img):
            return img.resize((self.size[1], self.size[0]), self.interp)

    class ColorJitter:
        def __init__(self, brightness, contrast, saturation, hue):
            self.ranges = (brightness, contrast, saturation, hue)

        @staticmethod
        def get_params(b, c, s, h):     # called once by MonoDataset.__init__ to probe the argument style
            return None

        def __call__(self, img):        # torchvision 0.8.2 ColorJitter.forward's random calls; the image is inverted as a marker
            for op in torch.randperm(4).tolist():
                torch.tensor(1.0).uniform_(*self.ranges[op])
            return Image.fromarray(255 - np.asarray(img, dtype=np.uint8))

    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms", ToTensor=ToTensor, Resize=Resize, ColorJitter=ColorJitter)
    tv.models = stub("torchvision.models")
    sys.path.insert(0, "/root/reference")


def rng_fingerprint():
    """SHA-1 of the full state of the three generators the loader draws from: equal fingerprints = same number of draws, in
    the same order, from the same seeds"""
    import hashlib
    return {"py_random": hashlib.sha1(repr(random.getstate()).encode()).hexdigest(),
            "np_random": hashlib.sha1(np.random.get_state()[1].tobytes() + bytes([np.random.get_state()[2] % 256, np.random.get_state()[2] // 256])).hexdigest(),
            "torch_random": hashlib.sha1(torch.get_rng_state().numpy().tobytes()).hexdigest()}


def describe(item, plain=None):
    """JSON-able description of an item: nothing that depends on resize / jitter arithmetic."""
    d = {"keys": sorted(repr(k) for k in item), "shapes": {}, "source": {}, "K": {}, "inv_K": {}}
    for k, v in item.items():
        d["shapes"][repr(k)] = [list(v.shape), str(v.dtype)]
    for f in FRAMES:
        for s in range(NSCALES):
            c = item[("color", f, s)]
            assert float((c - c.flatten()[0]).abs().max()) == 0.0       # constant image stays constant
            d["source"]["%d,%d" % (f, s)] = int(round(float(c.flatten()[0]) * 255))
    for s in range(NSCALES):
        d["K"][str(s)] = item[("K", s)].numpy().tolist()
        d["inv_K"][str(s)] = item[("inv_K", s)].numpy().tolist()
    d["aug_is_color"] = {str(f): bool(torch.equal(item[("color_aug", f, 0)], item[("color", f, 0)])) for f in FRAMES}
    return d


def main():
    install_stubs()
    from movedepth.datasets.kitti_dataset import KITTIRAWDataset

    out = {"H": H, "W": W, "frames": FRAMES, "num_scales": NSCALES, "lines": LINES, "nframes": NFRAMES, "blank": BLANK}
    with tempfile.TemporaryDirectory() as root:
        write_tree(root)
        ds = KITTIRAWDataset(root, LINES, H, W, FRAMES, NSCALES, is_train=False, img_ext=".png")
        out["eval_items"] = [describe(ds[i]) for i in range(len(LINES))]
        # flip detection needs a non-constant frame: a second tree with a left-to-right ramp, read at full information
        root2 = os.path.join(root, "ramp")
        for cam in (2, 3):
            d = os.path.join(root2, "seq/drive", "image_0%d/data" % cam)
            os.makedirs(d, exist_ok=True)
            for n in range(NFRAMES):
                a = np.zeros((80, 120, 3), np.uint8)
                a[:, :60] = 40 + n
                a[:, 60:] = 200 + n
                Image.fromarray(a).save(os.path.join(d, "%010d.png" % n))
        tr = KITTIRAWDataset(root2, ["seq/drive 2 l"] * 16, H, W, FRAMES, NSCALES, is_train=True, img_ext=".png")
        seed = 20240917
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        items = []
        for i in range(16):
            it = tr[i]
            c, a = it[("color", 0, 0)], it[("color_aug", 0, 0)]
            flipped = bool(c[0, 0, 0] > c[0, 0, -1])                    # bright half on the left
            augmented = not torch.equal(c, a)
            for f in FRAMES:                                            # one coin for every frame of the item
                cf, af = it[("color", f, 0)], it[("color_aug", f, 0)]
                assert bool(cf[0, 0, 0] > cf[0, 0, -1]) == flipped and (not torch.equal(cf, af)) == augmented
            items.append({"flipped": flipped, "augmented": augmented, **rng_fingerprint()})
        out["train_seed"], out["train_items"] = seed, items
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT, {"eval": len(out["eval_items"]), "train": len(items),
                         "coins": [(i["flipped"], i["augmented"]) for i in items]})


if __name__ == "__main__":
    main()
