"""movedepth_amd.datasets.KITTIRAWDataset against tests/golden/loader.json, recorded from the REFERENCE's own
MonoDataset.__getitem__ / KITTIRAWDataset (tools/gen_golden_loader.py; datasets/mono_dataset.py:134-237,
kitti_dataset.py:19-90): key set, shapes, which stored frame every ("color", f, s) comes from (sequence ends, side), K / inv_K per
scale, the blank-frame rule, and -- in training mode from the same seeds -- the two coins of every item and the state of all
three random generators after it.  Resized / jittered pixel VALUES are torchvision's in the reference and are not pinned."""
import hashlib
import json
import os
import random

import numpy as np
import torch
from PIL import Image

from conftest import GOLDEN
from movedepth_amd import datasets

G = json.load(open(os.path.join(GOLDEN, "loader.json")))
H, W, FRAMES, NS = G["H"], G["W"], G["frames"], G["num_scales"]


def _write_constant_tree(root):
    for cam, off in ((2, 0), (3, 100)):
        d = os.path.join(root, "seq/drive", "image_0%d/data" % cam)
        os.makedirs(d, exist_ok=True)
        for n in range(G["nframes"]):
            v = 0 if n == G["blank"] else 10 * n + 5 + off
            Image.fromarray(np.full((80, 120, 3), v, np.uint8)).save(os.path.join(d, "%010d.png" % n))


def _write_ramp_tree(root):
    for cam in (2, 3):
        d = os.path.join(root, "seq/drive", "image_0%d/data" % cam)
        os.makedirs(d, exist_ok=True)
        for n in range(G["nframes"]):
            a = np.zeros((80, 120, 3), np.uint8)
            a[:, :60] = 40 + n
            a[:, 60:] = 200 + n
            Image.fromarray(a).save(os.path.join(d, "%010d.png" % n))


def _fingerprint():
    return {"py_random": hashlib.sha1(repr(random.getstate()).encode()).hexdigest(),
            "np_random": hashlib.sha1(np.random.get_state()[1].tobytes() + bytes([np.random.get_state()[2] % 256, np.random.get_state()[2] // 256])).hexdigest(),
            "torch_random": hashlib.sha1(torch.get_rng_state().numpy().tobytes()).hexdigest()}


def test_item_structure_matches_the_reference(tmp_path):
    _write_constant_tree(str(tmp_path))
    ds = datasets.KITTIRAWDataset(str(tmp_path), G["lines"], H, W, FRAMES, NS, is_train=False, img_ext=".png")
    for i, want in enumerate(G["eval_items"]):
        it = ds[i]
        assert sorted(repr(k) for k in it) == want["keys"], G["lines"][i]
        for k, v in it.items():
            assert [list(v.shape), str(v.dtype)] == want["shapes"][repr(k)], (G["lines"][i], k)
        for f in FRAMES:
            for s in range(NS):
                c = it[("color", f, s)]
                assert float((c - c.flatten()[0]).abs().max()) == 0.0
                # the stored frame this tensor was read from (colour = 10 * frame + 5, + 100 on the right camera, 0 = blank)
                assert int(round(float(c.flatten()[0]) * 255)) == want["source"]["%d,%d" % (f, s)], (G["lines"][i], f, s)
            assert bool(torch.equal(it[("color_aug", f, 0)], it[("color", f, 0)])) == want["aug_is_color"][str(f)]
        for s in range(NS):
            assert np.array_equal(it[("K", s)].numpy(), np.array(want["K"][str(s)], np.float32)), (i, s)
            assert np.array_equal(it[("inv_K", s)].numpy(), np.array(want["inv_K"][str(s)], np.float32)), (i, s)


def test_training_coins_and_random_streams_match_the_reference(tmp_path):
    """Same seeds -> the same flip / augmentation decision for every item, and python `random`, numpy's and torch's global
    generators in the same state after every item as after the reference's __getitem__ (draw order and count:
    random.random() x 2, np.random.choice(6, 4, False), then per augmented image torch.randperm(4) + 4 uniform_ draws)."""
    _write_ramp_tree(str(tmp_path))
    ds = datasets.KITTIRAWDataset(str(tmp_path), ["seq/drive 2 l"] * len(G["train_items"]), H, W, FRAMES, NS, is_train=True, img_ext=".png")
    seed = G["train_seed"]
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    for i, want in enumerate(G["train_items"]):
        it = ds[i]
        c, a = it[("color", 0, 0)], it[("color_aug", 0, 0)]
        assert bool(c[0, 0, 0] > c[0, 0, -1]) == want["flipped"], i
        assert (not torch.equal(c, a)) == want["augmented"], i
        for f in FRAMES:
            cf, af = it[("color", f, 0)], it[("color_aug", f, 0)]
            assert bool(cf[0, 0, 0] > cf[0, 0, -1]) == want["flipped"] and (not torch.equal(cf, af)) == want["augmented"]
        got = _fingerprint()
        for k in ("py_random", "np_random", "torch_random"):
            assert got[k] == want[k], (i, k)
