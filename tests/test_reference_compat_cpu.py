"""Drop-in compatibility with the reference's networks and checkpoints, on the CPU:

* tests/golden/ckpt_manifest.json: state_dict key -> (dtype, shape) of every sub-model as the REFERENCE's own classes define
  them (ResNet-18 and ResNet-50 variants; written by tools/gen_golden_step.py from /root/reference/movedepth/networks).  This
  repo's `build_models` must produce exactly those keys, in that order: `{model}.pth` files then load either way with
  strict=True (the reference evaluator does: evaluate_depth.py:118-174).
* tests/golden/networks_forward.npz: the reference's classes, given formula weights (tools/step_fixture.formula_state), run
  forward in training mode; this repo's classes, given the same formula, must produce the same numbers.
* Trainer.save_model writes plain state_dicts (checked on the GPU in test_trainer_parity.py; the key sets are pinned here).
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, load_golden

sys.path.insert(0, os.path.join(ROOT, "tools"))
import step_fixture as fx  # noqa: E402

from movedepth_amd import networks  # noqa: E402
from movedepth_amd.options import MovedepthOptions  # noqa: E402
from movedepth_amd.trainer import build_models  # noqa: E402


@pytest.mark.parametrize("arch", [18, 50])
@pytest.mark.parametrize("fused_bn", [0, 1])
def test_state_dict_manifest_matches_reference(arch, fused_bn):
    man = json.load(open(os.path.join(GOLDEN, "ckpt_manifest.json")))
    opt = MovedepthOptions().parse(["--res_arch", str(arch), "--convex_up", "--weights_init", "scratch", "--hip_bn_relu",
                                    str(fused_bn)])
    models, main, mvs = build_models(opt)
    assert sorted(models) == sorted(n for n in man["files"] if n != "adam")
    assert set(main) | set(mvs) == set(models)
    for name, m in models.items():
        want = man["res%d" % arch][name]   # [[key, dtype, *shape], ...] in the reference's order
        have = [[k, str(v.dtype).replace("torch.", "")] + list(v.shape) for k, v in m.state_dict().items()]
        assert [e[0] for e in have] == [e[0] for e in want], "%s: key order / set differs from the reference's" % name
        assert have == want, name


def _check(g, key, t, rtol=1e-5):
    t = t.detach()
    sums = np.array([float(t.double().sum()), float(t.double().abs().sum())])
    np.testing.assert_allclose(sums, g[key + ":sums"], rtol=rtol, atol=1e-6, err_msg=key)
    want = g[key]
    got = t if t.numel() <= 20000 else t.flatten()[:: max(1, t.numel() // 4096)][:4096]
    np.testing.assert_allclose(got.numpy(), want, rtol=rtol, atol=1e-6, err_msg=key)


def _load(m):
    m.load_state_dict(fx.formula_state(m.state_dict()), strict=True)
    return m.train()


@pytest.mark.parametrize("arch", [18, 50])
def test_encoders_and_decoders_forward_match_reference(arch):
    g = load_golden("networks_forward")
    img, pair = torch.from_numpy(g["img"]), torch.from_numpy(g["pair"])
    enc = _load(networks.ResnetEncoder(arch, False))
    feats = enc(img)
    for i, f in enumerate(feats):
        _check(g, "enc%d_f%d" % (arch, i), f)
    dec = _load(networks.DepthDecoder(enc.num_ch_enc, [0, 1, 2, 3]))
    out = dec(feats, no_match=False)
    for s in range(4):
        _check(g, "disp%d_s%d" % (arch, s), out[("disp", s)])
    penc = _load(networks.ResnetEncoder(arch, False, num_input_images=2))
    pf = penc(pair)
    _check(g, "penc%d_f4" % arch, pf[-1])
    aa, tr = _load(networks.PoseDecoder(penc.num_ch_enc, num_input_features=1, num_frames_to_predict_for=2))([pf])
    _check(g, "pose%d_aa" % arch, aa)
    _check(g, "pose%d_tr" % arch, tr)


def test_mvs_networks_forward_match_reference():
    g = load_golden("networks_forward")
    img = torch.from_numpy(g["img"])
    mf, cf = _load(networks.FPN4(base_channels=8, scale=2))(img)
    _check(g, "fpn_match", mf)
    _check(g, "fpn_context", cf)
    _check(g, "uncert", _load(networks.UncertNet())(torch.from_numpy(g["entropy"])))
    for fused in (False, True):  # the fused-BatchNorm variant falls back to the same torch ops on the CPU
        _check(g, "reg3d", _load(networks.reg3d(16, 16, down_size=3, fused_bn=fused))(torch.from_numpy(g["vol"])), rtol=2e-5)
    _check(g, "reg2d", _load(networks.reg2d(16, 8))(torch.from_numpy(g["vol2"])), rtol=2e-5)
    up = _load(networks.convex_upsample_layer(feature_dim=32, scale=2))
    _check(g, "up", up(torch.from_numpy(g["up_depth"]), cf))
