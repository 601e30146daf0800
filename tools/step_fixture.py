"""Shared by tools/gen_golden_step.py (reference side, build container only) and tests/test_step_golden.py (GPU side):
the seeded sub-model weights and input frames of the whole-step fixtures.  Nothing here touches /root/reference.

The 113 MB of sub-model weights are not committed: both sides build this repo's `build_models` on the CPU under the same
seed (+ a seeded perturbation, below) and the fixture carries per-tensor checksums (step_inputs.npz `wsum:*`)."""
import numpy as np
import torch

B, H, W, D = 2, 64, 128, 16          # BASELINE config 1 image / bin count, two samples
WEIGHT_SEED, INPUT_SEED, STEP_SEED = 4242, 7, 99
BASE_ARGS = ["--height", str(H), "--width", str(W), "--num_depth_bins", str(D), "--batch_size", str(B), "--convex_up",
             "--weights_init", "scratch"]
MASK_FLAGS = ["--mask_mvs_conf", "--mask_mvs_dist", "--mask_mvs_auto", "--mvs_smooth_loss", "--photo_conf", "0.085",
              "--dist_thres", "0.468"]
# thresholds chosen so that both masks are non-trivial with these (untrained) weights: the probability volume is close to
# uniform (1/16 = 0.0625) and the disparities sit around 0.47
CASES = {"ep0_default": (0, []), "ep9_default": (9, []), "ep0_masks": (0, MASK_FLAGS), "ep9_masks": (9, MASK_FLAGS)}


def build_weights(extra_args=()):
    """This repo's sub-models under WEIGHT_SEED, on the CPU -> (options, {name: module})."""
    from movedepth_amd.options import MovedepthOptions
    from movedepth_amd.trainer import build_models

    opt = MovedepthOptions().parse(BASE_ARGS + list(extra_args) + ["--hip_bn_relu", "0"])
    torch.manual_seed(WEIGHT_SEED)
    models, _, _ = build_models(opt)
    # Fresh BatchNorm affine parameters are exactly 1 / 0 and the pose head starts near zero: perturb the 1-D parameters
    # (seeded) so that no gradient path is degenerate, and give the pose head a bias that is a real camera motion.
    g = torch.Generator().manual_seed(WEIGHT_SEED + 1)
    with torch.no_grad():
        for name in sorted(models):
            for pn, p in sorted(models[name].named_parameters()):
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
        # the decoder multiplies by 0.01: rotations of ~0.5 degrees, translations of a few centimetres of baseline units
        models["pose"].net[3].bias.copy_(torch.tensor([0.5, -0.8, 0.3, 4.0, -1.0, 6.0, -0.4, 0.7, -0.2, -3.0, 1.0, -5.0]))
    return opt, models


def make_frames():
    """Seeded frames + a mild photometric 'augmentation' so that color_aug != color (the two are used in different places)."""
    from movedepth_amd.synthetic import make_inputs

    inputs = make_inputs(B, H, W, (0, -1, 1), seed=INPUT_SEED, device="cpu")
    for f in (0, -1, 1):
        for s in range(4):
            inputs[("color_aug", f, s)] = (inputs[("color", f, s)] * 0.9 + 0.05).contiguous()
    return inputs


def checksums(models):
    """{name: (n_tensors, 2) float64 [sum, abs-sum] per floating-point state_dict entry, sorted by key}"""
    out = {}
    for name in sorted(models):
        sd = models[name].state_dict()
        out[name] = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for k, v in sorted(sd.items())
                              if v.dtype.is_floating_point], np.float64)
    return out


def formula_state(sd):
    """Deterministic weights for ANY module from its state_dict key order alone (so that the reference's classes and this
    repo's, built independently, can be given identical numbers without shipping them): entry i gets
    scale * sin(0.618 * arange(n) + i), BatchNorm scale / variance entries 1 + 0.1 * (that), integer buffers untouched."""
    out = {}
    for i, (k, v) in enumerate(sd.items()):
        if not v.dtype.is_floating_point:
            out[k] = v.clone()
            continue
        n = v.numel()
        base = torch.sin(0.618 * torch.arange(n, dtype=torch.float64) + i).reshape(v.shape)
        if k.endswith("running_var") or (v.dim() == 1 and k.endswith(".weight")):
            t = 1.0 + 0.1 * base
        else:
            fan = max(1, n // max(1, v.shape[0])) if v.dim() > 1 else 1
            t = base * (1.5 / fan ** 0.5 if v.dim() > 1 else 0.1)
        out[k] = t.to(v.dtype)
    return out
