"""Host logic of the network definitions on CPU (no GPU, no HIP library): every module class the trainer can
instantiate runs a tiny forward + backward, so an edit that breaks one of the less-used paths (reg2d is only built
for num_depth_bins < 8) is caught here rather than on the GPU box.  On CPU all convolutions are torch's."""
import pytest
import torch

from movedepth_amd import networks


def _runs(net, *inputs):
    out = net(*inputs)
    outs = list(out.values()) if isinstance(out, dict) else list(out) if isinstance(out, (tuple, list)) else [out]
    sum(o.float().sum() for o in outs if torch.is_tensor(o)).backward()
    assert all(p.grad is not None for p in net.parameters() if p.requires_grad)
    return outs


def test_reg3d_cpu_forward_backward():
    net = networks.reg3d(16, 16, 3)
    (out,) = _runs(net, torch.randn(1, 8, 16, 16, 16))  # (B,D,G,h,w)
    assert out.shape == (1, 8, 16, 16)


def test_reg3d_channels_last_cpu_uses_library_convs():
    """channels_last_3d on CPU must not reach for the HIP kernels."""
    net = networks.reg3d(16, 16, 3).to(memory_format=torch.channels_last_3d)
    (out,) = _runs(net, torch.randn(1, 8, 16, 16, 16))
    assert out.shape == (1, 8, 16, 16)


def test_reg2d_cpu_forward_backward():
    net = networks.reg2d(16, 8)
    (out,) = _runs(net, torch.randn(1, 4, 16, 16, 16))  # D=4 < 8: the 2-D regulariser
    assert out.shape == (1, 4, 16, 16)


def test_fpn4_cpu_forward_backward():
    net = networks.FPN4(8, 2)
    outs = _runs(net, torch.rand(1, 3, 64, 96))
    assert outs[0].shape[-2:] == (16, 24)


@pytest.mark.parametrize("layers", [18, 50])
def test_encoder_decoder_cpu(layers):
    enc = networks.ResnetEncoder(layers)
    dec = networks.DepthDecoder(enc.num_ch_enc)
    feats = enc(torch.rand(1, 3, 64, 96))
    out = dec(feats)
    assert out[("disp", 0)].shape == (1, 1, 64, 96)


def test_pose_and_upsample_layers_cpu():
    enc = networks.ResnetEncoder(18, num_input_images=2)
    pose = networks.PoseDecoder(enc.num_ch_enc, num_input_features=1, num_frames_to_predict_for=2)
    aa, tr = pose([enc(torch.rand(1, 6, 64, 96))])
    assert aa.shape == (1, 2, 1, 3) and tr.shape == (1, 2, 1, 3)
    up = networks.convex_upsample_layer(feature_dim=32, scale=2)
    d = up(torch.rand(1, 16, 24), torch.rand(1, 32, 16, 24))
    assert d.shape[-2:] == (64, 96)


def test_default_models_to_load_skips_models_this_configuration_does_not_build(tmp_path, capsys):
    """--models_to_load defaults to the reference's list, which names 'up' (built only with --convex_up) and the pose networks
    (absent under --load_pose): those are skipped with a note; a name outside the known set still raises, as upstream."""
    import pytest
    import torch

    from movedepth_amd.options import MovedepthOptions
    from movedepth_amd.trainer import Trainer

    opt = MovedepthOptions().parse([])
    assert "up" in opt.models_to_load
    t = Trainer.__new__(Trainer)
    t.opt = opt
    t.models = {"reg3d": torch.nn.Linear(2, 2)}
    torch.save(t.models["reg3d"].state_dict(), str(tmp_path / "reg3d.pth"))
    t._load_one(str(tmp_path), "reg3d")
    for n in ("up", "pose_encoder", "pose"):
        t._load_one(str(tmp_path), n)          # not built here: skipped
    assert "skipped" in capsys.readouterr().out
    with pytest.raises(KeyError):
        t._load_one(str(tmp_path), "pose, reg3d")
    with pytest.raises(FileNotFoundError):
        t.models["mono_depth"] = torch.nn.Linear(2, 2)
        t._load_one(str(tmp_path), "mono_depth")
