"""Network definitions the training step needs (producers/consumers of the hot path's tensors).

These are dense convolutions: they run on MIOpen/rocBLAS (MFMA), exactly as the north star reserves MFMA for
the encoder convs; no hand-written kernels here.  The definitions are this package's own (torchvision is not
available in the image) but keep the reference's parameter names, so its released `.pth` files load
unchanged (`encoder.conv1.weight`, `decoder.0.conv.conv.weight`, `net.0.weight`, `conv7.0.weight`, ...).

Reference (movedepth/networks/): ResnetEncoder resnet_encoder.py:74-121, FPN4 :311-391, reg3d :227-280,
reg2d :184-225, DepthDecoder depth_decoder.py:10-101, UncertNet :371-393, PoseDecoder pose_decoder.py:8-48,
convex_upsample_layer layers.py:184-198.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def _bn_act(bn, relu, x):
    """relu(bn(x)) -- one kernel when `bn` is a HipSyncBatchNorm that carries the ReLU (convert_hip_sync_batchnorm)"""
    y = bn(x)
    return y if getattr(bn, "relu", False) else relu(y)


# --------------------------------------------------------------------------- ResNet (torchvision-compatible keys)
class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = _bn_act(self.bn1, self.relu, self.conv1(x))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = _bn_act(self.bn1, self.relu, self.conv1(x))
        y = _bn_act(self.bn2, self.relu, self.conv2(y))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class _ResNetTrunk(nn.Module):
    """conv1 .. layer4 of a torchvision ResNet (no avgpool / fc), `in_images` stacked RGB frames as input."""

    def __init__(self, num_layers, in_images=1):
        super().__init__()
        block, reps = {18: (_BasicBlock, [2, 2, 2, 2]), 34: (_BasicBlock, [3, 4, 6, 3]),
                       50: (_Bottleneck, [3, 4, 6, 3])}[num_layers]
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3 * in_images, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._stage(block, 64, reps[0], 1)
        self.layer2 = self._stage(block, 128, reps[1], 2)
        self.layer3 = self._stage(block, 256, reps[2], 2)
        self.layer4 = self._stage(block, 512, reps[3], 2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _stage(self, block, planes, n, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*layers)


class ResnetEncoder(nn.Module):
    """Mono / pose encoder: 5 feature maps at strides 2..32; input normalised as (x - 0.45) / 0.225."""

    def __init__(self, num_layers, pretrained=False, num_input_images=1, **kwargs):
        super().__init__()
        if num_layers not in (18, 34, 50):
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4
        self.encoder = _ResNetTrunk(num_layers, num_input_images)
        if pretrained:
            self._load_imagenet(num_layers, num_input_images)

    def _load_imagenet(self, num_layers, num_input_images):
        import glob
        import os

        hits = glob.glob(os.path.join(os.path.dirname(__file__), "..", "pretrain_resnet", "resnet%d-*.pth" % num_layers))
        if not hits:
            raise FileNotFoundError("pretrained ResNet-%d weights not found under pretrain_resnet/ (no network access); "
                                    "use --weights_init scratch" % num_layers)
        sd = torch.load(hits[0], map_location="cpu")
        sd = {k: v for k, v in sd.items() if not k.startswith("fc.")}
        if num_input_images > 1:
            sd["conv1.weight"] = torch.cat([sd["conv1.weight"]] * num_input_images, 1) / num_input_images
        self.encoder.load_state_dict(sd)

    def forward(self, input_image):
        e = self.encoder
        x = (input_image - 0.45) / 0.225
        f0 = _bn_act(e.bn1, e.relu, e.conv1(x))
        f1 = e.layer1(e.maxpool(f0))
        f2 = e.layer2(f1)
        f3 = e.layer3(f2)
        f4 = e.layer4(f3)
        self.features = [f0, f1, f2, f3, f4]
        return self.features


# --------------------------------------------------------------------------- monodepth2-style decoders
class Conv3x3(nn.Module):
    def __init__(self, cin, cout, use_refl=True):
        super().__init__()
        self.pad = nn.ReflectionPad2d(1) if use_refl else nn.ZeroPad2d(1)
        self.conv = nn.Conv2d(int(cin), int(cout), 3)

    def forward(self, x):
        return self.conv(self.pad(x))


class ConvBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = Conv3x3(cin, cout)
        self.nonlin = nn.ELU(inplace=True)

    def forward(self, x):
        return self.nonlin(self.conv(x))


class DepthDecoder(nn.Module):
    """Sigmoid disparity at the requested scales.  `decoder` holds the convs in the reference's order:
    (upconv i 0, upconv i 1) for i = 4..0, then one dispconv per scale."""

    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True, **_unused):
        super().__init__()
        self.scales = list(scales)
        self.use_skips = use_skips
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        convs, self._idx = [], {}
        for i in range(4, -1, -1):
            cin = num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self._idx[("upconv", i, 0)] = len(convs)
            convs.append(ConvBlock(cin, self.num_ch_dec[i]))
            cin = self.num_ch_dec[i] + (num_ch_enc[i - 1] if (use_skips and i > 0) else 0)
            self._idx[("upconv", i, 1)] = len(convs)
            convs.append(ConvBlock(cin, self.num_ch_dec[i]))
        for s in self.scales:
            self._idx[("dispconv", s)] = len(convs)
            convs.append(Conv3x3(self.num_ch_dec[s], num_output_channels))
        self.decoder = nn.ModuleList(convs)
        self.sigmoid = nn.Sigmoid()

    def forward(self, input_features, no_disp=False, no_match=True, outs=0):
        outputs = {}
        x = input_features[-1]
        for i in range(4, -1 + outs, -1):
            x = self.decoder[self._idx[("upconv", i, 0)]](x)
            x = [F.interpolate(x, scale_factor=2, mode="nearest")]
            if self.use_skips and i > 0:
                x.append(input_features[i - 1])
            x = self.decoder[self._idx[("upconv", i, 1)]](torch.cat(x, 1))
            if i in self.scales and not no_disp:
                outputs[("disp", i)] = self.sigmoid(self.decoder[self._idx[("dispconv", i)]](x))
        self.outputs = outputs
        return outputs


class PoseDecoder(nn.Module):
    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        if num_frames_to_predict_for is None:
            num_frames_to_predict_for = num_input_features - 1
        self.num_frames_to_predict_for = num_frames_to_predict_for
        self.net = nn.ModuleList([
            nn.Conv2d(int(num_ch_enc[-1]), 256, 1),                       # squeeze
            nn.Conv2d(num_input_features * 256, 256, 3, stride, 1),       # pose 0
            nn.Conv2d(256, 256, 3, stride, 1),                            # pose 1
            nn.Conv2d(256, 6 * num_frames_to_predict_for, 1),             # pose 2
        ])
        self.relu = nn.ReLU()

    def forward(self, input_features):
        feats = [self.relu(self.net[0](f[-1])) for f in input_features]
        out = torch.cat(feats, 1)
        out = self.relu(self.net[1](out))
        out = self.relu(self.net[2](out))
        out = self.net[3](out).mean(3).mean(2)
        out = 0.01 * out.view(-1, self.num_frames_to_predict_for, 1, 6)
        return out[..., :3], out[..., 3:]


class UncertNet(nn.Module):
    """Entropy map -> trust-mono mask.  The reference adds the residual in place on a ReLU output
    (depth_decoder.py:390), which autograd rejects on current PyTorch; same values, out of place here."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(1, 8, 3, 1, 1, bias=False), nn.BatchNorm2d(8), nn.ReLU())
        self.conv2 = nn.Sequential(nn.Conv2d(8, 8, 3, 1, 1, bias=False), nn.BatchNorm2d(8), nn.ReLU())
        self.head_convs = nn.Conv2d(8, 1, 3, 1, 1, bias=False)

    def forward(self, x):
        out = self.conv2(self.conv1(x))
        return torch.sigmoid(self.head_convs(out + x))


# --------------------------------------------------------------------------- MVS feature net / regularisers
class Conv2d(nn.Module):
    """conv + BN (+ ReLU), the MVSNet-style unit of FPN4."""

    def __init__(self, cin, cout, kernel_size, stride=1, relu=True, bn=True, bn_momentum=0.1, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = nn.BatchNorm2d(cout, momentum=bn_momentum) if bn else None
        self.relu = relu

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
            if self.relu and getattr(self.bn, "relu", False):   # HipSyncBatchNorm carrying the ReLU
                return x
        return F.relu(x, inplace=True) if self.relu else x


class FPN4(nn.Module):
    """Matching features at 1/2**scale resolution (scale=2: 32 channels at H/4 x W/4) + a context feature."""

    def __init__(self, base_channels, scale=0, dcn=False):
        super().__init__()
        if dcn:
            raise NotImplementedError("--dcn needs the external DeformConvPack CUDA extension (not vendored upstream)")
        c = base_channels
        self.scale = scale
        self.conv0 = nn.Sequential(Conv2d(3, c, 3, 1, padding=1), Conv2d(c, c, 3, 1, padding=1))
        self.conv1 = nn.Sequential(Conv2d(c, 2 * c, 5, stride=2, padding=2), Conv2d(2 * c, 2 * c, 3, 1, padding=1),
                                   Conv2d(2 * c, 2 * c, 3, 1, padding=1))
        self.conv2 = nn.Sequential(Conv2d(2 * c, 4 * c, 5, stride=2, padding=2), Conv2d(4 * c, 4 * c, 3, 1, padding=1),
                                   Conv2d(4 * c, 4 * c, 3, 1, padding=1))
        self.conv3 = nn.Sequential(Conv2d(4 * c, 8 * c, 5, stride=2, padding=2), Conv2d(8 * c, 8 * c, 3, 1, padding=1),
                                   Conv2d(8 * c, 8 * c, 3, 1, padding=1))
        final = 8 * c
        if scale < 3:
            self.inner1 = nn.Conv2d(4 * c, final, 1, bias=True)
        if scale < 2:
            self.inner2 = nn.Conv2d(2 * c, final, 1, bias=True)
        if scale < 1:
            self.inner3 = nn.Conv2d(c, final, 1, bias=True)
        if scale == 3:
            self.out = nn.Conv2d(final, 8 * c, 1, bias=False)
        else:
            self.out = nn.Conv2d(final, c * 2 ** scale, 3, padding=1, bias=False)

    def forward(self, x):
        c0 = self.conv0(x)
        c1 = self.conv1(c0)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        f = c3
        if self.scale < 3:
            f = F.interpolate(f, scale_factor=2, mode="bilinear", align_corners=True) + self.inner1(c2)
        if self.scale < 2:
            f = F.interpolate(f, scale_factor=2, mode="bilinear", align_corners=True) + self.inner2(c1)
        if self.scale < 1:
            f = F.interpolate(f, scale_factor=2, mode="bilinear", align_corners=True) + self.inner3(c0)
        return self.out(f), (c3, c2, c1, c0)[3 - self.scale]


def _set_benchmark(value, grad):
    torch.backends.cudnn.benchmark = value
    return grad


class ConvBnReLU3D(nn.Module):
    def __init__(self, cin, cout, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm3d(cout)

    # True (reg3d.conv2, set by the trainer: --hip_conv2): the convolution as 16 x 16 channel blocks on the 16 -> 16 layer's bf16 x 3
    # kernels (ops.conv3d_cb) where they apply -- fp32 channels-last activations outside autocast, 3 x 3 x 3, stride 1, padding 1
    hip_cb = False

    def _conv(self, x):
        c = self.conv
        if self.hip_cb and x.is_cuda and not torch.is_autocast_enabled() and c.stride == (1, 1, 1) and c.padding == (1, 1, 1) and \
                c.dilation == (1, 1, 1) and c.groups == 1 and c.bias is None and ops.conv3d_cb_supported(x, c.weight) and \
                x.is_contiguous(memory_format=torch.channels_last_3d):
            return ops.conv3d_cb(x, c.weight)
        return c(x)

    def forward(self, x):
        y = self.bn(self._conv(x))
        # the fused modules include the ReLU
        return y if (isinstance(self.bn, FusedBNReLU3d) or getattr(self.bn, "relu", False)) else F.relu(y, inplace=True)


class FusedBNReLU3d(nn.Module):
    """BatchNorm3d followed by ReLU (and optionally a residual add) with BatchNorm3d's parameter / buffer names, so that
    checkpoints are interchangeable.  On the GPU, in training mode, on a channels_last_3d 16-channel tensor this is
    ops.bn_relu_3d (3 + 5 passes over the tensor instead of 13); anywhere else it is the torch ops.  Not a _BatchNorm
    subclass on purpose: SyncBatchNorm conversion skips it, `sync_group` makes it reduce its statistics itself."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.fused = True
        self.sync_group = None  # torch.distributed group (or dist.group.WORLD) for SyncBatchNorm semantics

    def forward(self, x, res=None):
        if self.training:
            self.num_batches_tracked.add_(1)
        if (self.fused and self.training and x.is_cuda and x.dim() == 5 and x.shape[1] in ops.BN_RELU_CHANNELS and
                x.is_contiguous(memory_format=torch.channels_last_3d)):
            return ops.bn_relu_3d(x, self.weight, self.bias, res, self.running_mean, self.running_var, self.momentum, self.eps,
                                  self.sync_group)
        if self.training and self.sync_group is not None:
            raise RuntimeError("FusedBNReLU3d: synchronised statistics need the fused GPU path (channels_last_3d, 16 channels)")
        y = F.relu(F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, self.training, self.momentum,
                                self.eps))
        return y if res is None else y + res


class HipSyncBatchNorm(nn.Module):
    """BatchNorm2d / BatchNorm3d with statistics over the global batch of `sync_group` (torch.nn.SyncBatchNorm's semantics; the
    reference's --ddp path converts every BatchNorm to it, trainer.py:69-135) on the hand-written kernels of csrc/syncbn.hip:
    one all-reduce of 2C sums per layer and direction.  Parameter / buffer names are BatchNorm's, so checkpoints are
    interchangeable.  Not a _BatchNorm subclass on purpose (torch's SyncBatchNorm conversion must skip it).  `relu`: the ReLU
    that follows the layer, fused (the caller then skips its own).  In evaluation mode, or where the kernels do not apply
    (CPU tensors, channel counts that are not multiples of 4), it is F.batch_norm -- without synchronisation, so training
    such a layer under a group raises."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, relu=False):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.relu = num_features, eps, momentum, relu
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.sync_group = None

    @classmethod
    def from_batchnorm(cls, bn, group=None):
        if bn.momentum is None:
            # cumulative moving average (momentum = 1 / num_batches_tracked): the kernels take a fixed momentum
            raise NotImplementedError("HipSyncBatchNorm: BatchNorm with momentum=None (cumulative average) is not supported")
        m = cls(bn.num_features, bn.eps, bn.momentum)
        if bn.affine:
            # the SAME Parameter objects, as torch's convert_sync_batchnorm does: an optimizer or parameter list built before the
            # conversion keeps training the tensors the layer uses
            m.weight, m.bias = bn.weight, bn.bias
        m.running_mean, m.running_var, m.num_batches_tracked = bn.running_mean, bn.running_var, bn.num_batches_tracked
        m.sync_group = group
        m.train(bn.training)
        return m.to(bn.weight.device if bn.affine else bn.running_mean.device)

    def forward(self, x):
        if self.training:
            self.num_batches_tracked.add_(1)
            if ops.sync_batch_norm_supported(x):
                return ops._SyncBatchNorm.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.momentum, self.eps,
                                                int(self.relu), self.sync_group)
            if self.sync_group is not None:
                raise RuntimeError("HipSyncBatchNorm: synchronised statistics need a GPU tensor with a multiple of 4 channels, got %s %s"
                                   % (x.device, tuple(x.shape)))
        y = F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, self.training, self.momentum, self.eps)
        return F.relu(y) if self.relu else y


def convert_hip_sync_batchnorm(module, group=None, fuse_relu=True):
    """Every BatchNorm2d / BatchNorm3d below `module` -> HipSyncBatchNorm sharing its parameters and buffers (what
    torch.nn.SyncBatchNorm.convert_sync_batchnorm does for torch's layer).  FusedBNReLU3d layers keep their own kernels and get
    `group` as their sync_group."""
    if isinstance(module, torch.nn.modules.batchnorm._BatchNorm) and not isinstance(module, nn.BatchNorm1d) and \
            module.track_running_stats and module.num_features % 4 == 0 and module.momentum is not None:
        return HipSyncBatchNorm.from_batchnorm(module, group)
    if isinstance(module, FusedBNReLU3d):
        module.sync_group = group
        return module
    if isinstance(module, torch.nn.modules.batchnorm._BatchNorm) and not isinstance(module, nn.SyncBatchNorm) and group is not None:
        # what the kernels do not take (BatchNorm1d, channel counts that are not multiples of 4, no running statistics) must not
        # silently stay an unsynchronised layer under --ddp --sync_bn: the reference converts EVERY BatchNorm
        # (trainer.py:69-135, convert_sync_batchnorm) -- so does this, with torch's own module for the leftovers
        if callable(group) and not isinstance(group, torch.distributed.ProcessGroup):
            # a rccl_direct.DirectAllReduce: torch's SyncBatchNorm would drive the direct path's OWN communicator from torch's stream
            # while ncclAllReduce runs on it from the compute stream -- one communicator on two streams, the hang rccl_direct's header
            # rules out.  The shipped networks have no such layer; a model that does must run without MD_DIRECT_RCCL.
            raise RuntimeError("convert_hip_sync_batchnorm: %s cannot run on the HIP kernels (BatchNorm1d, channels not a multiple of 4, "
                               "no running statistics or momentum=None) and the direct-RCCL path is on: run without MD_DIRECT_RCCL=1"
                               % type(module).__name__)
        return nn.SyncBatchNorm.convert_sync_batchnorm(module, group)
    for name, child in list(module.named_children()):
        new = convert_hip_sync_batchnorm(child, group, fuse_relu)
        if new is not child:
            setattr(module, name, new)
    if not fuse_relu:
        return module
    # the ReLU that follows a normalisation moves into its kernels (one launch fewer each way, and no separate pass over the tensor)
    # wherever the containing module is one of ours and calls the ReLU right after the layer
    hip = lambda m: isinstance(m, HipSyncBatchNorm)
    if isinstance(module, (_BasicBlock, _ResNetTrunk)) and hip(module.bn1):
        module.bn1.relu = True
    elif isinstance(module, _Bottleneck):
        for bn in (module.bn1, module.bn2):
            if hip(bn):
                bn.relu = True
    elif isinstance(module, Conv2d) and module.relu and hip(module.bn):
        module.bn.relu = True
    elif isinstance(module, ConvBnReLU3D) and hip(module.bn):
        module.bn.relu = True
    elif isinstance(module, nn.Sequential):
        kids = list(module.named_children())
        for (n0, m0), (n1, m1) in zip(kids, kids[1:]):
            if hip(m0) and isinstance(m1, nn.ReLU):
                m0.relu = True
                setattr(module, n1, nn.Identity())   # no parameters: the state_dict keeps its keys
    return module


def _up3d(cin, cout, k=3, pad=1, opad=1, stride=2):
    return nn.Sequential(nn.ConvTranspose3d(cin, cout, kernel_size=k, padding=pad, output_padding=opad, stride=stride,
                                            bias=False), nn.BatchNorm3d(cout), nn.ReLU(inplace=True))


class reg3d(nn.Module):
    """3-D U-Net over the grouped cost volume.  Input (B,D,G,h,w) as in the reference; its first op is the
    permute to (B,G,D,h,w), which is a no-op copy-wise when the volume was written in that layout by
    movedepth_amd.ops.costvol_grouped(layout='bgd')."""

    def __init__(self, in_channels, base_channels, down_size=3, fused_bn=False):
        super().__init__()
        c = base_channels
        self.down_size = down_size
        self.conv0 = ConvBnReLU3D(in_channels, c)
        self.conv1 = ConvBnReLU3D(c, 2 * c, stride=2)
        self.conv2 = ConvBnReLU3D(2 * c, 2 * c)
        if down_size >= 2:
            self.conv3 = ConvBnReLU3D(2 * c, 4 * c, stride=2)
            self.conv4 = ConvBnReLU3D(4 * c, 4 * c)
        if down_size >= 3:
            self.conv5 = ConvBnReLU3D(4 * c, 8 * c, stride=2)
            self.conv6 = ConvBnReLU3D(8 * c, 8 * c)
            self.conv7 = _up3d(8 * c, 4 * c)
        if down_size >= 2:
            self.conv9 = _up3d(4 * c, 2 * c)
        self.conv11 = _up3d(2 * c, c)
        self.prob = nn.Conv3d(c, 1, 3, stride=1, padding=1, bias=False)
        # BatchNorm + ReLU (+ the skip add after the three up-convolutions) on the fused kernels wherever the layer has 16, 32
        # or 64 channels: same state_dict keys (convN.bn.*, convM.1.*).
        # fused_bn / --hip_bn_relu: 1.3 ms per step faster than BatchNorm3d + ReLU + add on the two full-resolution layers alone
        # once the statistics are finalised in a kernel (with a dozen host-side tensor ops per call doing that, 0.6 ms slower).
        self.fused_bn = bool(fused_bn)
        self.fused_up = {}
        if self.fused_bn:
            # only the two full-resolution layers: with the half / quarter resolution ones fused too (32 and 64 channels,
            # 70 and 18 MB tensors) the step got slower again (49.43, 49.52 ms against 47.80, 47.81): five launches per
            # layer and direction cost more than the passes they save on small tensors
            for name in ("conv0",):
                m = getattr(self, name, None)
                if m is not None and m.bn.num_features in ops.BN_RELU_CHANNELS:
                    m.bn = FusedBNReLU3d(m.bn.num_features)
            for name in ("conv11",):
                m = getattr(self, name, None)
                if m is not None and m[1].num_features in ops.BN_RELU_CHANNELS:
                    m[1] = FusedBNReLU3d(m[1].num_features)
                    m[2] = nn.Identity()
                    self.fused_up[name] = True

    # MIOpen solver search ("find") only for this module's convolutions: without it the fp32 3-D convs fall back to
    # naive kernels (1.67 s fwd+bwd), with it for every conv of the model the first step takes ~18 min of kernel
    # compilation.  torch reads the benchmark flag when a conv (or its backward) executes, so it is switched on
    # around the forward and, through tensor hooks, around this module's part of the backward pass.
    find_convs = True
    hip_conv0_wgrad = True        # False: the whole first convolution on the library
    lib_conv0_fwd_dgrad = False   # True: only its weight gradient hand-written (A/B)
    hip_prob = True   # False: keep `prob` on the library convolution too (used by the A/B in tools/ and tests)

    def forward(self, inputs):
        if not self.find_convs or not inputs.is_cuda:
            return self._forward(inputs)
        prev = torch.backends.cudnn.benchmark
        torch.backends.cudnn.benchmark = True
        try:
            out = self._forward(inputs)
        finally:
            torch.backends.cudnn.benchmark = prev
        if out.requires_grad:
            # backward: switch the search on when the gradient reaches this module's output and off again when it leaves
            # the module -- at the input's gradient if the input needs one, otherwise at the weight gradient of the first
            # library convolution (the last node of this module's backward).  One-shot hooks, so a detached input can
            # never leave the process-wide flag on for the 2-D networks' backward.
            first = self.conv1.conv if (self.hip_conv0_wgrad and not self.lib_conv0_fwd_dgrad) else self.conv0.conv
            state = {}

            def leave(g):
                torch.backends.cudnn.benchmark = prev
                h = state.pop("h", None)
                if h is not None:
                    h.remove()
                return g

            def enter(g):
                torch.backends.cudnn.benchmark = True
                if not inputs.requires_grad and first.weight.requires_grad:
                    state["h"] = first.weight.register_hook(leave)
                return g

            if inputs.requires_grad:
                inputs.register_hook(leave)
            out.register_hook(enter)
        return out

    def _up(self, name, x, skip):
        """skip + relu(bn(conv_transpose(x))); with the fused module the add rides in the normalisation pass"""
        m = getattr(self, name)
        if self.fused_up.get(name):
            return m[1](m[0](x), res=skip)
        return skip + m(x)

    def _forward(self, inputs):
        x = inputs.permute(0, 2, 1, 3, 4)  # B,D,G,h,w -> B,G,D,h,w (a view)
        # channels_last_3d (NDHWC) when the module was converted to it: MIOpen's fp32 3-D convolutions are ~75x
        # faster in that layout on gfx950 (21.6 ms vs 1.67 s fwd+bwd at 6x16x96x48x160); no copy if the volume was
        # written that way (ops.costvol_grouped(layout='ndhwc'))
        cl = self.conv0.conv.weight.is_contiguous(memory_format=torch.channels_last_3d) and \
            not self.conv0.conv.weight.is_contiguous()
        if cl and x.is_cuda and self.hip_conv0_wgrad and tuple(self.conv0.conv.weight.shape) == (16, 16, 3, 3, 3):
            # first layer on the MFMA kernels (library: 1.0 / 1.4 / 3.1 ms fwd / data gradient / weight gradient); they
            # read a planar (`bgd`) or a channels-last volume in place and hand the gradient back in the same layout
            if not (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last_3d)):
                x = x.contiguous(memory_format=torch.channels_last_3d)
            c0 = self.conv0.bn(ops.conv3d_16(x, self.conv0.conv.weight, self.lib_conv0_fwd_dgrad))
            if not (isinstance(self.conv0.bn, FusedBNReLU3d) or getattr(self.conv0.bn, "relu", False)):   # as ConvBnReLU3D.forward
                c0 = F.relu(c0, inplace=True)
        else:
            x = x.contiguous(memory_format=torch.channels_last_3d) if cl else x.contiguous()
            c0 = self.conv0(x)
        c2 = self.conv2(self.conv1(c0))
        if self.down_size >= 2:
            c4 = self.conv4(self.conv3(c2))
            x = c4
            if self.down_size >= 3:
                x = self._up("conv7", self.conv6(self.conv5(c4)), c4)
            x = self._up("conv9", x, c2)
        else:
            x = c2
        x = self._up("conv11", x, c0)
        # last layer (C -> 1): hand-written kernels instead of the library's GEMM-shaped ones (1177 / 285 / 2568 us
        # fwd / bwd-data / bwd-weight at 6x16x96x48x160 against ~50 us of memory traffic each); other channel counts
        # and the NCDHW mode stay with the library convolution
        if cl and x.is_cuda and x.shape[1] in ops.CONV3D_C1_CHANNELS and self.hip_prob:
            return ops.conv3d_c1(x, self.prob.weight).squeeze(1)
        return self.prob(x).squeeze(1)


class reg2d(nn.Module):
    """Per-hypothesis 2-D regulariser used when num_depth_bins < 8 (1x3x3 kernels)."""

    def __init__(self, input_channel=128, base_channel=32):
        super().__init__()
        c, k, s, p = base_channel, (1, 3, 3), (1, 2, 2), (0, 1, 1)
        self.conv0 = ConvBnReLU3D(input_channel, c, kernel_size=k, pad=p)
        self.conv1 = ConvBnReLU3D(c, 2 * c, kernel_size=k, stride=s, pad=p)
        self.conv2 = ConvBnReLU3D(2 * c, 2 * c)
        self.conv3 = ConvBnReLU3D(2 * c, 4 * c, kernel_size=k, stride=s, pad=p)
        self.conv4 = ConvBnReLU3D(4 * c, 4 * c)
        self.conv5 = ConvBnReLU3D(4 * c, 8 * c, kernel_size=k, stride=s, pad=p)
        self.conv6 = ConvBnReLU3D(8 * c, 8 * c)
        self.conv7 = _up3d(8 * c, 4 * c, k, p, p, s)
        self.conv9 = _up3d(4 * c, 2 * c, k, p, p, s)
        self.conv11 = _up3d(2 * c, c, k, p, p, s)
        self.prob = nn.Conv3d(8, 1, 1, stride=1, padding=0)

    def forward(self, x):
        x = x.permute(0, 2, 1, 3, 4).contiguous()
        c0 = self.conv0(x)
        c2 = self.conv2(self.conv1(c0))
        c4 = self.conv4(self.conv3(c2))
        x = c4 + self.conv7(self.conv6(self.conv5(c4)))
        x = c2 + self.conv9(x)
        x = c0 + self.conv11(x)
        return self.prob(x).squeeze(1)


class convex_upsample_layer(nn.Module):
    """Predicts the 9-tap convex-combination mask from the context feature and upsamples the MVS depth."""

    def __init__(self, feature_dim, scale=2):
        super().__init__()
        self.scale = scale
        self.upsample_mask = nn.Sequential(
            nn.Conv2d(feature_dim, 64, 3, stride=1, padding=1, bias=False),
            nn.ReLU(inplace=True),
            nn.Conv2d(64, (2 ** scale) ** 2 * 9, 1, stride=1, padding=0, bias=False))

    def forward(self, depth, feat):
        from .layers import convex_upsample

        return convex_upsample(depth, self.upsample_mask(feat), self.scale)
