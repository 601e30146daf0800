"""Import the read-only reference checkout (/root/reference) in THIS container.

Only used by tools/gen_golden.py to produce the committed fixtures under
tests/golden/.  Nothing in tests/, bench.py or the product imports this module:
the reference does not exist on the GPU box.

The reference needs modules this image lacks (cv2, tensorboardX, pykitti,
skimage, torchvision).  None of them is touched by the hot path, so empty
stand-in *modules* (no behaviour) are registered before the import; the
reference files themselves are never edited, copied or byte-compiled.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _tv_resnet_from_build():
    """torchvision.models-shaped ResNet classes (constructor signatures, `_make_layer`, module names) assembled from THIS
    repo's own ResNet blocks (movedepth_amd.networks._BasicBlock / _Bottleneck), so that the reference's ResnetEncoder /
    ResNetMultiImageInput (networks/resnet_encoder.py:21-121) can be instantiated without torchvision.
    Parameter names = torchvision's = the build's, so state_dicts move between the two freely."""
    import torch.nn as nn

    from movedepth_amd.networks import _BasicBlock, _Bottleneck

    class ResNet(nn.Module):
        def __init__(self, block, layers, num_classes=1000, **_):
            super().__init__()
            self.inplanes = 64
            self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
            self.layer1 = self._make_layer(block, 64, layers[0])
            self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
            self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
            self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))   # deleted by the reference's ResnetEncoder (:103-104)
            self.fc = nn.Linear(512 * block.expansion, num_classes)

        def _make_layer(self, block, planes, blocks, stride=1):
            down = None
            if stride != 1 or self.inplanes != planes * block.expansion:
                down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                     nn.BatchNorm2d(planes * block.expansion))
            mods = [block(self.inplanes, planes, stride, down)]
            self.inplanes = planes * block.expansion
            mods += [block(self.inplanes, planes) for _ in range(1, blocks)]
            return nn.Sequential(*mods)

    def resnet18(pretrained=False, **kw):
        return ResNet(_BasicBlock, [2, 2, 2, 2], **kw)

    def resnet50(pretrained=False, **kw):
        return ResNet(_Bottleneck, [3, 4, 6, 3], **kw)

    return dict(ResNet=ResNet, BasicBlock=_BasicBlock, Bottleneck=_Bottleneck, resnet18=resnet18, resnet50=resnet50)


def load_reference(with_trainer=False, working_resnet=False):
    """Returns (layers_module, Trainer or None, networks or None).  working_resnet: the torchvision stand-in carries
    instantiable ResNet-18/50 classes (built from this repo's blocks) instead of bare names, for full process_batch runs."""
    sys.dont_write_bytecode = True  # never write __pycache__ into the reference tree
    import torch  # noqa: F401  (import before trainer.py pins OMP_NUM_THREADS)

    if "cv2" not in sys.modules:
        _stub("cv2", setNumThreads=lambda n: None)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import movedepth.layers as L

    if not with_trainer:
        return L, None, None

    import torch.nn as nn

    _stub("tensorboardX", SummaryWriter=object)
    _stub("pykitti")
    sk = _stub("skimage")
    sk.transform = _stub("skimage.transform")

    # torchvision stand-in: class names only, enough for `import` to succeed.
    class _ResNet(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tv = _stub("torchvision")
    if working_resnet:
        r = _tv_resnet_from_build()
        resnet_mod = _stub("torchvision.models.resnet", BasicBlock=r["BasicBlock"], Bottleneck=r["Bottleneck"], model_urls={})
        models = _stub("torchvision.models", ResNet=r["ResNet"], resnet=resnet_mod, resnet18=r["resnet18"], resnet34=None,
                       resnet50=r["resnet50"], resnet101=None, resnet152=None)
    else:
        resnet_mod = _stub("torchvision.models.resnet", BasicBlock=object, Bottleneck=object, model_urls={})
        models = _stub("torchvision.models", ResNet=_ResNet, resnet=resnet_mod,
                       resnet18=None, resnet34=None, resnet50=None, resnet101=None, resnet152=None)
    transforms = _stub("torchvision.transforms")
    tv.models = models
    tv.transforms = transforms

    from movedepth.trainer import Trainer
    from movedepth import networks

    # the shipped in-place `out += x` breaks autograd on torch>=2;
    # same forward values, non in-place.
    def _uncert_forward(self, x):
        out = self.conv1(x)
        out = self.conv2(out)
        out = out + x
        out = self.head_convs(out)
        return torch.sigmoid(out)

    networks.UncertNet.forward = _uncert_forward
    return L, Trainer, networks
