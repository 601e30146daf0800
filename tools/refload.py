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


def load_reference(with_trainer=False):
    """Returns (layers_module, Trainer or None, networks or None)."""
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
