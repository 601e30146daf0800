"""GPU diagnostic: accuracy of reg3d.conv0's three implementations against fp64, at kernel level and through the whole
regulariser (where BatchNorm on small batches amplifies rounding differences in the volume gradient)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy
import torch
from movedepth_amd import networks, ops

torch.backends.cudnn.benchmark = False
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
cl = lambda t: t.contiguous(memory_format=torch.channels_last_3d)

torch.manual_seed(0)
B, D, H, W = 2, 16, 24, 32
x = torch.randn(B, 16, D, H, W, device="cuda")
w = torch.randn(16, 16, 3, 3, 3, device="cuda") * 0.05
gy = torch.randn(B, 16, D, H, W, device="cuda")
y64 = torch.nn.functional.conv3d(x.double().cpu(), w.double().cpu(), padding=1)
dx64, dw64 = torch.autograd.grad(torch.nn.functional.conv3d(x.double().cpu().requires_grad_(True), w.double().cpu(), padding=1), [], allow_unused=True) if False else (None, None)
x64 = x.double().cpu().requires_grad_(True); w64 = w.double().cpu().requires_grad_(True)
y64 = torch.nn.functional.conv3d(x64, w64, padding=1)
dx64, dw64 = torch.autograd.grad(y64, (x64, w64), gy.double().cpu())
print("kernel level, rel err vs fp64 (y, dx, dw):")
for name, xin, lib in (("HIP channels-last", cl(x), False), ("HIP planar", x.contiguous(), False), ("library", cl(x), True)):
    xi = xin.detach().requires_grad_(True); wi = w.detach().requires_grad_(True)
    y = ops.conv3d_16(xi, wi, lib)
    dx, dw = torch.autograd.grad(y, (xi, wi), cl(gy))
    if lib:
        dw = torch.ops.aten.convolution_backward(cl(gy), cl(x), w, None, [1]*3, [1]*3, [1]*3, False, [0]*3, 1, [False, True, False])[1]
    print("  %-18s %.2e %.2e %.2e" % (name, rel(y.cpu(), y64), rel(dx.cpu(), dx64), rel(dw.cpu(), dw64)))

print("through reg3d (train-mode BN), rel err of logits / d_volume / d_conv0_weight vs an fp64 copy of the net:")
torch.manual_seed(4)
net = networks.reg3d(16, 16, 3).cuda().to(memory_format=torch.channels_last_3d)
net64 = copy.deepcopy(net).double().cpu()
net64.hip_conv0_wgrad = False; net64.hip_prob = False; net64.find_convs = False
G, h, wd = 16, 24, 32
for layout in ("bdg", "bgd", "ndhwc"):
    torch.manual_seed(7)
    if layout == "bdg":
        vol = torch.randn(B, D, G, h, wd, device="cuda")
    elif layout == "bgd":
        vol = torch.randn(B, G, D, h, wd, device="cuda").permute(0, 2, 1, 3, 4)
    else:
        vol = torch.randn(B, D, h, wd, G, device="cuda").permute(0, 1, 4, 2, 3)
    v64 = vol.double().cpu().requires_grad_(True)
    net64.zero_grad(); o64 = net64(v64); o64.square().mean().backward()
    ref = (o64, v64.grad, net64.conv0.conv.weight.grad)
    for mode, hip, lib_fd in (("HIP all", True, False), ("HIP wgrad", True, True), ("library", False, False)):
        net.hip_conv0_wgrad, net.lib_conv0_fwd_dgrad = hip, lib_fd
        net.zero_grad()
        v = vol.detach().requires_grad_(True)
        o = net(v); o.square().mean().backward()
        got = (o, v.grad, net.conv0.conv.weight.grad)
        print("  %-6s %-10s %s" % (layout, mode, "  ".join("%.2e" % rel(a.cpu(), b) for a, b in zip(got, ref))))
