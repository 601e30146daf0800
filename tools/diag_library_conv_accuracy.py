"""GPU diagnostic: accuracy of the LIBRARY 16->16 3x3x3 convolution against fp64, with and without MIOpen's solver
search (torch.backends.cudnn.benchmark), at the shape of the reg3d parity test and at BASELINE config 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from movedepth_amd import ops

rel = lambda a, b: ((a.double().cpu() - b) .norm() / b.norm()).item()
cl = lambda t: t.contiguous(memory_format=torch.channels_last_3d)
ARGS = ([1] * 3, [1] * 3, [1] * 3, False, [0] * 3, 1)
for shape in ((2, 16, 24, 32), (1, 96, 48, 160)):
    B, D, H, W = shape
    torch.manual_seed(0)
    x = torch.randn(B, 16, D, H, W, device="cuda"); w = torch.randn(16, 16, 3, 3, 3, device="cuda") * 0.05
    gy = torch.randn(B, 16, D, H, W, device="cuda")
    x64 = x.double().cpu().requires_grad_(True); w64 = w.double().cpu().requires_grad_(True)
    y64 = torch.nn.functional.conv3d(x64, w64, padding=1)
    dx64, dw64 = torch.autograd.grad(y64, (x64, w64), gy.double().cpu())
    y64 = y64.detach()
    print("shape", shape)
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        y = torch.ops.aten.convolution(cl(x), cl(w), None, *ARGS)
        dx, dw, _ = torch.ops.aten.convolution_backward(cl(gy), cl(x), cl(w), None, *ARGS, [True, True, False])
        print("  library, solver search %-5s  y %.2e  dx %.2e  dw %.2e" % (bench, rel(y, y64), rel(dx, dx64), rel(dw, dw64)))
    xi, wi = cl(x).requires_grad_(True), w.clone().requires_grad_(True)
    y = ops.conv3d_16(xi, wi); dx, dw = torch.autograd.grad(y, (xi, wi), cl(gy))
    print("  hand-written MFMA kernels      y %.2e  dx %.2e  dw %.2e" % (rel(y, y64), rel(dx, dx64), rel(dw, dw64)))
