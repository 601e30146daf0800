"""GPU: reg3d.conv0's weight gradient, hand-written MFMA kernel against the library, BASELINE config-2 volume."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from movedepth_amd import ops

torch.backends.cudnn.benchmark = True
B, D, H, W = 6, 96, 48, 160
cl = lambda t: t.contiguous(memory_format=torch.channels_last_3d)
x = cl(torch.randn(B, 16, D, H, W, device="cuda"))
w = cl(torch.randn(16, 16, 3, 3, 3, device="cuda") * 0.05).requires_grad_(True)
gy = cl(torch.randn(B, 16, D, H, W, device="cuda"))
y = ops.conv3d_16(x, w)
gflop = 2 * 27 * 16 * 16 * B * D * H * W / 1e9


def ev(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


t = ev(lambda: torch.autograd.grad(y, w, gy, retain_graph=True))
tl = ev(lambda: torch.ops.aten.convolution_backward(gy, x, w.detach(), None, [1] * 3, [1] * 3, [1] * 3, False, [0] * 3, 1,
                                                    [False, True, False]), n=5, warm=2)
print("conv0 weight gradient 16->16 %dx%dx%dx%d: HIP %.1f us = %.1f TF/s (%.1f%% of 157.3 fp32 peak)   library %.1f us = %.1f TF/s"
      % (B, D, H, W, t, gflop / t * 1e3, gflop / t * 1e3 / 157.3 * 100, tl, gflop / tl * 1e3))  # GFLOP / us * 1e3 = TF/s
