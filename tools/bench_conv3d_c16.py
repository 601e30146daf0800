"""GPU: reg3d.conv0 (16->16, 3x3x3) on the hand-written MFMA kernels against the library, BASELINE config-2 volume."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from movedepth_amd import ops

torch.backends.cudnn.benchmark = True
B, D, H, W = 6, 96, 48, 160
cl = lambda t: t.contiguous(memory_format=torch.channels_last_3d)
x = cl(torch.randn(B, 16, D, H, W, device="cuda")).requires_grad_(True)
w = cl(torch.randn(16, 16, 3, 3, 3, device="cuda") * 0.05).requires_grad_(True)
gy = cl(torch.randn(B, 16, D, H, W, device="cuda"))
gflop = 2 * 27 * 16 * 16 * B * D * H * W / 1e9
ARGS = ([1] * 3, [1] * 3, [1] * 3, False, [0] * 3, 1)


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


xd, wd = x.detach(), w.detach()
# one graph per direction, so that the backward of each row runs exactly one kernel (needs_input_grad follows
# requires_grad of the forward's inputs, not the `inputs=` of autograd.grad)
y_dx = ops.conv3d_16(x, wd)
y_dw = ops.conv3d_16(xd, w)
rows = [("fwd", lambda: ops.conv3d_16(xd, wd), lambda: torch.ops.aten.convolution(xd, wd, None, *ARGS)),
        ("bwd-data", lambda: torch.autograd.grad(y_dx, x, gy, retain_graph=True),
         lambda: torch.ops.aten.convolution_backward(gy, xd, wd, None, *ARGS, [True, False, False])),
        ("bwd-weight", lambda: torch.autograd.grad(y_dw, w, gy, retain_graph=True),
         lambda: torch.ops.aten.convolution_backward(gy, xd, wd, None, *ARGS, [False, True, False]))]
print("conv0 16->16 %dx%dx%dx%d, %.1f GFLOP per direction, fp32 MFMA peak 157.3 TF/s" % (B, D, H, W, gflop))
for name, mine, lib in rows:
    t = ev(mine)
    tl = float("nan") if os.environ.get("NO_LIB") == "1" else ev(lib, n=5, warm=2)
    # GFLOP / us * 1e3 = TF/s
    print("  %-10s HIP %7.1f us = %5.1f TF/s (%4.1f%% of peak)   library %7.1f us = %5.1f TF/s"
          % (name, t, gflop / t * 1e3, gflop / t * 1e3 / 157.3 * 100, tl, gflop / tl * 1e3), flush=True)
