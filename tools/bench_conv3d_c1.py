"""GPU: the hand-written C->1 convolution (reg3d.prob) against the library convolution, per direction, at the
BASELINE config-2 volume.  Bytes are algorithmic: fwd reads x + writes y; bwd-data reads gy + writes dx;
bwd-weight reads x + gy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from movedepth_amd import ops

torch.backends.cudnn.benchmark = True
B, C, D, H, W = 6, int(os.environ.get("C", 16)), 96, 48, 160
x = torch.randn(B, C, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
w = (torch.randn(1, C, 3, 3, 3, device="cuda") * 0.1).contiguous(memory_format=torch.channels_last_3d)
gy = torch.randn(B, 1, D, H, W, device="cuda")
xb, yb = x.numel() * 4, gy.numel() * 4


def ev(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


# one graph per direction: needs_input_grad follows requires_grad of the forward's inputs, so a graph in which both x
# and w require grad runs BOTH backward kernels whatever `inputs=` autograd.grad is given (the first version of this
# script did that and reported the sum of the two as each one's time)
# ROTATE input volumes (default 4 = 1.1 GB) so that no launch finds its input in the 256 MB Infinity Cache: re-reading one
# 283 MB volume measured the weight gradient at 63 us where the training step sees 94 us (profiles/r02_*)
ROT = int(os.environ.get("ROTATE", 4))
xs = [x] + [torch.randn_like(x) for _ in range(ROT - 1)]
gys = [gy] + [torch.randn_like(gy) for _ in range(ROT - 1)]
xrs = [t.clone().requires_grad_(True) for t in xs]
wr = w.clone().requires_grad_(True)
yh_dx = [ops.conv3d_c1(t, w) for t in xrs]
yh_dw = [ops.conv3d_c1(t, wr) for t in xs]
mask = lambda m: torch.ops.aten.convolution_backward(gy, x, w, None, [1] * 3, [1] * 3, [1] * 3, False, [0] * 3, 1, m)
it = [0]


def nxt():
    it[0] += 1
    return it[0] % ROT


def f_fwd():
    ops.conv3d_c1(xs[nxt()], w)


def f_bd():
    i = nxt()
    torch.autograd.grad(yh_dx[i], xrs[i], gys[i], retain_graph=True)


def f_bw():
    i = nxt()
    torch.autograd.grad(yh_dw[i], wr, gys[i], retain_graph=True)


rows = [("fwd", f_fwd, lambda: torch.nn.functional.conv3d(x, w, padding=1), xb + yb),
        ("bwd-data", f_bd, lambda: mask([True, False, False]), xb + yb),
        ("bwd-weight", f_bw, lambda: mask([False, True, False]), xb + yb)]
skip_lib = os.environ.get("NO_LIB") == "1"
print("conv3d_c1 B=%d C=%d %dx%dx%d  lib=%s  rotating over %d input volumes" % (B, C, D, H, W, os.environ.get("MOVEDEPTH_HIP_LIB", "default"), ROT))
for name, mine, lib, nbytes in rows:
    t = ev(mine)
    tl = float("nan") if skip_lib else ev(lib, n=5, warm=2)
    print("  %-10s %8.1f us  %6.0f GB/s (%4.1f%% of 8 TB/s)   library %8.1f us" % (name, t, nbytes / t * 1e-3, nbytes / t * 1e-3 / 80, tl), flush=True)
