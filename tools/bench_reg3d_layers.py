"""GPU: per-layer MIOpen time of reg3d's convolutions (fwd / bwd-data / bwd-weight separately), channels_last_3d,
solver search on -- to see which layers dominate the 3-D conv time of the training step.  FLOP rates are against
the direct-convolution count (2 * taps * Cin * Cout per output voxel; transposed convs: per input voxel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.backends.cudnn.benchmark = True
dev = "cuda"
B, D, h, w = 6, 96, 48, 160
c = 16
# name, cin, cout, stride, transposed, input dims divisor
LAYERS = [("conv0", c, c, 1, False, 1), ("conv1", c, 2 * c, 2, False, 1), ("conv2", 2 * c, 2 * c, 1, False, 2),
          ("conv3", 2 * c, 4 * c, 2, False, 2), ("conv4", 4 * c, 4 * c, 1, False, 4), ("conv5", 4 * c, 8 * c, 2, False, 4),
          ("conv6", 8 * c, 8 * c, 1, False, 8), ("conv7", 8 * c, 4 * c, 2, True, 8), ("conv9", 4 * c, 2 * c, 2, True, 4),
          ("conv11", 2 * c, c, 2, True, 2), ("prob", c, 1, 1, False, 1)]


def ev_time(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


tot = [0.0, 0.0, 0.0]
print("%-7s %-22s %9s %9s %9s   GFLOP  TF/s(f,d,w)" % ("layer", "shape", "fwd us", "bwd-d us", "bwd-w us"))
for name, cin, cout, stride, tr, div in LAYERS:
    dims = (D // div, h // div, w // div)
    x = torch.randn(B, cin, *dims, device=dev).contiguous(memory_format=torch.channels_last_3d)
    wshape = (cin, cout, 3, 3, 3) if tr else (cout, cin, 3, 3, 3)
    wt = (torch.randn(*wshape, device=dev) * 0.05).contiguous(memory_format=torch.channels_last_3d)
    opad = [1, 1, 1] if tr else [0, 0, 0]
    conv = lambda: torch.ops.aten.convolution(x, wt, None, [stride] * 3, [1] * 3, [1] * 3, tr, opad, 1)
    y = conv()
    gy = torch.randn_like(y)
    bwd = lambda mask: torch.ops.aten.convolution_backward(gy, x, wt, None, [stride] * 3, [1] * 3, [1] * 3, tr, opad, 1, mask)
    tf = ev_time(conv)
    td = ev_time(lambda: bwd([True, False, False]))
    tw = ev_time(lambda: bwd([False, True, False]))
    vox = (x if tr else y)[:, 0].numel()
    gflop = 2 * 27 * cin * cout * vox / 1e9
    print("%-7s %-22s %9.0f %9.0f %9.0f   %5.1f  %.1f %.1f %.1f" % (name, "%d->%d s%d%s %s" % (cin, cout, stride, "T" if tr else "", "x".join(map(str, dims))),
                                                                     tf, td, tw, gflop, gflop / tf * 1e3, gflop / td * 1e3, gflop / tw * 1e3), flush=True)
    for i, t in enumerate((tf, td, tw)):
        tot[i] += t
print("total  fwd %.2f ms  bwd-data %.2f ms  bwd-weight %.2f ms  (one reg3d pass; the step runs two)" % tuple(t / 1e3 for t in tot))
