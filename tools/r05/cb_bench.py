#!/usr/bin/env python3
"""reg3d's interior stride-1 layers at BASELINE config 2: md_conv3d_cb_* (16 x 16 channel blocks on the bf16 x 3 kernels) against the library."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from movedepth_amd import ops, miopen_setup
miopen_setup.use_shipped_cache(0)
torch.backends.cudnn.benchmark = True

def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n

for C, D, H, W in ((32, 48, 24, 80), (64, 24, 12, 40), (128, 12, 6, 20)):
    B = 6
    cl = torch.channels_last_3d
    x = torch.randn(B, C, D, H, W, device="cuda").contiguous(memory_format=cl)
    w = (torch.randn(C, C, 3, 3, 3, device="cuda") * 0.05).contiguous(memory_format=cl)
    gy = torch.randn(B, C, D, H, W, device="cuda").contiguous(memory_format=cl)
    gf = 2 * B * D * H * W * C * C * 27 / 1e9
    A = ([1] * 3, [1] * 3, [1] * 3, False, [0] * 3, 1)
    lib_f = t(lambda: torch.ops.aten.convolution(x, w, None, *A))
    lib_d = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, *A, [True, False, False]))
    lib_w = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, *A, [False, True, False]))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yd, yw = ops.conv3d_cb(xr, w), ops.conv3d_cb(x, wr)      # (needs_input_grad is fixed at forward time: one graph per gradient)
    my_f = t(lambda: ops.conv3d_cb(x, w))
    my_d = t(lambda: torch.autograd.grad(yd, xr, gy, retain_graph=True))
    my_w = t(lambda: torch.autograd.grad(yw, wr, gy, retain_graph=True))
    print("%3d -> %3d  %dx%dx%dx%d  %.1f GFLOP per direction: forward lib %6.1f us / blocks %6.1f us;  data gradient %6.1f / %6.1f;  weight gradient %6.1f / %6.1f"
          % (C, C, B, D, H, W, gf, lib_f, my_f, lib_d, my_d, lib_w, my_w))
