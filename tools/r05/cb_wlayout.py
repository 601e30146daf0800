#!/usr/bin/env python3
"""Does the weights' memory order matter to the bf16 x 3 forward?  The same convolution with the weight tensor stored (Co,Ci,taps)
channels-last (co stride Ci*27, ci stride 1) and stored transposed (co stride 1, ci stride Co*27)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from movedepth_amd import ops

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n

cl = torch.channels_last_3d
for C, D, H, W in ((32, 48, 24, 80), (16, 96, 48, 160)):
    B = 6
    x = torch.randn(B, C, D, H, W, device="cuda").contiguous(memory_format=cl)
    gy = torch.randn(B, C, D, H, W, device="cuda").contiguous(memory_format=cl)
    w = (torch.randn(C, C, 3, 3, 3, device="cuda") * 0.05)
    variants = {"contiguous (co, ci, k)": w.contiguous(), "channels_last_3d (co, k, ci)": w.contiguous(memory_format=cl),
                "transposed storage (ci, k, co)": w.permute(1, 0, 2, 3, 4).contiguous(memory_format=cl).permute(1, 0, 2, 3, 4)}
    f = ops.conv3d_cb if C > 16 else ops.conv3d_16
    ref = None
    for name, wv in variants.items():
        assert torch.equal(wv, w)
        xr = x.clone().requires_grad_(True)
        y = f(xr, wv)
        if ref is None: ref = y.detach().clone()
        assert torch.equal(y.detach(), ref)
        tf = t(lambda: f(x, wv))
        td = t(lambda: torch.autograd.grad(y, xr, gy, retain_graph=True))
        print("%d -> %d %-32s strides %-22s forward %6.1f us   data gradient %6.1f us" % (C, C, name, tuple(wv.stride())[:2] + (wv.stride()[4],), tf, td))
