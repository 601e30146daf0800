"""Stand-alone timing of md_softmax_entropy_localmax fwd / bwd at config 2's shape (B=6, D=96, 48x160): torch events, rotating
over 8 logit buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from movedepth_amd import ops

B, D, h, w = 6, 96, 48, 160
bufs = [torch.randn(B, D, h, w, device="cuda").requires_grad_(True) for _ in range(8)]
mn, mx = torch.full((B, 1, h, w), 0.01, device="cuda"), torch.full((B, 1, h, w), 1.0, device="cuda")
gd, ge = torch.randn(B, 1, h, w, device="cuda"), torch.randn(B, 1, h, w, device="cuda")


def run(n):
    tf = tb = 0.0
    for i in range(n):
        x = bufs[i % 8]
        x.grad = None
        e = [torch.cuda.Event(True) for _ in range(3)]
        e[0].record()
        depth, ent = ops.softmax_entropy_localmax(x, mn, mx, radius=1)[:2]
        e[1].record()
        torch.autograd.backward((depth, ent), (gd.view_as(depth), ge.view_as(ent)))
        e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    return tf / n * 1e3, tb / n * 1e3


run(5)
print("softmax_entropy_localmax  fwd %.1f us  bwd %.1f us (incl. launch + autograd glue)" % run(40))
