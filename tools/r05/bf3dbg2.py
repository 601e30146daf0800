import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from movedepth_amd import ops
B, D, H, W = 1, 3, 8, 40
cl = lambda t: t.contiguous(memory_format=torch.channels_last_3d)
x = torch.zeros(B, 16, D, H, W, device="cuda")
x[:, :, :, :, :] = torch.arange(W, device="cuda").float().view(1, 1, 1, 1, W) + 100.0
x = x * (1 + torch.arange(16, device="cuda").float().view(1, 16, 1, 1, 1) * 0.001)
x = cl(x)
for kw in (0, 2):
    w = torch.zeros(16, 16, 3, 3, 3, device="cuda")
    w[:, :, 1, 1, kw] = torch.eye(16, device="cuda")
    y = ops.conv3d_16(x, w)
    print("kw", kw, "y[0,0,1,3,:] =", [round(v, 1) for v in y[0, 0, 1, 3, :].tolist()])
    print("      y[0,5,1,3,:] =", [round(v, 1) for v in y[0, 5, 1, 3, :].tolist()])
