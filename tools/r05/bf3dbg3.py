import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from movedepth_amd import ops
B, D, H, W = 1, 4, 8, 40
cl = lambda t: t.contiguous(memory_format=torch.channels_last_3d)
x = torch.zeros(B, 16, D, H, W, device="cuda")
x += torch.arange(W, device="cuda").float().view(1, 1, 1, 1, W)
x += 100.0 * torch.arange(H, device="cuda").float().view(1, 1, 1, H, 1)
x += 1000.0 * (1 + torch.arange(D, device="cuda").float().view(1, 1, D, 1, 1))
xc = cl(x)
for kw in (0, 1, 2):
    w = torch.zeros(16, 16, 3, 3, 3, device="cuda")
    w[:, :, 1, 1, kw] = torch.eye(16, device="cuda")
    y = ops.conv3d_16(xc, w)
    ref = torch.nn.functional.conv3d(xc, w, padding=1)
    for (d, h) in ((1, 3), (2, 0), (0, 7), (3, 4)):
        print("kw", kw, "d,h", d, h, "y:", [int(v) for v in y[0, 0, d, h, :6].tolist()], [int(v) for v in y[0, 0, d, h, 13:19].tolist()], " ref:", [int(v) for v in ref[0, 0, d, h, :6].tolist()], [int(v) for v in ref[0, 0, d, h, 13:19].tolist()])
