import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from movedepth_amd import ops
torch.manual_seed(0)
B, D, H, W = 1, 5, 8, 40
cl = lambda t: t.contiguous(memory_format=torch.channels_last_3d)
x = cl(torch.randn(B, 16, D, H, W, device="cuda"))
for name, (kd, kh, kw) in {"centre": (1, 1, 1), "kw0": (1, 1, 0), "kw2": (1, 1, 2), "kh0": (1, 0, 1), "kh2": (1, 2, 1), "kd0": (0, 1, 1), "kd2": (2, 1, 1)}.items():
    for mode in ("eye", "rand"):
        w = torch.zeros(16, 16, 3, 3, 3, device="cuda")
        w[:, :, kd, kh, kw] = torch.eye(16, device="cuda") if mode == "eye" else torch.randn(16, 16, device="cuda")
        y = ops.conv3d_16(x, w)
        ref = torch.nn.functional.conv3d(x, w, padding=1)
        err = (y - ref).abs().max().item()
        rel = ((y - ref).norm() / ref.norm()).item()
        bad = (y - ref).abs().amax(dim=(0, 1, 2, 3))   # per column
        print("%-7s %-4s max err %.3e rel %.3e  bad columns: %s" % (name, mode, err, rel, [i for i, v in enumerate(bad.tolist()) if v > 1e-3][:12]))
w = torch.randn(16, 16, 3, 3, 3, device="cuda") * 0.1
y = ops.conv3d_16(x, w); ref = torch.nn.functional.conv3d(x, w, padding=1)
print("full random: rel %.3e" % ((y - ref).norm() / ref.norm()).item())
