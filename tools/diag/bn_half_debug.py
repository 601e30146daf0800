import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from movedepth_amd import ops
torch.manual_seed(5)
for dtype in (torch.bfloat16, torch.float16):
    shape = (3, 64, 24, 40)
    C = 64
    x = (torch.randn(*shape, device="cuda") * 1.5 + 0.3).to(dtype).contiguous(memory_format=torch.channels_last)
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.2
    y = ops.sync_batch_norm(x, gamma, beta, None, None, 0.1, 1e-5)
    yb = torch.nn.functional.batch_norm(x.double(), None, None, gamma.double(), beta.double(), True, 0.1, 1e-5)
    yr = yb.to(dtype)
    d = (y.double() - yb).abs()
    print(dtype, "max abs err", float(d.max()), "mismatches vs rounded reference", int((y != yr).sum()), "of", y.numel())
    e = (d / yb.abs().clamp_min(1e-3))
    idx = torch.nonzero(e > 2.0 ** -10)
    print("  n rel>2^-10:", idx.shape[0], "first:", idx[:5].tolist())
    for i in idx[:5].tolist():
        print("   ", float(y[tuple(i)]), float(yb[tuple(i)]))
