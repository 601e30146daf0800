"""Host-side cost per BatchNorm call (tiny tensor: the GPU work is negligible): ops.sync_batch_norm against F.batch_norm."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from movedepth_amd import ops

x = torch.randn(2, 64, 6, 10, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
w, b = torch.ones(64, device="cuda", requires_grad=True), torch.zeros(64, device="cuda", requires_grad=True)
rm, rv = torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")
g = torch.randn_like(x)


def run(f, n=2000):
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    host = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    return host * 1e6, (time.perf_counter() - t) / n * 1e6


def hip():
    y = ops.sync_batch_norm(x, w, b, rm, rv, 0.1, 1e-5)
    y.backward(g)


def lib():
    y = torch.nn.functional.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5)
    y.backward(g)


def hip_fwd():
    with torch.no_grad():
        ops.sync_batch_norm(x, w, b, rm, rv, 0.1, 1e-5)


def lib_fwd():
    with torch.no_grad():
        torch.nn.functional.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5)


for name, f in (("hip fwd+bwd", hip), ("torch fwd+bwd", lib), ("hip fwd", hip_fwd), ("torch fwd", lib_fwd)):
    h, t = run(f)
    print("%-14s host %.1f us per call, wall %.1f us" % (name, h, t))
