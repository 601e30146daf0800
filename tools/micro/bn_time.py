import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from movedepth_amd import ops
cl = lambda t: t.contiguous(memory_format=torch.channels_last_3d)
x = cl(torch.randn(6, 16, 96, 48, 160, device="cuda")); res = cl(torch.randn_like(x)); gy = cl(torch.randn_like(x))
g, b = torch.ones(16, device="cuda", requires_grad=True), torch.zeros(16, device="cuda", requires_grad=True)
rm, rv = torch.zeros(16, device="cuda"), torch.ones(16, device="cuda")
names = ["md_bn_relu_stats", "md_bn_relu_apply", "md_bn_relu_bwd_reduce", "md_bn_relu_bwd_dx"]
def ev(fn, n=10, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
def fused():
    xr = x.detach().requires_grad_(True)
    y = ops.bn_relu_3d(xr, g, b, res, rm, rv); y.backward(gy)
def lib():
    xr = x.detach().requires_grad_(True)
    y = torch.relu(torch.nn.functional.batch_norm(xr, rm, rv, g, b, True, 0.1, 1e-5)) + res; y.backward(gy)
print("fused fwd+bwd %.0f us   torch ops fwd+bwd %.0f us" % (ev(fused), ev(lib)))
ops.enable_kernel_timing(names)
for _ in range(5): fused()
torch.cuda.synchronize()
for k, v in ops.kernel_times_us().items(): print("  %-24s %.0f us" % (k, v["avg_us"]))
