"""GPU: where does the conv time go?  reg3d / encoders fwd+bwd under different library settings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from movedepth_amd import networks

def timeit(fn, n=3, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3

mode = sys.argv[1] if len(sys.argv) > 1 else "default"
torch.backends.cudnn.benchmark = mode in ("benchmark", "benchmark_cl")
dev = "cuda"
B, D, h, w = 6, int(os.environ.get("D", 96)), 48, 160
reg = networks.reg3d(16, 16, 3).to(dev)
x = torch.randn(B, 16, D, h, w, device=dev)
if mode.endswith("cl"):
    reg = reg.to(memory_format=torch.channels_last_3d)
    x = x.contiguous(memory_format=torch.channels_last_3d)
xin = x.permute(0, 2, 1, 3, 4).requires_grad_(True)
def f_reg():
    y = reg(xin); y.sum().backward()
t0 = time.time(); f_reg(); torch.cuda.synchronize(); print(mode, "reg3d first call %.1f s" % (time.time() - t0))
print(mode, "reg3d fwd+bwd ms", timeit(f_reg))
enc = networks.ResnetEncoder(18).to(dev); dec = networks.DepthDecoder(enc.num_ch_enc).to(dev)
img = torch.rand(B, 3, 192, 640, device=dev)
def f_enc():
    o = dec(enc(img)); sum(v.sum() for v in o.values()).backward()
f_enc(); print(mode, "resnet18+decoder fwd+bwd ms", timeit(f_enc))
fpn = networks.FPN4(8, 2).to(dev)
def f_fpn():
    a, b_ = fpn(img); (a.sum() + b_.sum()).backward()
f_fpn(); print(mode, "fpn4 fwd+bwd ms", timeit(f_fpn))
