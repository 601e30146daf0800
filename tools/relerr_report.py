import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from conftest import load_golden, relerr
from movedepth_amd import ops
def dev(a, g=False): return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda().requires_grad_(g)
for tag in ["small","white","oob","zv2","c64g8"]:
    g = load_golden("costvol_"+tag); G=int(g["G"])
    r,s = dev(g["ref"],True), dev(g["src0"],True)
    vol = ops.costvol_grouped(r, s, dev(g["K"]), dev(g["invK"]), dev(g["pose"][:,0]), G, depth_priors=dev(g["hyp"]))
    (vol*dev(g["grad_out"])).sum().backward()
    print(tag, "vol %.2e d_ref %.2e d_src %.2e maxabs %.2e" % (relerr(vol.detach().cpu().numpy(), g["grouped0"]), relerr(r.grad.cpu().numpy(), g["d_ref"]), relerr(s.grad.cpu().numpy(), g["d_src0"]), np.abs(vol.detach().cpu().numpy()-g["grouped0"]).max()))
