"""GPU diagnostic: replicate tests/test_hip_parity.py::test_reg3d_conv0_paths_agree[bgd] (same seeds), several
repeats, errors of every mode against an fp64 copy and pairwise, to tell a reproducible discrepancy from
run-to-run variation."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from movedepth_amd import networks

rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
layout = sys.argv[1] if len(sys.argv) > 1 else "bgd"
torch.manual_seed(4)
net = networks.reg3d(16, 16, 3).cuda().to(memory_format=torch.channels_last_3d)
B, D, G, h, w = 2, 16, 16, 24, 32
if layout == "bdg":
    vol = torch.randn(B, D, G, h, w, device="cuda")
elif layout == "bgd":
    vol = torch.randn(B, G, D, h, w, device="cuda").permute(0, 2, 1, 3, 4)
else:
    vol = torch.randn(B, D, h, w, G, device="cuda").permute(0, 1, 4, 2, 3)
net64 = copy.deepcopy(net).double().cpu()
net64.hip_conv0_wgrad = False; net64.hip_prob = False; net64.find_convs = False
v64 = vol.double().cpu().requires_grad_(True)
o64 = net64(v64); o64.square().mean().backward()
ref = (o64, v64.grad, net64.conv0.conv.weight.grad)
print("layout", layout, " |d_volume| fp64 norm %.3e  max %.3e" % (v64.grad.norm().item(), v64.grad.abs().max().item()))
for rep in range(3):
    outs = {}
    for mode, hip, lib_fd in (("HIP all", True, False), ("HIP wgrad", True, True), ("library", False, False)):
        net.hip_conv0_wgrad, net.lib_conv0_fwd_dgrad = hip, lib_fd
        net.zero_grad()
        v = vol.detach().requires_grad_(True)
        o = net(v); o.square().mean().backward()
        torch.cuda.synchronize()
        outs[mode] = (o.detach().cpu(), v.grad.detach().cpu(), net.conv0.conv.weight.grad.detach().cpu())
        print("  rep %d %-10s vs fp64: %s" % (rep, mode, "  ".join("%.2e" % rel(a, b) for a, b in zip(outs[mode], ref))))
    for m in ("HIP all", "HIP wgrad"):
        print("  rep %d %-10s vs library: %s" % (rep, m, "  ".join("%.2e" % rel(a, b) for a, b in zip(outs[m], outs["library"]))))

# ---- ReLU knife-edge check: activation masks after every ReLU, each fp32 mode against the fp64 copy
print("ReLU masks that differ from the fp64 network (module: HIP all / library), and the smallest |pre-activation| there:")
import torch.nn.functional as F
def masks(model, v, hip, lib_fd):
    model.hip_conv0_wgrad, model.lib_conv0_fwd_dgrad = hip, lib_fd
    rec = {}
    hooks = []
    for name, m in model.named_modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            hooks.append(m.register_forward_hook(lambda mod, i, o, name=name: rec.__setitem__(name, o.detach().double().cpu().clone())))
    with torch.no_grad():
        model.train(); model(v)
    for hk in hooks: hk.remove()
    return rec
m64 = masks(net64, vol.double().cpu(), False, False)
mh = masks(net, vol.detach(), True, False)
ml = masks(net, vol.detach(), False, False)
for name in m64:
    r = m64[name]
    dh = ((mh[name] > 0) != (r > 0)); dl = ((ml[name] > 0) != (r > 0))
    if dh.any() or dl.any():
        sm = lambda d: ("%.1e" % r[d].abs().min().item()) if d.any() else "-"
        print("  %-10s flips %d / %d of %d   |bn out| at flips: %s / %s   (rms of layer %.2e)" % (name, int(dh.sum()), int(dl.sum()), r.numel(), sm(dh), sm(dl), r.pow(2).mean().sqrt().item()))
