#!/usr/bin/env python3
"""Out-of-bounds probe for the plane-sweep entry points: every tensor argument in turn is placed at the very END of its own 12 MB
allocation (its own allocator segment: what lies behind is usually unmapped), the forward and the backward are run, and the process
dies with a GPU memory fault if a kernel reads or writes past that tensor.  One (case, argument) per subprocess.
usage: python tools/diag/oob_probe.py            (GPU)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [dict(B=2, C=32, G=16, h=24, w=40, D=12), dict(B=1, C=32, G=16, h=48, w=160, D=32), dict(B=2, C=32, G=16, h=16, w=64, D=8, rot=0.3, trans=2.0),
         dict(B=1, C=64, G=16, h=12, w=20, D=5), dict(B=6, C=32, G=16, h=48, w=160, D=96)]
ARGS = ["ref", "src", "K", "invK", "pose", "prior", "z", "gout", "none"]


def at_end(t):
    import torch
    n = t.numel() * t.element_size()
    big = torch.empty(12 * 1024 * 1024, dtype=torch.uint8, device="cuda")
    tail = big[big.numel() - n:].view(t.dtype).view(t.shape if t.is_contiguous() else -1)
    if not t.is_contiguous():   # channels-last: keep the strides
        tail = tail.as_strided(t.shape, t.stride())
    tail.copy_(t)
    return tail


def one(ci, arg, dtype):
    import torch
    sys.path.insert(0, ROOT)
    from movedepth_amd import ops
    c = CASES[ci]
    B, C, G, h, w, D = (c[k] for k in "BCGhwD")
    dt = {"f32": torch.float32, "f16": torch.float16}[dtype]
    g = torch.Generator(device="cuda").manual_seed(ci)
    mk = lambda *s: torch.randn(*s, device="cuda", generator=g)
    t = dict(ref=mk(B, C, h, w).to(dt).contiguous(memory_format=torch.channels_last), src=mk(B, C, h, w).to(dt).contiguous(memory_format=torch.channels_last),
             prior=2 + 20 * torch.rand(B, 1, h, w, device="cuda", generator=g))
    K = torch.tensor([[0.58 * w, 0, 0.5 * w, 0], [0, 1.92 * h, 0.5 * h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device="cuda").repeat(B, 1, 1)
    t["K"], t["invK"] = K, torch.linalg.pinv(K)
    from movedepth_amd.layers import transformation_from_parameters
    t["pose"] = transformation_from_parameters(mk(B, 1, 3) * c.get("rot", 0.01), mk(B, 1, 3) * c.get("trans", 0.05)).contiguous()
    t["z"] = (30.0 * t["pose"][:, 2, 3]).contiguous()
    if arg in t:
        t[arg] = at_end(t[arg])
    r, s = t["ref"].requires_grad_(True), t["src"].requires_grad_(True)
    vol = ops.costvol_grouped(r, s, t["K"], t["invK"], t["pose"], G, prior=t["prior"], ndepth=D, scale_fac=0.3, z_trans=t["z"], layout="ndhwc")
    torch.cuda.synchronize()
    gout = torch.randn(vol.shape, device="cuda", generator=g).to(dt)
    if arg == "gout":
        gout = at_end(gout.permute(0, 1, 3, 4, 2).contiguous()).permute(0, 1, 4, 2, 3)
    for force in (False, True):     # library partition, then a cost-balanced one
        ops.backward_policy().force_balance = force
        r.grad = s.grad = None
        vol.backward(gout, retain_graph=True)
        torch.cuda.synchronize()
    print("ok")


if __name__ == "__main__":
    if len(sys.argv) == 4:
        one(int(sys.argv[1]), sys.argv[2], sys.argv[3])
        sys.exit(0)
    bad = 0
    for dtype in ("f32", "f16"):
        for ci in range(len(CASES)):
            for arg in ARGS:
                p = subprocess.run([sys.executable, __file__, str(ci), arg, dtype], capture_output=True, text=True)
                ok = p.returncode == 0 and "ok" in p.stdout
                if not ok:
                    bad += 1
                    tail = [l for l in (p.stderr + p.stdout).splitlines() if "fault" in l.lower() or "Error" in l][-2:]
                    print("FAULT %s case %d (%s) with `%s` at the end of its allocation: rc %d %s" % (dtype, ci, CASES[ci], arg, p.returncode, tail))
    print("oob probe: %d failures" % bad)
    sys.exit(1 if bad else 0)
