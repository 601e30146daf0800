"""Debug: how far do the parameter gradients of one training step move when one weight tensor is perturbed by 1e-7 relative?
(per-pixel decisions -- localmax's arg-max, min over frames -- make the step discontinuous: this measures by how much)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from movedepth_amd.options import MovedepthOptions
from movedepth_amd.synthetic import make_inputs
from movedepth_amd.trainer import Trainer
ARGV = ["--height", "64", "--width", "128", "--num_depth_bins", "16", "--convex_up", "--weights_init", "scratch",
        "--miopen_find", "0", "--automask_noise", "host", "--grad_bucket_mb", "8", "--learning_rate", "1e-3", "--disable_automasking"]


def run(perturb, peak, extra):
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
    opt = MovedepthOptions().parse(ARGV + ["--batch_size", "4"] + extra)
    torch.manual_seed(50); np.random.seed(50)
    t = Trainer(opt)
    with torch.no_grad():
        if peak:
            t.models["reg3d"].prob.weight.mul_(peak)
        if perturb:
            t.models["mvs_encoder"].conv0[0].conv.weight.mul_(1.0 + perturb)
    t.set_train()
    shards = [make_inputs(2, 64, 128, opt.frame_ids, seed=200 + r, device=t.device) for r in range(2)]
    batch = {k: torch.cat([s[k] for s in shards], 0) for k in shards[0]}
    torch.manual_seed(300); np.random.seed(300)
    t.train_step(batch)
    torch.cuda.synchronize()
    return {mn + "." + pn: p.grad.detach().double().cpu().numpy() for mn, m in t.models.items() for pn, p in m.named_parameters() if p.grad is not None}


for extra in ([], ["--force_sync_bn", "1"]):
    for peak in (0.0, 40.0):
        a, b = run(0.0, peak, extra), run(1e-7, peak, extra)
        per = {}
        for n in a:
            per.setdefault(n.split(".")[0], []).append(n)
        out = {}
        for m, ns in per.items():
            va, vb = np.concatenate([a[n].ravel() for n in ns]), np.concatenate([b[n].ravel() for n in ns])
            out[m] = "%.1e" % (np.linalg.norm(va - vb) / np.linalg.norm(va))
        print("extra", extra, "peak", peak, out, flush=True)
