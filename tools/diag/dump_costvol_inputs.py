"""Diagnostics: run the trainer of `bench.py --trainer_args=...` for a few steps and save the geometric inputs of one plane-sweep
call (K, invK, pose, prior, z_trans, the library's pose pre-pass flags) to gpurun_out/ for inspection off the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from movedepth_amd import miopen_setup
miopen_setup.use_shipped_cache(0)
from movedepth_amd import ops
from movedepth_amd.options import MovedepthOptions
from movedepth_amd.synthetic import make_inputs
from movedepth_amd.trainer import Trainer

steps = int(os.environ.get("STEPS", "25"))
argv = ["--height", "192", "--width", "640", "--num_depth_bins", "96", "--batch_size", "6", "--res_arch", "18", "--prior_scale", "2",
        "--convex_up", "--weights_init", "scratch", "--learning_rate", "2e-4", "--local_rank", "0"] + sys.argv[1:]
opt = MovedepthOptions().parse(argv)
torch.manual_seed(1234); np.random.seed(1234)
t = Trainer(opt); t.set_train()
inputs = make_inputs(opt.batch_size, opt.height, opt.width, opt.frame_ids, seed=0, device=t.device)
dump = {}
orig = ops.costvol_grouped
def spy(ref, src, K, invK, pose, G, depth_priors=None, prior=None, ndepth=None, scale_fac=0.3, z_trans=None, type="inverse", layout="bgd"):
    if dump.get("armed") and "K" not in dump:
        dump.update(K=K.detach().float().cpu().numpy(), invK=invK.detach().float().cpu().numpy(), pose=pose.detach().float().cpu().numpy(),
                    prior=None if prior is None else prior.detach().float().cpu().numpy(),
                    z_trans=None if z_trans is None else z_trans.detach().float().cpu().numpy(), ndepth=ndepth, scale_fac=scale_fac,
                    type=type, shape=tuple(ref.shape), G=G)
    return orig(ref, src, K, invK, pose, G, depth_priors=depth_priors, prior=prior, ndepth=ndepth, scale_fac=scale_fac, z_trans=z_trans,
                type=type, layout=layout)
ops.costvol_grouped = spy
import movedepth_amd.trainer as tr
if hasattr(tr, "ops"): tr.ops.costvol_grouped = spy
for i in range(steps):
    dump["armed"] = i == steps - 1
    t.train_step(dict(inputs))
torch.cuda.synchronize()
out = os.path.join(ROOT, "gpurun_out", "costvol_inputs.npz")
np.savez(out, **{k: v for k, v in dump.items() if isinstance(v, np.ndarray)}, meta=np.array([dump.get("ndepth") or 0, dump.get("scale_fac") or 0.0, dump.get("G") or 0] + list(dump.get("shape", ()))))
print("saved", out, {k: (v.shape if isinstance(v, np.ndarray) else v) for k, v in dump.items()})
