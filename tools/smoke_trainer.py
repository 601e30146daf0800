"""GPU: one small training step end to end (config 1 shape) -- used while developing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from movedepth_amd.options import MovedepthOptions
from movedepth_amd.synthetic import make_inputs
from movedepth_amd.trainer import Trainer
H, W, D, B = int(os.environ.get("H", 64)), int(os.environ.get("W", 128)), int(os.environ.get("D", 16)), int(os.environ.get("B", 1))
opt = MovedepthOptions().parse(["--height", str(H), "--width", str(W), "--num_depth_bins", str(D), "--batch_size", str(B),
                                "--convex_up", "--weights_init", "scratch"])
torch.manual_seed(0)
t = Trainer(opt); t.set_train()
inp = make_inputs(B, H, W, opt.frame_ids, seed=0, device=t.device)
for i in range(3):
    t0 = time.time()
    out, losses = t.train_step(dict(inp))
    torch.cuda.synchronize()
    print("step", i, "loss %.5f" % float(losses["loss"].detach()), {k: round(float(v.detach()), 5) for k, v in losses.items() if k != "loss"}, "%.3fs" % (time.time() - t0))
assert all(torch.isfinite(p.grad).all() for m in t.models.values() for p in m.parameters() if p.grad is not None)
print("keys", sorted(str(k) for k in out.keys())[:12], "...")
