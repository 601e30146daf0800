"""GPU: which torch ops launch layout/copy kernels inside one training step (torch.profiler, op -> shapes -> CUDA time)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
from movedepth_amd.options import MovedepthOptions
from movedepth_amd.synthetic import make_inputs
from movedepth_amd.trainer import Trainer
from movedepth_amd import miopen_setup

miopen_setup.use_shipped_cache(0)

argv = ["--height", "192", "--width", "640", "--num_depth_bins", "96", "--batch_size", "6", "--res_arch", "18", "--prior_scale", "2",
        "--convex_up", "--weights_init", "scratch", "--learning_rate", "2e-4"] + sys.argv[1:]
if "--miopen_find" not in argv and miopen_setup.find_db_hits(0):
    argv += ["--miopen_find", "2"]
opt = MovedepthOptions().parse(argv)
torch.manual_seed(0); np.random.seed(0)
t = Trainer(opt); t.set_train()
inputs = make_inputs(6, 192, 640, opt.frame_ids, seed=0, device=t.device)
for _ in range(3):
    t.train_step(dict(inputs))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    t.train_step(dict(inputs))
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
only = os.environ.get("OPS")  # comma-separated aten op names; default: every op, by self GPU time
for e in prof.key_averages(group_by_input_shape=True):
    if only and e.key not in only.split(","):
        continue
    us = e.self_device_time_total if hasattr(e, "self_device_time_total") else e.self_cuda_time_total
    if us <= 0:
        continue
    k = (e.key, str(e.input_shapes)[:100])
    agg[k][0] += e.count
    agg[k][1] += us
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print("total self GPU time of one step: %.2f ms over %d (op, shape) groups" % (tot / 1e3, len(rows)))
print("%-44s %-102s %6s %10s" % ("op", "input shapes", "calls", "gpu us"))
for (k, shp), (c, us) in rows[:int(os.environ.get("TOP", "60"))]:
    print("%-44s %-102s %6d %10.0f" % (k[:44], shp, c, us))

byop = collections.defaultdict(lambda: [0, 0.0])
for (k, shp), (c, us) in rows:
    byop[k][0] += c
    byop[k][1] += us
print("\n---- by op")
for k, (c, us) in sorted(byop.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-60s %6d calls %10.0f us" % (k[:60], c, us))
