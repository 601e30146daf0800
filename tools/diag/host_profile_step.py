"""Where the launching thread spends a training step: cProfile over K steps of bench.py's workload (no synchronisation inside the
window), top functions by own and cumulative time.  usage: python tools/diag/host_profile_step.py [trainer args ...]"""
import cProfile
import contextlib
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from movedepth_amd import miopen_setup

miopen_setup.use_shipped_cache(0)
from movedepth_amd.options import MovedepthOptions
from movedepth_amd.synthetic import make_inputs
from movedepth_amd.trainer import Trainer

argv = ["--height", "192", "--width", "640", "--num_depth_bins", "96", "--batch_size", "6", "--res_arch", "18", "--prior_scale", "2",
        "--convex_up", "--weights_init", "scratch", "--learning_rate", "2e-4", "--miopen_find", "2" if miopen_setup.find_db_hits(0) else "1"]
argv += sys.argv[1:]
opt = MovedepthOptions().parse(argv)
torch.manual_seed(1234)
np.random.seed(1234)
with contextlib.redirect_stdout(sys.stderr):
    t = Trainer(opt)
t.set_train()
inputs = make_inputs(opt.batch_size, opt.height, opt.width, opt.frame_ids, seed=0, device=t.device)
for _ in range(12):
    t.train_step(dict(inputs))
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    t.train_step(dict(inputs))
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print("un-profiled: host issue %.2f ms per step, wall %.2f ms per step" % (1e3 * th / K, 1e3 * tt / K))
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(K):
    t.train_step(dict(inputs))
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print("\n".join(l[:230] for l in s.getvalue().splitlines() if l.strip()))
