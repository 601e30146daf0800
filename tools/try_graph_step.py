"""GPU experiment: capture one training step (forward, backward, Adam) in a HIP graph and replay it.
Reports whether capture works and the replayed step time against the eager one.  Not used by the product."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from movedepth_amd.options import MovedepthOptions
from movedepth_amd.synthetic import make_inputs
from movedepth_amd.trainer import Trainer

argv = ["--height", "192", "--width", "640", "--num_depth_bins", "96", "--batch_size", "6", "--res_arch", "18", "--prior_scale", "2",
        "--convex_up", "--weights_init", "scratch", "--learning_rate", "2e-4"] + sys.argv[1:]
opt = MovedepthOptions().parse(argv)
torch.manual_seed(0); np.random.seed(0)
t = Trainer(opt); t.set_train()
# capturable optimizer state (step counters on the device)
for g in t.model_optimizer.param_groups:
    g["capturable"] = True
inputs = make_inputs(6, 192, 640, opt.frame_ids, seed=0, device=t.device)


def eager(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        t.train_step(dict(inputs))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(8):
        t.train_step(dict(inputs))
torch.cuda.current_stream().wait_stream(s)
print("eager ms/step: %.2f" % eager(20), flush=True)
g = torch.cuda.CUDAGraph()
try:
    t.model_optimizer.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        outputs, losses = t.train_step(dict(inputs))
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print("graph replay ms/step: %.2f   loss %.5f" % ((time.perf_counter() - t0) / 20 * 1e3, float(losses["loss"])))
except Exception as e:  # noqa: BLE001
    import traceback
    traceback.print_exc()
    print("capture failed:", type(e).__name__, str(e)[:400])
