"""GPU: the bench step with every torch BatchNorm forced onto torch's native kernels (what SyncBatchNorm is built from),
for --hip_bn_relu 1 (two fused layers) and 2 (every supported reg3d layer): the per-rank cost of synchronised statistics."""
import os, sys, runpy
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
_bn = torch.nn.functional.batch_norm
def _bn_native(*a, **k):
    with torch.backends.cudnn.flags(enabled=False):
        return _bn(*a, **k)
if os.environ.get("MD_BN_NATIVE", "1") == "1":
    torch.nn.functional.batch_norm = _bn_native
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [os.path.join(root, "bench.py"), "--steps", "30", "--warmup", "15", "--no_cpu_baseline"] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
