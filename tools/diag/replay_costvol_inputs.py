"""Diagnostics: replay the geometric inputs saved by dump_costvol_inputs.py (tests/golden-style .npz) through md_costvol_fwd / _bwd
with random features and print the library's per-kernel times (channels-last backward / scatter backward)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from movedepth_amd import ops

d = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "costvol_inputs.npz"))
meta = d["meta"]; D, sf, G = int(meta[0]), float(meta[1]), int(meta[2]); B, C, h, w = (int(v) for v in meta[3:7])
dt = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32
dev = lambda a: torch.from_numpy(a).cuda()
pose = d["pose"].copy()
if os.environ.get("POSE_TZ"):   # what-if: another translation along the optical axis
    pose[:, 2, 3] = float(os.environ["POSE_TZ"])
ref = torch.randn(B, C, h, w, device="cuda").to(dt).requires_grad_(True)
src = torch.randn(B, C, h, w, device="cuda").to(dt).requires_grad_(True)
ops.enable_library_kernel_timing(True)
for _ in range(6):
    ref.grad = src.grad = None
    vol = ops.costvol_grouped(ref, src, dev(d["K"]), dev(d["invK"]), dev(pose), G, prior=dev(d["prior"]), ndepth=D, scale_fac=sf, layout="ndhwc")
    vol.backward(torch.randn_like(vol))
torch.cuda.synchronize()
sfx = "_bf16" if dt == torch.bfloat16 else ""
for k, v in ops.library_kernel_times_us(["md_costvol_fwd" + sfx, "md_costvol_bwd" + sfx, "md_costvol_bwd_wild" + sfx]).items():
    print("%-28s avg %8.1f us  min %8.1f" % (k, v["avg_us"], v["min_us"]))
print("volume nonzero fraction %.4f" % float((vol != 0).float().mean()))
