"""md_conv3d_c1_fwd at config 2's volume (6 x 16 x 96 x 48 x 160) on cold inputs (four volumes in rotation), timed by the dispatch
events inside the library; the kernel variant comes from the environment (MD_CONV3D_C1_GLDS, MD_CONV3D_C1_DS: one process each).
Also checks the result against the library convolution."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from movedepth_amd import ops

B, C, D, H, W = 6, 16, 96, 48, 160
torch.manual_seed(0)
xs = [torch.randn(B, C, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d) for _ in range(4)]
w = (torch.randn(1, C, 3, 3, 3, device="cuda") * 0.1).contiguous(memory_format=torch.channels_last_3d)
with torch.no_grad():
    y = ops.conv3d_c1(xs[0], w)
    ref = torch.nn.functional.conv3d(xs[0], w, padding=1)
    err = float((y - ref).abs().max() / ref.abs().max())
    for _ in range(8):
        for x in xs:
            ops.conv3d_c1(x, w)
    torch.cuda.synchronize()
    ops.enable_library_kernel_timing(True)
    for _ in range(10):
        for x in xs:
            ops.conv3d_c1(x, w)
    torch.cuda.synchronize()
    t = ops.library_kernel_times_us(["md_conv3d_c1_fwd"])["md_conv3d_c1_fwd"]
ops.enable_library_kernel_timing(False)
us = sorted(t["all_us"])
print("GLDS=%s DS=%s: avg %.1f us  median %.1f  min %.1f  (%.0f GB/s algorithmic)  max rel err vs library %.2e" % (
    os.environ.get("MD_CONV3D_C1_GLDS", "0"), os.environ.get("MD_CONV3D_C1_DS", "-"), t["avg_us"], us[len(us) // 2], us[0],
    (B * C * D * H * W + B * D * H * W) * 4 / t["avg_us"] * 1e-3, err))
