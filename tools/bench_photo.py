#!/usr/bin/env python3
"""Micro-benchmark of the fused photometric kernels at BASELINE config 2 (B=6, 192x640, 2 source frames): the mono group
(4 scales, disparity pyramid, auto-mask), the MVS group and the fused-depth group.  Wall time per call from torch events
(includes the Python wrapper); run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations.  GPU only.
WANT_PIX=0: the mono group without its sample-grid outputs (what the trainer asks for with --lazy_sample_grids 1)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from movedepth_amd import ops  # noqa: E402
from movedepth_amd.layers import transformation_from_parameters  # noqa: E402
from movedepth_amd.synthetic import make_inputs  # noqa: E402


def timeit(fn, iters, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=6)
    ap.add_argument("--H", type=int, default=192)
    ap.add_argument("--W", type=int, default=640)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--unfused", type=int, default=1, help="also time the per-op kernels of round 2 on the same inputs")
    a = ap.parse_args()
    B, H, W = a.B, a.H, a.W
    torch.manual_seed(0)
    inp = make_inputs(B, H, W, [0, -1, 1], seed=0, device="cuda")
    planar = [inp[("color", 0, 0)], inp[("color", -1, 0)], inp[("color", 1, 0)]]
    packed = ops.pack_rgbx(planar)                    # what the trainer does once per step
    target, srcs = packed[0], packed[1:]
    ptarget, psrcs = planar[0], planar[1:]
    K, invK = inp[("K", 0)], inp[("inv_K", 0)]
    Ts = [transformation_from_parameters(0.01 * torch.randn(B, 1, 3, device="cuda"), 0.05 * torch.randn(B, 1, 3, device="cuda"),
                                         invert=(i == 0)).requires_grad_(True) for i in range(2)]
    disps = [(0.02 + 0.3 * torch.rand(B, 1, H >> s, W >> s, device="cuda")).requires_grad_(True) for s in range(4)]
    depth = (2 + 20 * torch.rand(B, H, W, device="cuda")).requires_grad_(True)
    noise = torch.randn(4, B, 1, H, W, device="cuda") * 1e-5
    img_mb = B * 3 * H * W * 4 / 1e6

    def mono(bwd):
        ident = ops.identity_loss(target, srcs)
        out = ops.photometric_loss(target, srcs, Ts, K, invK, disps, is_disp=True, ident_min=ident, noise=noise, want_pix=bool(int(os.environ.get("WANT_PIX", "1"))))
        if bwd:
            sum(out["loss"]).backward()

    def mvs(bwd):
        out = ops.photometric_loss(target, srcs, [t.detach() for t in Ts], K, invK, [depth], mvs_mode=True, want_oob=True, want_mask=True)
        if bwd:
            out["loss"][0].backward()

    def unfused_mono(bwd):
        ident = torch.cat([ops.reprojection_loss(s, ptarget) for s in psrcs], 1)
        tot = 0
        for s in range(4):
            d = ops.disp_to_depth_up(disps[s], H, W, 0.1, 100.0)
            rl = torch.cat([ops.reprojection_loss(ops.warp_border(psrcs[f], d, K, invK, Ts[f], want_pix=True)[0], ptarget) for f in range(2)], 1)
            tot = tot + ops.masked_min_loss(rl, ident, noise[s])[0]
        if bwd:
            tot.backward()

    print("B=%d %dx%d, one image = %.2f MB" % (B, H, W, img_mb))
    for name, fn in (("mono group, fused (identity + 4 scales x 2 frames)", mono), ("MVS group, fused (2 frames)", mvs)) + \
            ((("mono group, per-op kernels of round 2", unfused_mono),) if a.unfused else ()):
        f = timeit(lambda: fn(False), a.iters)
        fb = timeit(lambda: fn(True), a.iters)
        print("  %-52s fwd %7.1f us   fwd+bwd %7.1f us (wall, incl. Python)" % (name, f, fb))
    # the kernels' own durations (HIP events inside the library around each launch), per training step's worth of calls: identity +
    # mono group + MVS group, forward and backward
    ops.enable_library_kernel_timing(True)
    n = 20
    for _ in range(n):
        mono(True)
        mvs(True)
    torch.cuda.synchronize()
    t = ops.library_kernel_times_us(["md_photo_fwd", "md_photo_bwd"])
    ops.enable_library_kernel_timing(False)
    for k, v in t.items():
        print("  kernel only %-14s %7.1f us per step's worth of calls (%d dispatches each, avg %.1f us)"
              % (k, v["avg_us"] * v["launches"] / n, v["launches"] // n, v["avg_us"]))


if __name__ == "__main__":
    main()
