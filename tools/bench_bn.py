#!/usr/bin/env python3
"""GPU time of training-mode BatchNorm forward + backward on the model's layer shapes: csrc/syncbn.hip (ops.sync_batch_norm) against
F.batch_norm on the library (MIOpen) and on torch's native kernels (what torch.nn.SyncBatchNorm is built from)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from movedepth_amd import ops  # noqa: E402

SHAPES = [(6, 64, 96, 320), (6, 64, 48, 160), (6, 128, 24, 80), (6, 256, 12, 40), (6, 512, 6, 20), (6, 8, 192, 640), (6, 16, 96, 320),
          (6, 32, 48, 160), (6, 8, 48, 160), (6, 16, 96, 48, 160), (6, 32, 48, 24, 80), (6, 64, 24, 12, 40)]


def time_fn(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


for shape in SHAPES:
    fmt = torch.channels_last if len(shape) == 4 else torch.channels_last_3d
    C = shape[1]
    x = torch.randn(*shape, device="cuda").contiguous(memory_format=fmt).requires_grad_(True)
    g = torch.randn(*shape, device="cuda").contiguous(memory_format=fmt)
    w, b = torch.ones(C, device="cuda", requires_grad=True), torch.zeros(C, device="cuda", requires_grad=True)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    mb = x.numel() * 4 / 1e6
    iters = 200 if mb < 20 else 30

    def hip():
        ops.sync_batch_norm(x, w, b, rm, rv, 0.1, 1e-5).backward(g)

    def lib():
        torch.nn.functional.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5).backward(g)

    def native():
        with torch.backends.cudnn.flags(enabled=False):
            torch.nn.functional.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5).backward(g)

    ops.enable_library_kernel_timing(True)
    t_hip = time_fn(hip, iters)
    kt = ops.library_kernel_times_us(["md_bn_stats", "md_bn_apply", "md_bn_bwd_reduce", "md_bn_bwd_dx"])
    ops.enable_library_kernel_timing(False)
    t_lib, t_nat = time_fn(lib, iters), time_fn(native, iters)
    ks = " ".join("%s %.1f" % (k[6:], v["avg_us"]) for k, v in kt.items())
    print("%-24s %6.1f MB  hip %7.1f us (kernels: %s = %.1f; ideal 8 passes at 5 TB/s %.1f)  library %7.1f us  torch-native %7.1f us" % (
        shape, mb, t_hip, ks, sum(v["avg_us"] for v in kt.values()), 8 * mb / 5.0, t_lib, t_nat))
