"""Diagnostics: how the oracle's OpenMP regions scale on this host (run on the GPU box): affinity mask, cgroup quota, and the
time of each hot-path op at 1..N threads."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
rng = np.random.default_rng(0)
H, W, D, C, G = 192, 640, 96, 32, 16
h, w = H // 4, W // 4
f32 = np.float32
ref, src = rng.standard_normal((1, C, h, w)).astype(f32), rng.standard_normal((1, C, h, w)).astype(f32)
Kq = np.array([[0.58 * w, 0, 0.5 * w, 0], [0, 1.92 * h, 0.5 * h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], f32)[None]
K0 = np.array([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]], f32)[None]
iKq, iK0 = np.linalg.pinv(Kq[0]).astype(f32)[None], np.linalg.pinv(K0[0]).astype(f32)[None]
T = oracle.transformation_from_parameters(np.array([[0.0, 0.01, 0.0]], f32), np.array([[0.05, 0.0, 0.03]], f32))
prior = (2 + 20 * rng.random((1, 1, h, w))).astype(f32)
img, tgt = rng.random((1, 3, H, W), dtype=f32), rng.random((1, 3, H, W), dtype=f32)
depth = (2 + 20 * rng.random((1, 1, H, W))).astype(f32)
gvol = rng.standard_normal((1, D, G, h, w)).astype(f32)
gl = rng.standard_normal((1, 1, H, W)).astype(f32)
hyp = oracle.schedule_depth_range(prior, D, 0.3, None, "inverse")


def tm(f, n=3):
    f()
    t = time.time()
    for _ in range(n):
        f()
    return (time.time() - t) / n


ops = {"cv_fwd": lambda: oracle.costvol_grouped(ref, src, Kq, iKq, hyp, T, G),
       "cv_bwd": lambda: oracle.costvol_grouped_bwd(gvol, ref, src, Kq, iKq, hyp, T),
       "warp": lambda: oracle.warp(img, depth, K0, iK0, T),
       "reproj": lambda: oracle.reproj_loss(img, tgt),
       "reproj_bwd": lambda: oracle.reproj_loss_bwd(gl, img, tgt),
       "warp_bwd": lambda: oracle.warp_bwd(img, img, depth, K0, iK0, T),
       "smooth": lambda: oracle.smooth_loss(depth, img, True)}
nmax = len(os.sched_getaffinity(0))
nt = 1
while True:
    oracle.set_num_threads(nt)
    print(nt, {k: "%.1f" % (1e3 * tm(f)) for k, f in ops.items()}, flush=True)
    if nt >= nmax:
        break
    nt = min(nt * 2, nmax)
