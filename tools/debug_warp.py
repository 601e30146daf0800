import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle
from movedepth_amd import ops
from test_hip_parity import smooth_field, kitti_K, rand_pose, dev, host
rng = np.random.default_rng(3)
B, H, W = 2, 192, 640
img = smooth_field(rng, (B, 3, H, W), 8)
depth = (2 + 20 * smooth_field(rng, (B, 1, H, W), 16)).astype(np.float32)
K, invK = kitti_K(H, W, B)
T = rand_pose(oracle, rng, B, 0.01, 0.1)
gout = smooth_field(rng, (B, 3, H, W), 8, 0.2, 1.0)
exp, exp_pix = oracle.warp(img, depth, K, invK, T)
exp_dd, exp_dT = oracle.warp_bwd(gout, img, depth, K, invK, T)
d, t = dev(depth, True), dev(T, True)
out, pix, _ = ops.warp_border(dev(img), d, dev(K), dev(invK), t, want_pix=True)
(out * dev(gout)).sum().backward()
dd = host(d.grad).reshape(exp_dd.shape)
err = np.abs(dd - exp_dd)
scale = np.abs(exp_dd).max()
bad = err > 1e-3 * scale
print("d_depth scale", scale, "outliers", bad.sum(), "of", bad.size, "max err", err.max())
ys = np.argwhere(bad)[:10]
ix = (exp_pix[..., 0] + 1) / 2 * (W - 1); iy = (exp_pix[..., 1] + 1) / 2 * (H - 1)
for b, y, x in ys:
    print("  at", b, y, x, "gpu", dd[b, y, x], "cpu", exp_dd[b, y, x], "ix,iy", ix[b, y, x], iy[b, y, x])
print("d_T gpu\n", host(t.grad)[0], "\ncpu\n", exp_dT[0])
# float64 reference of dT from oracle d-terms: accumulate GPU-side d_depth? compare sums in float64 using torch on cpu
