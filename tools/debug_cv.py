import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
from movedepth_amd import ops
from test_hip_parity import smooth_field, kitti_K, rand_pose, dev, host
oracle.build()
rng = np.random.default_rng(29)
B, C, G, h, w, D = 1, 32, 16, 8, 16, int(os.environ.get("DD", "2"))
ref = smooth_field(rng, (B, C, h, w), 2, -1, 1); src = smooth_field(rng, (B, C, h, w), 2, -1, 1)
K, invK = kitti_K(h, w, B)
prior = (2 + 20 * rng.random((B, 1, h, w))).astype(np.float32)
pose = rand_pose(oracle, rng, B, 0.02, 0.1)
z = 30.0 * pose[:, 2, 3]
hyp = oracle.schedule_depth_range(prior, D, 0.3, z, "inverse")
exp = oracle.costvol_grouped(ref, src, K, invK, hyp, pose, G)
for fused in (True, False):
    kw = dict(prior=dev(prior), ndepth=D, scale_fac=0.3, z_trans=dev(z), type="inverse") if fused else dict(depth_priors=dev(hyp))
    vol = host(ops.costvol_grouped(dev(ref), dev(src), dev(K), dev(invK), dev(pose), G, layout="ndhwc", **kw))
    err = np.abs(vol - exp)
    print("fused", fused, "max err", err.max(), "rel", np.linalg.norm(vol - exp) / np.linalg.norm(exp))
    for d in range(D):
        e = err[0, d].max(0)  # h,w
        bad = np.argwhere(e > 1e-4)
        print(" d", d, "max", err[0, d].max(), "bad pixels", len(bad), bad[:12].tolist())
    eg = err[0].max(axis=(0, 2, 3)); print(" per group max err", np.round(eg, 4))
