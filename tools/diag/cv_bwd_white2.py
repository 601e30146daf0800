"""Per-hypothesis bisect of the d_ref mismatch at (b=5, y=38, x=3): channels-last kernel against the first-generation one."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import oracle
from movedepth_amd import ops
from test_hip_parity import full_size_case, dev, host
from conftest import relerr
oracle.build()
B = 6
c = full_size_case(oracle, np.random.default_rng(21), B, 32, 16, 48, 160, 96, "white")
args = (dev(c["K"]), dev(c["invK"]), dev(c["pose"]), 16)
kw = dict(prior=dev(c["prior"]), ndepth=96, scale_fac=0.3, z_trans=dev(c["z"]), type="inverse", layout="ndhwc")

def run(gout, env):
    for k, v in env.items():
        os.environ[k] = v
    r, s = dev(c["ref"], True), dev(c["src"], True)
    vol = ops.costvol_grouped(r, s, *args, **kw)
    vol.backward(gout)
    for k in env:
        os.environ.pop(k)
    return host(r.grad), host(s.grad)

g = dev(c["gout"])
for env in ({}, {"MD_COSTVOL_TWO_PHASE": "0"}, {"MD_COSTVOL_NWG_BWD": "360"}, {"MD_COSTVOL_NWG_BWD": "1440"}, {"MD_COSTVOL_CL_NW_BWD": "8"}):
    dr, ds = run(g, env)
    print(env, "d_ref", relerr(dr, c["exp_dref"]), "d_src", relerr(ds, c["exp_dsrc"]), "b5", relerr(dr[5], c["exp_dref"][5]))
# per hypothesis
bad = []
for d in range(96):
    go = torch.zeros(B, 96, 16, 48, 160, device="cuda")
    go[5, d, :, 38, 3] = 1.0
    a, a_s = run(go, {})
    b_, b_s = run(go, {"MD_COSTVOL_CL": "0"})
    e = np.abs(a[5, :, 38, 3] - b_[5, :, 38, 3]).max()
    es = np.abs(a_s - b_s).max()
    if e > 1e-4 or es > 1e-4:
        bad.append(d)
        print("d=%d d_ref diff %.3e d_src diff %.3e  cl %s gen1 %s" % (d, e, es, a[5, :3, 38, 3], b_[5, :3, 38, 3]))
print("bad steps:", bad)
hyp = oracle.schedule_depth_range(c["prior"], 96, 0.3, c["z"], "inverse")
pix = oracle.backproject_project(hyp[5, :, 38, 3].reshape(96, 1), c["invK"][5:6], c["K"][5:6], c["pose"][5:6], 1, 1)[1] if False else None
# sample positions of that pixel along d (same arithmetic as the oracle: project the pixel at each hypothesis)
K, invK, T = c["K"][5].astype(np.float64), c["invK"][5].astype(np.float64), c["pose"][5].astype(np.float64)
ray = invK[:3, :3] @ np.array([3.0, 38.0, 1.0])
P = (K @ T)[:3]
for d in range(96):
    X = np.append(ray * hyp[5, d, 38, 3], 1.0)
    cc = P @ X
    print("d", d, "hyp %.4f" % hyp[5, d, 38, 3], "ix %.5f iy %.5f" % (cc[0] / cc[2], cc[1] / cc[2]), "*" if d in bad else "")
