"""Where does d_ref differ from the oracle at the config-2 launch shape with a white-noise prior?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import oracle
from movedepth_amd import ops
from test_hip_parity import full_size_case, dev, host
from conftest import relerr
oracle.build()
B = int(os.environ.get("DB", 6))
c = full_size_case(oracle, np.random.default_rng(21), B, 32, 16, 48, 160, 96, os.environ.get("PRIOR", "white"))
r, s = dev(c["ref"], True), dev(c["src"], True)
vol = ops.costvol_grouped(r, s, dev(c["K"]), dev(c["invK"]), dev(c["pose"]), 16, prior=dev(c["prior"]), ndepth=96,
                          scale_fac=0.3, z_trans=dev(c["z"]), type="inverse", layout=os.environ.get("LAYOUT", "ndhwc"))
vol.backward(dev(c["gout"]))
for nm, got, exp in (("d_ref", host(r.grad), c["exp_dref"]), ("d_src", host(s.grad), c["exp_dsrc"])):
    e = np.abs(got - exp)
    print(nm, "rel", relerr(got, exp), "max", e.max(), "scale", np.abs(exp).max())
    for b in range(B):
        print("  b=%d rel %.2e" % (b, relerr(got[b], exp[b])), "z", c["z"][b])
    eb = e.sum(axis=1)       # (B,h,w)
    thr = 1e-3 * np.abs(exp).max() * exp.shape[1]
    bad = np.argwhere(eb > thr)
    print("  pixels with summed-channel error >", thr, ":", len(bad), "of", eb.size)
    if len(bad):
        print("  first bad:", bad[:20].tolist())
        ys, xs = bad[:, 1], bad[:, 2]
        print("  y hist (rows%8):", np.bincount(ys % 8, minlength=8).tolist(), " x hist (cols%16):", np.bincount(xs % 16, minlength=16).tolist())
        bb, yy, xx = bad[0]
        print("  got", got[bb, :4, yy, xx], "exp", exp[bb, :4, yy, xx])
