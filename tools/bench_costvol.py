#!/usr/bin/env python3
"""Micro-benchmark of the plane-sweep kernels at BASELINE config 2 (B=6, 48x160, D=96, C=32, G=16, fp32).
Prints time per launch and algorithmic GB/s (bytes model of SURVEY 8d / DESIGN.md).  GPU only."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from movedepth_amd import ops  # noqa: E402


def time_fn(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=6)
    ap.add_argument("--h", type=int, default=48)
    ap.add_argument("--w", type=int, default=160)
    ap.add_argument("--D", type=int, default=96)
    ap.add_argument("--C", type=int, default=32)
    ap.add_argument("--G", type=int, default=16)
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--layout", default="bgd")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--dtype", default=os.environ.get("DT", "f32"), choices=["f32", "bf16", "f16"], help="feature-map and volume element type")
    ap.add_argument("--prior", default=os.environ.get("PRIOR", "white"), choices=["white", "smooth", "const", "kitti"],
                    help="depth prior: white noise per pixel in [2,22) (adversarial: neighbouring pixels sweep unrelated epipolar "
                         "segments), a smooth field (what the mono decoder produces), or a constant")
    ap.add_argument("--feat", default=os.environ.get("FEAT", "nhwc"), choices=["nchw", "nhwc"],
                    help="feature-map layout: planar [B,C,h,w] or channels-last [B,h,w,C] (what the encoder produces in channels_last)")
    ap.add_argument("--rotate", type=int, default=8,
                    help="write into N different output buffers in turn (N x 283 MB >> the 256 MB Infinity Cache), so that a "
                         "launch cannot benefit from lines of its own output left in cache by the previous launch")
    a = ap.parse_args()
    B, h, w, D, C, G = a.B, a.h, a.w, a.D, a.C, a.G
    torch.manual_seed(0)
    dev = "cuda"
    K = torch.tensor([[0.58 * w, 0, 0.5 * w, 0], [0, 1.92 * h, 0.5 * h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device=dev).repeat(B, 1, 1)
    invK = torch.linalg.pinv(K)
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    eb = 4 if a.dtype == "f32" else 2
    mf = torch.channels_last if a.feat == "nhwc" else torch.contiguous_format
    ref = torch.randn(B, C, h, w, device=dev).to(tdt).contiguous(memory_format=mf).requires_grad_(True)
    src = torch.randn(B, C, h, w, device=dev).to(tdt).contiguous(memory_format=mf).requires_grad_(True)
    if a.prior == "white":
        prior = 2 + 20 * torch.rand(B, 1, h, w, device=dev)
    elif a.prior == "smooth":
        coarse = torch.rand(B, 1, max(2, h // 12), max(2, w // 12), device=dev)
        prior = 2 + 20 * torch.nn.functional.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=True)
    elif a.prior == "kitti":   # driving scene: ground plane + facades (movedepth_amd/synthetic.py driving_scene)
        from movedepth_amd.synthetic import driving_scene
        prior = torch.from_numpy(driving_scene(B, h, w)[0]).to(dev)
    else:
        prior = torch.full((B, 1, h, w), float(os.environ.get("PRIOR_CONST", "8.0")), device=dev)
    pose = torch.eye(4, device=dev).repeat(B, 1, 1)
    pose[:, 0, 3], pose[:, 2, 3] = float(os.environ.get("POSE_TX", "0.05")), float(os.environ.get("POSE_TZ", "0.03"))
    pose[:, 1, 3] = float(os.environ.get("POSE_TY", "0.0"))
    if os.environ.get("POSE_ROT"):   # "wild" poses of an untrained pose network: axis-angle ~ N(0, rot^2), translation ~ N(0, trans^2)
        from movedepth_amd.layers import transformation_from_parameters
        gen = torch.Generator(device=dev).manual_seed(7)
        pose = transformation_from_parameters(torch.randn(B, 1, 3, device=dev, generator=gen) * float(os.environ["POSE_ROT"]),
                                              torch.randn(B, 1, 3, device=dev, generator=gen) * float(os.environ.get("POSE_TRANS", "2.0")))
    if os.environ.get("POSE_KITTI"):   # +-POSE_KITTI m along the optical axis, small yaw / pitch / roll (driving_scene)
        from movedepth_amd.synthetic import driving_scene
        pose = torch.from_numpy(driving_scene(B, h, w, speed=float(os.environ["POSE_KITTI"]))[1]).to(dev)
    hyp = ops.schedule_depth_range(prior, D, 0.3)
    kw = dict(prior=prior, ndepth=D, scale_fac=0.3) if a.fused else dict(depth_priors=hyp)
    vol = ops.costvol_grouped(ref, src, K, invK, pose, G, layout=a.layout, **kw)
    g = torch.randn_like(vol)

    # the backward's per-launch policy (ops.BackwardPolicy): BAL / TABLE = 0 | 1 pin the cost-balanced partition / the cell-table build,
    # unset: the automatic choice from the previous launch's census and per-item costs
    pol = ops.backward_policy()
    pol.reset()
    for envname, attr in (("BAL", "force_balance"), ("TABLE", "force_table"), ("FINE", "force_fine")):
        if os.environ.get(envname) in ("0", "1"):
            setattr(pol, attr, os.environ[envname] == "1")

    keep = [None] * max(a.rotate - 1, 0)
    cnt = [0]

    def fwd():
        with torch.no_grad():
            # rotate=1: the result is dropped at once, the caching allocator hands the same block to every launch (the
            # first version of this script always did that).  rotate=N: the last N-1 results stay alive while the next
            # one is allocated, so N distinct blocks are written in turn.
            out = ops.costvol_grouped(ref, src, K, invK, pose, G, layout=a.layout, **kw)
            if keep:
                keep[cnt[0] % len(keep)] = out
                cnt[0] += 1

    def bwd():
        vol.backward(g, retain_graph=True)

    hyp_bytes = 4 * B * h * w if a.fused else 4 * B * D * h * w
    fbytes = 2 * eb * B * C * h * w + hyp_bytes + eb * B * D * G * h * w + 192 * B
    bbytes = eb * B * D * G * h * w + 2 * eb * B * C * h * w + hyp_bytes + 2 * 4 * B * C * h * w
    bigs = [torch.empty_like(vol.contiguous()) for _ in range(max(a.rotate, 1))]
    big = bigs[0]
    zc = [0]

    def fill():
        bigs[zc[0] % len(bigs)].zero_()
        zc[0] += 1

    t_fill = time_fn(fill, a.iters)
    src_big = torch.randn_like(big)
    t_copy = time_fn(lambda: big.copy_(src_big), a.iters)
    print("  ref (rotate=%d): zero_ of %.0f MB %.1f us (%.0f GB/s); copy_ (one buffer pair) %.1f us (%.0f GB/s r+w)" % (
        len(bigs), big.numel() * 4 / 1e6, t_fill, big.numel() * 4 / t_fill / 1e3, t_copy, 2 * big.numel() * 4 / t_copy / 1e3))
    del bigs
    # one backward first: the forward's slicing (ops.BackwardPolicy.forward_flags) follows the census of the last backward of this shape,
    # as it does from the second training step on
    vol.backward(g, retain_graph=True)
    torch.cuda.synchronize()
    ops.enable_library_kernel_timing(True)
    tf = time_fn(fwd, a.iters)
    tb = time_fn(bwd, a.iters)
    torch.cuda.synchronize()
    sfx = {"f32": "", "bf16": "_bf16", "f16": "_f16"}[a.dtype]
    lib_t = ops.library_kernel_times_us(["md_costvol_fwd" + sfx, "md_costvol_bwd" + sfx])
    ops.enable_library_kernel_timing(False)
    env = {k: v for k, v in os.environ.items() if k.startswith("MD_") or k in ("BAL", "TABLE", "FINE")}
    cst = pol.costs()
    if cst is not None and cst.sum() > 0:
        print("  backward policy: %d launches, %d on the cell-table build, %d on a cost-balanced partition, %d forwards on fine slices; gathered share %.3f; "
              "windows per segment %.3f; item cycles max / mean %.2f" % (pol.launches, pol.table_launches, pol.balanced_launches, pol.fine_launches,
                                                                         pol.gathered_share(), pol.windows_per_segment(), float(cst.max()) * len(cst) / float(cst.sum())))
    print("costvol B=%d %dx%d D=%d C=%d G=%d fused=%d layout=%s feat=%s dtype=%s prior=%s env=%s" % (B, h, w, D, C, G, a.fused, a.layout, a.feat, a.dtype, a.prior, env))
    print("  fwd %8.1f us  %7.1f MB  %7.0f GB/s (%.1f%% of 8 TB/s)" % (tf, fbytes / 1e6, fbytes / tf / 1e3, fbytes / tf / 1e3 / 80))
    print("  bwd %8.1f us  %7.1f MB  %7.0f GB/s (%.1f%% of 8 TB/s)  [includes 2 memsets + autograd glue]" % (tb, bbytes / 1e6, bbytes / tb / 1e3, bbytes / tb / 1e3 / 80))
    if os.environ.get("MD_CV_STATS"):
        import ctypes
        from movedepth_amd import _lib
        lib = _lib.load()
        for nm, fn in (("fwd", fwd), ("bwd", bwd)):
            lib.md_costvol_stats(1, None)
            fn()
            wgb = (ctypes.c_ulonglong * 8192)()
            nwg = lib.md_costvol_stats_wg(wgb, 8192)
            buf = (ctypes.c_ulonglong * 8)()
            lib.md_costvol_stats(0, buf)
            if os.environ.get("MD_CV_WGSTATS"):   # a -DMD_CL_WGSTATS=1 build: 8 numbers per workgroup
                import numpy as np
                rec = np.array(list(wgb)[:nwg], dtype=np.float64).reshape(-1, 8)
                rec = rec[rec[:, 0] > 0]
                names = ["lifetime", "windows", "gather sub-slices / gifts", "gather steps / own range done at", "cell-change blocks", "lanes in them", "fit attempts", "miss redo / ranges taken"]
                print("  per-workgroup records %s: %d workgroups" % (nm, len(rec)))
                for i, n_ in enumerate(names):
                    c = rec[:, i]
                    print("    %-36s mean %10.1f  median %10.1f  p90 %10.1f  max %10.1f  corr with lifetime %+.2f" % (
                        n_, c.mean(), np.median(c), np.percentile(c, 90), c.max(), (np.corrcoef(c, rec[:, 0])[0, 1] if c.std() > 0 else 0.0)))
                A = np.stack([np.ones(len(rec)), rec[:, 1], rec[:, 3], rec[:, 4], rec[:, 6]], 1)
                coef, *_ = np.linalg.lstsq(A, rec[:, 0], rcond=None)
                print("    least squares: lifetime ~ %.0f + %.0f x windows + %.0f x gather steps + %.0f x blocks + %.0f x fit attempts (cycles)" % tuple(coef))
                if os.environ.get("MD_CV_WGSTATS_DUMP"):
                    np.save(os.environ["MD_CV_WGSTATS_DUMP"] + "_" + nm + ".npy", rec)
                lib.md_costvol_stats(0, None)
                continue
            life = sorted(x for x in list(wgb)[:nwg] if x)
            if life:
                n = len(life)
                print("  workgroup lifetimes %s (shader cycles): %d workgroups, mean %.0f, median %d, p90 %d, max %d  (max / mean %.2f)" % (
                    nm, n, sum(life) / n, life[n // 2], life[(9 * n) // 10], life[-1], life[-1] * n / sum(life)))
            v = list(buf)
            if os.environ.get("MD_CV_TIMELINE"):
                print("  timeline %s (shader cycles of thread 0, summed over workgroups): %s" % (nm, v))
                continue
            print("  stats %s: segments %d, windows staged %d, fit attempts %d, lanes redoing misses %d, cell-change blocks %d "
                  "(%.1f lanes each), wave-steps %d" % (nm, v[0], v[1], v[2], v[3], v[4], v[5] / max(v[4], 1), v[6]))
    for k, v in lib_t.items():
        nb = fbytes if "fwd" in k else bbytes
        print("  kernel only (dispatch start/stop events inside the library) %-24s avg %7.1f us  min %7.1f us  %d launches  %.1f%% of 8 TB/s" % (
            k, v["avg_us"], v["min_us"], v["launches"], nb / v["avg_us"] / 1e3 / 80))


if __name__ == "__main__":
    main()
