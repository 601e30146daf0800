#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): train-step images/s at 192x640, D=96 (config 2: ResNet-18, batch 6 per
GPU, fp32, synthetic KITTI-shaped frames resident in HBM) + the plane-sweep kernel's HBM roofline fraction.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W          # one rank per GPU, RCCL gradient all-reduce

A step = Trainer.process_batch + backward + optimizer step on one batch.  Prints ONE JSON line on rank 0.
`roofline`: md_costvol_fwd timed live with HIP events (torch's current stream = the launch stream) inside the
timed region; algorithmic bytes per launch from DESIGN.md's model.  `cpu_baseline`: the C oracle (a port of the
reference's CPU path, NOT the product) on a bounded sample, N=1 rank 0 only.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (guides/MI355X_MICROARCH.md)


def costvol_source_hash():
    """SHA-256 over the sources the plane-sweep kernels are compiled from: the committed PMC summary (profiles/
    costvol_fwd_pmc.json, tools/make_profiles.sh) is stamped with it, and is quoted as `roofline.traffic` only for the kernels
    it was collected on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("costvol_cl.inc", "costvol.hip", "md_common.hpp"):
        h.update(open(os.path.join(ROOT, "movedepth_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def costvol_fwd_bytes(B, C, G, h, w, D, fused, eb=4):
    """ref + src features, hypotheses (or the prior when the schedule is fused), grouped volume, K/invK/T;
    eb = bytes per feature / volume element (4, or 2 under --amp)."""
    hyp = 4 * B * h * w if fused else 4 * B * D * h * w
    return 2 * eb * B * C * h * w + hyp + eb * B * D * G * h * w + 192 * B


def usable_cpus():
    """CPUs this process can actually keep busy: the affinity mask, capped by the cgroup's CPU-time quota (a container that sees
    256 hardware threads but is granted 16 CPUs' worth of time runs 128 OpenMP threads SLOWER than one: measured on the pool's
    boxes, tools/diag/cpu_scaling_probe.py).  -> (count, description)"""
    import math

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    desc = "no cgroup CPU quota"
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            lim = max(1, int(math.ceil(int(q) / int(per))))
            desc = "cgroup quota %s/%s = %d CPUs" % (q, per, lim)
            n = min(n, lim)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                lim = max(1, int(math.ceil(q / per)))
                desc = "cgroup quota %d/%d = %d CPUs" % (q, per, lim)
                n = min(n, lim)
        except Exception:
            pass
    return max(1, n), desc


def cpu_baseline(opt):
    """Hot path of ONE sample of the workload (1/6 of a batch) through the C oracle, all host threads."""
    import numpy as np

    import oracle

    rng = np.random.default_rng(0)
    H, W, D, C, G = opt.height, opt.width, opt.num_depth_bins, 32, opt.reg3d_c
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

    def one_sample():
        hyp = oracle.schedule_depth_range(prior, D, 0.3, None, "inverse")
        for _ in range(2):  # plain + mask-augmented pass
            oracle.costvol_grouped(ref, src, Kq, iKq, hyp, T, G)
            oracle.costvol_grouped_bwd(gvol, ref, src, Kq, iKq, hyp, T)
        for _ in range(12):  # 8 mono + 2 MVS + 2 fuse warps, each with its SSIM+L1 loss, forward and backward
            warped, _ = oracle.warp(img, depth, K0, iK0, T)
            oracle.reproj_loss(warped, tgt)
            gp = oracle.reproj_loss_bwd(gl, warped, tgt)
            oracle.warp_bwd(gp, img, depth, K0, iK0, T)
        for _ in range(2):  # identity losses
            oracle.reproj_loss(img, tgt)
        for s in range(4):
            oracle.smooth_loss(depth[:, :, ::2 ** s, ::2 ** s], img[:, :, ::2 ** s, ::2 ** s], True)

    def timed(budget_s, max_runs):
        one_sample()  # warm
        t0 = time.time()
        n = 0
        while n < 2 or (time.time() - t0 < budget_s and n < max_runs):
            one_sample()
            n += 1
        return (time.time() - t0) / n, n

    # ONE thread: what the reference itself runs on (it pins OMP / MKL to one thread per process, trainer.py:2-4) -- the headline
    # figure of this object, as in earlier rounds.  ALL cores beside it: the oracle's loops run over (sample, plane,
    # row) with thread-private accumulators (no atomics), so one sample scales; the thread count that ran fastest of
    # {every hardware thread, half of them (one per core where SMT is on)} is reported with its count.
    ncpu, quota = usable_cpus()
    oracle.set_num_threads(1)
    dt1, n1 = timed(12.0, 8)
    best = None
    for nt in sorted({ncpu, max(1, ncpu // 2)}, reverse=True):
        if nt == 1:
            continue
        oracle.set_num_threads(nt)
        dtn, nn = timed(4.0, 40)
        if best is None or dtn < best[0]:
            best = (dtn, nn, nt)
    oracle.set_num_threads(1)
    res = {"value": 1.0 / dt1, "unit": "images/s (hot path only, no conv nets)", "cores": 1, "kind": "port",
           "ms_per_image": 1e3 * dt1,
           "sample": "1 sample (1/6 batch) of config 2: 2x cost volume fwd+bwd (48x160, D=%d, C=32->G=%d), 12x "
                     "warp+SSIM/L1 fwd+bwd at %dx%d, identity + smoothness losses; C oracle (a port of the reference's "
                     "CPU path, not the product), %d runs of %.2f s on 1 thread (what the reference's trainer.py:2-4 forces; "
                     "%d hardware threads on this box, %s)" % (D, G, H, W, n1, dt1, os.cpu_count() or 0, quota)}
    if best is not None:
        res["all_cores"] = {"value": 1.0 / best[0], "unit": "images/s (hot path only, no conv nets)", "cores": best[2],
                            "ms_per_image": 1e3 * best[0], "speedup_over_1_thread": dt1 / best[0],
                            "sample": "the same sample, %d runs of %.3f s with %d OpenMP threads (%s)" % (best[1], best[0], best[2], quota)}
    return res


HOT_PATH_ENTRY_POINTS = ["md_costvol_fwd", "md_costvol_bwd", "md_photo_fwd", "md_photo_bwd", "md_pack_rgbx", "md_smooth_multi_fwd",
                         "md_smooth_multi_bwd"]


def gpu_hot_path_ms_per_image(opt, device, iters=20):
    """The SAME list of hot-path operations the cpu_baseline times (2x plane sweep forward + backward, the photometric groups
    of one step forward + backward, identity and smoothness losses; no convolution networks), on the GPU at the workload's
    batch size, in ms per image: the like-for-like figure beside cpu_baseline (the headline images/s includes the networks)."""
    from movedepth_amd import ops

    B, H, W, D, C, G = opt.batch_size, opt.height, opt.width, opt.num_depth_bins, 32, opt.reg3d_c
    h, w = H // 4, W // 4
    g = torch.Generator(device=device).manual_seed(0)
    rnd = lambda *sh: torch.randn(*sh, device=device, generator=g)
    uni = lambda *sh: torch.rand(*sh, device=device, generator=g)
    ref, src = rnd(B, C, h, w).requires_grad_(True), rnd(B, C, h, w).requires_grad_(True)

    def Kmat(hh, ww):
        K = torch.tensor([[0.58 * ww, 0, 0.5 * ww, 0], [0, 1.92 * hh, 0.5 * hh, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device=device)
        return K.repeat(B, 1, 1), torch.linalg.pinv(K).repeat(B, 1, 1)

    Kq, iKq = Kmat(h, w)
    K0, iK0 = Kmat(H, W)
    T = torch.eye(4, device=device).repeat(B, 1, 1)
    T[:, 0, 3], T[:, 2, 3] = 0.05, 0.03
    Ts = [T.clone().requires_grad_(True), T.clone().requires_grad_(True)]
    prior = 2 + 20 * uni(B, 1, h, w)
    target, srcs = uni(B, 3, H, W), [uni(B, 3, H, W), uni(B, 3, H, W)]
    disps = [(0.02 + 0.3 * uni(B, 1, H >> s, W >> s)).requires_grad_(True) for s in range(4)]
    imgs = [uni(B, 3, H >> s, W >> s) for s in range(4)]
    depth = (2 + 20 * uni(B, H, W)).requires_grad_(True)
    noise = rnd(4, B, 1, H, W) * 1e-5
    gvol = None

    def one_batch():
        nonlocal gvol
        for _ in range(2):  # plain + mask-augmented pass
            vol = ops.costvol_grouped(ref, src, Kq, iKq, T, G, prior=prior, ndepth=D, scale_fac=0.3, layout="ndhwc")
            if gvol is None:
                gvol = torch.randn_like(vol)
            vol.backward(gvol)
        ident = ops.identity_loss(target, srcs)
        mono = ops.photometric_loss(target, srcs, Ts, K0, iK0, disps, is_disp=True, ident_min=ident, noise=noise, want_pix=True)
        sm = ops.smooth_losses(disps, imgs)
        (sum(mono["loss"]) + sum(sm)).backward()
        for kw in (dict(mvs_mode=True, want_oob=True, want_mask=True), dict(ssim_w=0.0, want_mask=True)):  # MVS, fused depth
            out = ops.photometric_loss(target, srcs, [t.detach() for t in Ts], K0, iK0, [depth], **kw)
            out["loss"][0].backward()

    for _ in range(3):
        one_batch()
    torch.cuda.synchronize()
    # Two clocks.  (1) The sum of the kernels' own durations, from start / stop events tied to every dispatch inside the library
    # (md_kernel_timing_*, the clock rocprofv3's kernel trace reads): the figure to compare across boxes.  (2) Wall time around
    # the loop: includes the host's launch latency whenever the GPU runs dry, which on ~1 ms of kernels per batch it does --
    # round 3's figure moved 1.8x between two hosts for that reason.  The two gradient fills (hipMemsetAsync, ~3 us each) are
    # DMA / blit operations, not dispatches of ours, and are in (2) only.
    ops.enable_library_kernel_timing(True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        one_batch()
    b.record()
    torch.cuda.synchronize()
    per_entry = ops.library_kernel_times_us(HOT_PATH_ENTRY_POINTS)
    ops.enable_library_kernel_timing(False)
    kern_ms = sum(sum(v["all_us"]) for v in per_entry.values()) * 1e-3 / iters / B
    return kern_ms, a.elapsed_time(b) / iters / B, {k: {"us_per_batch": sum(v["all_us"]) / iters, "dispatches_per_batch": v["launches"] / iters}
                                                    for k, v in per_entry.items()}


def parallax_cases(opt, device, iters=24, rotate=8):
    """The roofline kernels at the workload's launch shape on inputs with the parallax the headline's synthetic batch does not have
    (BASELINE.md 3 puts translations at 3-5 cm against 2-22 m; a random-init pose network adds little): stand-alone launches,
    outputs rotating over `rotate` volumes so that a launch never meets its own lines in the Infinity Cache, durations from the
    dispatch events inside the library.  Cases (movedepth_amd/synthetic.driving_scene; tests/test_hip_parity.py holds the kernels to
    the oracle on exactly these inputs' kind): a driving scene at 1 m per frame (t_z / depth 0.0125-0.2), the same at 2 m, and
    'moderate' poses (axis-angle N(0, 0.05^2), translation N(0, 0.3^2)) against a steep smooth prior of 2-22 m.  Secondary figures:
    the headline `roofline` stays the forward inside the training step."""
    from movedepth_amd import ops
    from movedepth_amd.layers import transformation_from_parameters
    from movedepth_amd.synthetic import driving_scene

    B, D, C, G = opt.batch_size, opt.num_depth_bins, 32, opt.reg3d_c
    h, w = opt.height // 4, opt.width // 4
    g = torch.Generator(device=device).manual_seed(5)
    mk = lambda: torch.randn(B, C, h, w, device=device, generator=g).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ref, src = mk(), mk()
    K = torch.tensor([[0.58 * w, 0, 0.5 * w, 0], [0, 1.92 * h, 0.5 * h, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device=device).repeat(B, 1, 1)
    iK = torch.linalg.pinv(K)
    coarse = torch.rand(B, 1, max(2, h // 12), max(2, w // 12), device=device, generator=g)
    smooth = 2 + 20 * torch.nn.functional.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=True)
    gp = torch.Generator(device=device).manual_seed(7)
    moderate = transformation_from_parameters(torch.randn(B, 1, 3, device=device, generator=gp) * 0.05,
                                              torch.randn(B, 1, 3, device=device, generator=gp) * 0.3)
    cases = {}
    for name, speed in (("driving scene, 1 m per frame", 1.0), ("driving scene, 2 m per frame", 2.0)):
        pr, po = driving_scene(B, h, w, speed=speed)
        cases[name] = (torch.from_numpy(pr).to(device), torch.from_numpy(po).to(device))
    cases["moderate poses (axis-angle N(0,0.05^2), translation N(0,0.3^2)), smooth prior 2-22 m"] = (smooth, moderate)
    fbytes = 2 * 4 * B * C * h * w + 4 * B * h * w + 4 * B * D * G * h * w + 192 * B
    bbytes = 4 * B * D * G * h * w + 2 * 4 * B * C * h * w + 4 * B * h * w + 2 * 4 * B * C * h * w
    out = {}
    for name, (prior, pose) in cases.items():
        keep = [None] * (rotate - 1)
        vol = ops.costvol_grouped(ref, src, K, iK, pose, G, prior=prior, ndepth=D, scale_fac=0.3, layout="ndhwc")
        gvol = torch.randn_like(vol)
        for it in range(iters + 4):
            if it == 4:
                torch.cuda.synchronize()
                ops.enable_library_kernel_timing(ops.TIME_ROOFLINE)
            with torch.no_grad():
                keep[it % len(keep)] = ops.costvol_grouped(ref, src, K, iK, pose, G, prior=prior, ndepth=D, scale_fac=0.3, layout="ndhwc")
            vol.backward(gvol, retain_graph=True)
        torch.cuda.synchronize()
        t = ops.library_kernel_times_us(["md_costvol_fwd", "md_costvol_bwd"])
        ops.enable_library_kernel_timing(False)
        f, bw = t["md_costvol_fwd"], t["md_costvol_bwd"]
        out[name] = {"fwd_avg_us": f["avg_us"], "fwd_frac": fbytes / f["avg_us"] * 1e-3 / HBM_PEAK_GBS, "bwd_avg_us": bw["avg_us"],
                     "bwd_frac": bbytes / bw["avg_us"] * 1e-3 / HBM_PEAK_GBS, "launches": f["launches"]}
        del keep, vol, gvol
    return {"shape": "B=%d, %dx%d, D=%d, C=%d, G=%d, fp32, channels-last features and volume, stand-alone launches on %d rotating outputs" % (
                B, h, w, D, C, G, rotate),
            "algorithmic_bytes_per_launch": {"fwd": fbytes, "bwd": bbytes}, "peak": HBM_PEAK_GBS, "unit": "GB/s", "cases": out}


def start_watchdog(seconds):
    """A hang must become a non-zero exit code.  torch's own watchdog covers collectives issued through torch.distributed (the
    process group is created with a timeout, trainer.py); it does not see a rank that stalls elsewhere, nor ncclAllReduce calls
    made directly (rccl_direct, opt-in).  A daemon thread ends the PROCESS -- os._exit, no clean-up that could block on the same
    hang -- so that torch.distributed.run tears the other ranks down and the launcher returns rc != 0."""
    if seconds <= 0:
        return None
    import threading

    def fire():
        sys.stderr.write("bench.py: watchdog: not finished after %.0f s (rank %s) -- aborting with exit code 124\n" % (seconds, os.environ.get("RANK", "0")))
        sys.stderr.flush()
        os._exit(124)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)     # >= 20 warm-up + >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch_per_gpu", type=int, default=6)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--trainer_args", default="", help="extra movedepth_amd options, e.g. '--hip_prob_conv 0' for an A/B")
    ap.add_argument("--epoch", type=int, default=0, help="trainer.epoch during the run (> ztrans_start_epc: velocity-guided bins)")
    ap.add_argument("--watchdog_s", type=float, default=float(os.environ.get("MD_BENCH_WATCHDOG_S", "0")),
                    help="abort with exit code 124 when the run has not finished after this many seconds (default: 1800 + 60 per step for --gpus > 1, "
                         "where a rank that died or a collective that cannot complete leaves the others waiting; off for one GPU)")
    a = ap.parse_args()
    # default for N > 1: 30 minutes plus a generous per-step allowance (an 8-rank shared-GPU gloo run takes 22 s per step: a long
    # --steps run must not be reported as a hang)
    multi = a.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1
    start_watchdog(a.watchdog_s if a.watchdog_s > 0 else ((1800.0 + 60.0 * (a.steps + a.warmup)) if multi else 0.0))
    # stdout carries exactly one JSON line.  The convolution libraries write diagnostics to file descriptor 1 from C++ (CK's
    # "GridwiseOp: Problemsize descriptor dimension check failure" under fp16 autocast): point fd 1 at stderr for the run and
    # keep the real stdout for the line.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("launch --gpus %d with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (a.gpus, a.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # MIOpen's find results / compiled kernels, shipped in-tree (a private copy per rank), see movedepth_amd/miopen_setup.py
    from movedepth_amd import miopen_setup

    miopen_setup.use_shipped_cache(rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

    from movedepth_amd import ops
    from movedepth_amd.options import MovedepthOptions
    from movedepth_amd.synthetic import make_inputs
    from movedepth_amd.trainer import Trainer

    argv = ["--height", "192", "--width", "640", "--num_depth_bins", "96", "--batch_size", str(a.batch_per_gpu),
            "--res_arch", "18", "--prior_scale", "2", "--convex_up", "--weights_init", "scratch", "--learning_rate", "2e-4",
            "--local_rank", str(local_rank)]
    argv += a.trainer_args.split()
    if "--miopen_find" not in argv and a.trainer_args:
        argv += ["--miopen_find", "1"]  # another workload: its 2-D convolutions are not in the shipped db, a search would take minutes
    if "--miopen_find" not in argv:
        # every convolution on its searched solver when the shipped find-db is honoured here (47.3 -> 43.5 ms per step), the
        # 3-D regulariser's only otherwise (a full search of this workload is ~9 minutes)
        share = os.environ.get("MD_SHARE_GPU", "0") == "1"
        argv += ["--miopen_find", "2" if miopen_setup.find_db_hits(0 if share else local_rank) else "1"]
    if world > 1:
        argv.append("--ddp")
    share_gpu = os.environ.get("MD_SHARE_GPU", "0") == "1"
    if share_gpu:
        argv[argv.index("--local_rank") + 1] = "0"
    opt = MovedepthOptions().parse(argv)
    torch.manual_seed(1234 + rank)
    import numpy as np

    np.random.seed(1234 + rank)
    import contextlib

    with contextlib.redirect_stdout(sys.stderr):  # the trainer's banner (reference trainer.py:163-165) must not share
        trainer = Trainer(opt)                     # stdout with the one JSON line
    trainer.set_train()
    trainer.epoch = a.epoch
    dev = trainer.device
    inputs = make_inputs(opt.batch_size, opt.height, opt.width, opt.frame_ids, seed=rank, device=dev)  # resident in HBM

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        trainer.train_step(dict(inputs))
    CB_KERNELS = ["md_conv3d_cb_fwd", "md_conv3d_cb_bwd_data", "md_conv3d_cb_bwd_weight"]   # reg3d.conv2 (32 -> 32) as 16 x 16 channel blocks
    CONV_KERNELS = CB_KERNELS + ["md_conv3d_c16_fwd", "md_conv3d_c16_bwd_data", "md_conv3d_c16_bwd_weight", "md_conv3d_c1_fwd",
                    "md_conv3d_c1_bwd_data", "md_conv3d_c1_bwd_weight"]
    sfx = {"none": "", "bf16": "_bf16", "fp16": "_f16"}[opt.amp]
    # plane-sweep kernels: HIP events recorded inside the library directly around the kernel launch, on the launch stream
    # (events recorded from Python around the ctypes call also time the host's launch latency whenever the GPU has run dry:
    # 77 vs 57 us for the forward inside this step); the convolution kernels are read from the same in-library events below and
    # additionally timed from Python around the whole call (entry_point_avg_us)
    EXTRA = [k for k in os.environ.get("MD_BENCH_EXTRA_KERNELS", "").split(",") if k]   # diagnostics: more entry points, to stderr
    ops.enable_kernel_timing(CONV_KERNELS + EXTRA)
    # only the roofline kernels are timed inside the timed region: a timed dispatch costs the stream ~5 us, and timing the 444
    # BatchNorm launches of a synchronised-BatchNorm step read as +2.6 ms per step; the photometric / BatchNorm kernels
    # are timed over PHOTO_STEPS extra steps after it
    ops.enable_library_kernel_timing(ops.TIME_ROOFLINE)
    if os.environ.get("MD_CV_STATS"):   # diagnostics: work counters of the plane-sweep kernels over the timed steps
        from movedepth_amd import _lib as _l
        _l.load().md_costvol_stats(1, None)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        _, losses = trainer.train_step(dict(inputs))
    host_issue = time.perf_counter() - t0   # the launching thread's own time for K steps (nothing in a step waits for the GPU)
    barrier()
    elapsed = time.perf_counter() - t0
    loss_val = float(losses["loss"].detach())
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if os.environ.get("MD_CV_STATS"):
        import ctypes as _ct
        _buf = (_ct.c_ulonglong * 8)()
        _l.load().md_costvol_stats(0, _buf)
        print("costvol stats over %d steps (fwd+bwd launches together): segments %d, windows staged %d, fit attempts %d, "
              "lanes redoing misses %d, cell-change blocks %d (%.1f lanes each), wave-steps %d" % (
                  a.steps, _buf[0], _buf[1], _buf[2], _buf[3], _buf[4], _buf[5] / max(_buf[4], 1), _buf[6]), file=sys.stderr)
    times = ops.kernel_times_us()
    for k_ in EXTRA:
        if k_ in times:
            print("%s: %d launches, %.1f us per step (host-side events around the call)" % (
                k_, times[k_]["launches"], times[k_]["avg_us"] * times[k_]["launches"] / a.steps), file=sys.stderr)
    times.update(ops.library_kernel_times_us(["md_costvol_fwd" + sfx, "md_costvol_bwd" + sfx]))
    # the convolution kernels: the same dispatch events (main kernel only -- the weight gradients' small finish kernels are
    # separate dispatches); the Python-side figure (argument checks, workspace, finish kernel, launch gaps) is kept beside it
    entry_us = {k_: times[k_]["avg_us"] for k_ in CONV_KERNELS if k_ in times}
    times.update(ops.library_kernel_times_us(CONV_KERNELS))
    # the photometric / smoothness kernels inside the step (all dispatches of an entry point together: forward = main + finish
    # kernels, backward = main + finish + up-sampling adjoint): kernel time per step and the slowest single dispatch
    photo_in_step = {}
    PHOTO_STEPS = int(os.environ.get("MD_BENCH_PHOTO_STEPS", "10"))   # 0 under tools/profile_step.sh, which counts the steps in the trace
    ops.enable_library_kernel_timing(ops.TIME_PHOTOMETRIC | ops.TIME_BATCHNORM)   # (drops the records read above)
    for _ in range(PHOTO_STEPS):
        trainer.train_step(dict(inputs))
    torch.cuda.synchronize()
    for k_, v_ in ops.library_kernel_times_us([n_ for n_ in HOT_PATH_ENTRY_POINTS if not n_.startswith("md_costvol")] +
                                              ["md_bn_stats", "md_bn_apply", "md_bn_bwd_reduce", "md_bn_bwd_dx"]).items():
        if PHOTO_STEPS:
            photo_in_step[k_] = {"us_per_step": sum(v_["all_us"]) / PHOTO_STEPS, "dispatches_per_step": v_["launches"] / PHOTO_STEPS,
                             "max_dispatch_us": max(v_["all_us"])}
    # the photometric groups against memory: ALGORITHMIC bytes per step (the mono group's forward reads 54 MB and writes
    # 168 MB -- 8 warped frames, 8 sample grids, 4 depths -- = 222 MB, its backward 148 MB; a one-scale group (MVS, fused depth) 74 MB forward;
    # the identity loss 38 MB; the one-scale backwards pro rata) over the entry point's kernel time in the step.  These kernels are not
    # memory-bound -- the vector ALU is the busiest unit (~52 % of the cycles beside the per-pixel gathers; 13-27 % of HBM): the counters
    # are in profiles/r06_photo_pmc.txt -- the figure is here because the roofline of every hot-path kernel is asked for.
    if not a.trainer_args and opt.height == 192 and opt.width == 640 and opt.batch_size == 6:
        # (with --lazy_sample_grids the mono forward does not write its 8 sample grids: 8 x B*H*W*8 B = 47 MB less)
        grids = 0.0 if getattr(opt, "lazy_sample_grids", 0) else 8 * opt.batch_size * opt.height * opt.width * 8.0
        for k_, nbytes in (("md_photo_fwd", 222e6 - 47.2e6 + grids + 2 * 74e6 + 38e6), ("md_photo_bwd", 148e6 + 2 * 74e6 * 148.0 / 222.0)):
            if k_ in photo_in_step:
                e_ = photo_in_step[k_]
                e_["algorithmic_bytes_per_step"] = nbytes
                e_["achieved_gbs"] = nbytes / e_["us_per_step"] * 1e-3
                e_["frac_of_hbm_peak"] = e_["achieved_gbs"] / HBM_PEAK_GBS
                e_["bound"] = "vector ALU ~52 % busy + per-pixel gathers (profiles/r06_photo_pmc.txt), not HBM"
    if os.environ.get("MD_BENCH_DUMP_TIMES"):
        for k_ in ("md_costvol_fwd" + sfx, "md_costvol_bwd" + sfx):
            print(k_, " ".join("%.0f" % t for t in times.get(k_, {}).get("all_us", [])), file=sys.stderr)
    ops.enable_library_kernel_timing(False)

    # What one step sends, per rank (the first real N-GPU run should diagnose itself): one more step, outside the
    # timed region, with the counters of the two places a collective is issued from (dp.GradSync._reduce: gradient buckets;
    # ops._group_all_reduce: the BatchNorm layers' 2C sums), gathered to rank 0.
    collectives = None
    if world > 1 and trainer.grad_sync is not None:
        gs = trainer.grad_sync
        c0, b0, n0, y0 = gs.reduce_calls, gs.reduce_bytes, ops.GROUP_ALL_REDUCE[0], ops.GROUP_ALL_REDUCE[1]
        trainer.train_step(dict(inputs))
        torch.cuda.synchronize()
        mine = {"rank": rank, "bucket_all_reduces": gs.reduce_calls - c0, "bucket_bytes": gs.reduce_bytes - b0,
                "batchnorm_all_reduces": ops.GROUP_ALL_REDUCE[0] - n0, "batchnorm_bytes": ops.GROUP_ALL_REDUCE[1] - y0}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        direct = trainer.direct_all_reduce is not None
        collectives = {
            "path": ("ncclAllReduce on the compute stream, one communicator for buckets and BatchNorm statistics (MD_DIRECT_RCCL=1)" if direct else
                     "torch.distributed all_reduce on the process group's stream (%s), buckets async behind backward" % dist.get_backend()),
            "per_step_per_rank": per_rank,
            "identical_on_every_rank": all({k: v for k, v in r.items() if k != "rank"} == {k: v for k, v in per_rank[0].items() if k != "rank"}
                                           for r in per_rank),
            "grad_bucket_mb": opt.grad_bucket_mb, "buckets": len(gs.buckets), "sync_bn": bool(opt.sync_bn), "sync_bn_impl": opt.sync_bn_impl,
        }
    if rank == 0:
        gb = opt.batch_size * world
        h, w = opt.height // 4, opt.width // 4
        fbytes = costvol_fwd_bytes(opt.batch_size, 32, opt.reg3d_c, h, w, opt.num_depth_bins, fused=True, eb=2 if sfx else 4)
        kt = times.get("md_costvol_fwd" + sfx, {})
        ach = fbytes / (kt["avg_us"] * 1e-6) / 1e9 if kt else None
        # HBM bytes per launch from the PMC counters are NOT measured in this run: they come from a separate rocprofv3 --pmc
        # pass (tools/pmc_costvol.sh) whose result is committed under profiles/; reported under its own key with provenance
        traffic_profile, traffic = None, None
        pmc = os.path.join(ROOT, "profiles", "costvol_fwd_pmc.json")
        if os.path.exists(pmc) and not sfx:   # the counter file was collected for the fp32 kernel
            traffic_profile = json.load(open(pmc))
            # quoted as `traffic` only when the counters were collected on exactly these kernel sources
            if traffic_profile.get("kernel_source_sha256") == costvol_source_hash():
                traffic = traffic_profile.get("hbm_bytes_per_launch")
            else:
                traffic_profile["stale"] = "collected on other kernel sources (sha256 differs): not quoted as roofline.traffic"
        out = {
            "metric": "train-step images/sec at 192x640, D=96; cost-volume HBM GB/s vs roofline" if not a.trainer_args else
                      "train-step images/sec at %dx%d, D=%d; cost-volume HBM GB/s vs roofline" % (opt.height, opt.width, opt.num_depth_bins),
            "value": gb * a.steps / elapsed, "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            # wall time of the launching thread's loop WITHOUT the final synchronise.  Not evidence of a host limit: the HIP queue
            # back-pressures, so in a GPU-bound step (this one: 43.4 ms of kernel time per step in the trace) the thread is
            # held at the queue's depth and this equals ms_per_step too; it only says something when it is well BELOW ms_per_step
            "host_issue_ms_per_step": 1e3 * host_issue / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"none": "f32", "bf16": "bf16", "fp16": "f16"}[opt.amp], "data": "synthetic",
            "config": {"workload": ("BASELINE config %d: " % (2 if world == 1 else 3) if not a.trainer_args else "") +
                                   "KITTI %dx%d, ResNet%d, D=%d, batch %d/GPU, %s, %d-frame cost volume%s, "
                                   "process_batch+backward+Adam%s" % (
                                       opt.height, opt.width, opt.res_arch, opt.num_depth_bins, opt.batch_size,
                                       {"none": "fp32", "bf16": "bf16 autocast", "fp16": "fp16 autocast"}[opt.amp],
                                       len(opt.matching_ids), ", velocity-guided bins" if a.epoch > opt.ztrans_start_epc else "",
                                       (" [" + a.trainer_args + "]") if a.trainer_args else ""),
                       "global_batch": gb, "parallelism": "dp%d" % world, "final_loss": loss_val, "collectives": collectives},
            "roofline": {"bound": "hbm", "kernel": "md_costvol_fwd%s (plane-sweep cost volume, fused schedule + group mean)" % sfx,
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (ach / HBM_PEAK_GBS) if ach else None,
                         "traffic": traffic, "traffic_from_profile": traffic_profile, "algorithmic_bytes_per_launch": fbytes,
                         "timing": "HIP events inside libmovedepth_hip.so around the kernel launch (md_kernel_timing_*)",
                         "avg_launch_us": kt.get("avg_us"), "min_launch_us": kt.get("min_us"), "median_launch_us": kt.get("median_us"), "launches_timed": kt.get("launches"),
                         "bwd_avg_launch_us": times.get("md_costvol_bwd" + sfx, {}).get("avg_us")},
        }
        # the 3-D regulariser's first / last convolutions (hand-off either side of it), same live HIP-event timing (avg_us = the
        # kernel's own dispatch, as rocprofv3 reports it; entry_point_avg_us = events recorded from Python around the whole call):
        # 16->16 / 32->32 are MFMA-bound (2*27*Ci*Co flop per voxel; peak: mfma_peak below), 16->1 is HBM-bound
        # (the 16-channel volume read or written once plus the 1-channel one)
        vox = opt.batch_size * opt.num_depth_bins * h * w
        conv = {}
        # the 16 -> 16 / 32 -> 32 kernels multiply on the bf16 matrix pipe with three-piece operands (six bf16 products per fp32 product):
        # their bound is the dense bf16 peak / 6 in fp32-equivalent flop
        BF16_PEAK_TF = 2500.0

        def mfma_peak(name):   # (a -DMD_C16_BF3=0 A/B build runs the fp32 MFMA kernels instead: peak 157.3)
            return (BF16_PEAK_TF / 6, "dense bf16 MFMA peak / 6 (bf16 x 3 operands: six products per fp32 product)")

        for name in CONV_KERNELS:
            kc = times.get(name)
            if not kc:
                continue
            e = {"avg_us": kc["avg_us"], "launches_timed": kc["launches"], "entry_point_avg_us": entry_us.get(name)}
            if "_cb_" in name:
                # 32 -> 32 at half resolution: avg_us is ONE dispatch (forward / data gradient: one per input block = half the layer's
                # 2 * 27 * 32 * 32 flop per voxel each; weight gradient: all of it), entry_point_avg_us the whole call
                per = 2 * 27 * 32 * 32 * (vox / 8) * (1.0 if name.endswith("weight") else 0.5)
                e["bound"], e["achieved"], e["unit"] = "mfma", per / kc["avg_us"] * 1e-6, "TFLOP/s (fp32-equivalent)"
                e["peak"], e["peak_note"] = mfma_peak(name)
                e["shape"] = "reg3d.conv2: %d x 32 x %d x %d x %d" % (opt.batch_size, opt.num_depth_bins // 2, h // 2, w // 2)
            elif "c16" in name:
                e["bound"], e["achieved"], e["unit"] = "mfma", 2 * 27 * 16 * 16 * vox / kc["avg_us"] * 1e-6, "TFLOP/s (fp32-equivalent)"
                e["peak"], e["peak_note"] = mfma_peak(name)
            else:
                e["bound"], e["achieved"], e["peak"], e["unit"] = "hbm", 4 * vox * (opt.reg3d_c + 1) / kc["avg_us"] * 1e-3, HBM_PEAK_GBS, "GB/s"
            e["frac"] = e["achieved"] / e["peak"]
            conv[name] = e
        if world == 1 and not a.trainer_args and os.environ.get("MD_BENCH_PARALLAX", "1") == "1":
            out["roofline"]["parallax_cases"] = parallax_cases(opt, dev)
        if conv:
            out["reg3d_handoff_kernels"] = conv
        if photo_in_step:
            out["photometric_kernels_in_step"] = photo_in_step
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(opt)
            if not a.trainer_args:
                gpu_ms, gpu_wall_ms, per_entry = gpu_hot_path_ms_per_image(opt, dev)
                out["cpu_baseline"]["gpu_hot_path_ms_per_image"] = gpu_ms
                out["cpu_baseline"]["gpu_hot_path_wall_ms_per_image"] = gpu_wall_ms
                out["cpu_baseline"]["gpu_hot_path_kernels"] = per_entry
                out["cpu_baseline"]["gpu_hot_path_note"] = ("the same operations on the GPU (batch %d): %.3f ms of kernel time per image (sum of "
                                                            "the dispatches' own durations, host-independent; %.3f ms wall incl. launch latency) "
                                                            "against %.0f ms on one host thread" %
                                                            (opt.batch_size, gpu_ms, gpu_wall_ms, out["cpu_baseline"]["ms_per_image"]))
        # the library's bf16 / fp16 kernels printf diagnostics to stdout (C stdio, flushed at exit when stdout is a pipe):
        # push those out first so that the JSON is the last line
        import ctypes
        ctypes.CDLL(None).fflush(None)
        if opt.amp != "none":
            sys.stdout.write("\n")
        print(json.dumps(out), file=json_out, flush=True)
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
