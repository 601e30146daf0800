"""The DEFAULT data-parallel path of the trainer -- `--ddp --sync_bn 1`, what `bench.py --gpus N` runs (reference
trainer.py:69-135: every sub-model through SyncBatchNorm.convert_sync_batchnorm + DistributedDataParallel) -- with two ranks of
the real Trainer, one process each, both on cuda:0 over gloo (MD_SHARE_GPU=1; RCCL on a multi-GPU node, same torch.distributed
calls).  networks.HipSyncBatchNorm (csrc/syncbn.hip; --sync_bn_impl hip, the default) or torch.nn.SyncBatchNorm (--sync_bn_impl torch) for every
BatchNorm of the five networks, networks.FusedBNReLU3d.sync_group for the two fused full-resolution layers of the regulariser,
--bn_counter_on_host left at its default.

With synchronised statistics and equal shard sizes the two-rank step IS the single-process step on the concatenated batch,
provided every rank normalises its loss over the same number of pixels: auto-masking is switched off for this test (each rank
divides by its own mask sum) and both ranks draw the same erase rectangle.  Checked against that big-batch run:
  * the gradients every rank holds after the all-reduce: 1e-4 norm-wise over all parameters, 5e-3 per sub-network (per-pixel
    arg-max / arg-min decisions can fall differently between a batch-4 and a batch-2 run);
  * BatchNorm running statistics after the step (library SyncBatchNorm layers and the fused layers);
  * weights identical on both ranks after the optimizer step;
  * the number of collectives of the step = gradient buckets + one per BatchNorm call in the forward (statistics) + one per
    BatchNorm call in the backward: nothing else talks.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

ARGV = ["--height", "64", "--width", "128", "--num_depth_bins", "16", "--convex_up", "--weights_init", "scratch",
        "--miopen_find", "0", "--automask_noise", "host", "--grad_bucket_mb", "8", "--learning_rate", "1e-3",
        "--disable_automasking"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _named(t):
    return [(mn + "." + pn, p) for mn, m in t.models.items() for pn, p in m.named_parameters()]


def _bn_buffers(t):
    out = {}
    for mn, m in t.models.items():
        for bn, b in m.named_buffers():
            if bn.endswith(("running_mean", "running_var")):
                out[mn + "." + bn] = b.detach().float().cpu().numpy().copy()
    return out


PEAK = 40.0


def _sharpen(t, peak):
    """The regulariser's last layer (-> logits over the hypotheses) scaled up, in every run alike: peaked probability volumes instead
    of the near-uniform ones of random-init networks (fewer arg-max ties; not none, see the test's docstring)."""
    if peak:
        with torch.no_grad():
            t.models["reg3d"].prob.weight.mul_(peak)


def _worker(rank, world, port, q, impl="hip", peak=0.0):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), MD_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist

        from movedepth_amd import networks
        from movedepth_amd.options import MovedepthOptions
        from movedepth_amd.synthetic import make_inputs
        from movedepth_amd.trainer import Trainer

        torch.backends.cudnn.benchmark = False
        torch.backends.cudnn.deterministic = True
        opt = MovedepthOptions().parse(ARGV + ["--batch_size", "2", "--ddp", "--sync_bn_impl", impl])     # --sync_bn 1 is the default
        torch.manual_seed(50 + rank)
        np.random.seed(50 + rank)
        t = Trainer(opt)
        _sharpen(t, peak)
        t.set_train()
        sync_layers = [m for net in t.models.values() for m in net.modules()
                       if isinstance(m, torch.nn.SyncBatchNorm) or
                       (isinstance(m, (networks.FusedBNReLU3d, networks.HipSyncBatchNorm)) and m.sync_group is not None)]
        assert any(isinstance(m, networks.HipSyncBatchNorm) for m in sync_layers) == (impl == "hip")
        n_plain = sum(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and not isinstance(m, torch.nn.SyncBatchNorm)
                      for net in t.models.values() for m in net.modules())
        bn_calls = {"n": 0}
        for m in sync_layers:
            m.register_forward_hook(lambda *_: bn_calls.__setitem__("n", bn_calls["n"] + 1))
        counts = {"all_reduce": 0, "all_gather": 0}
        orig = {k: getattr(dist, k) for k in ("all_reduce", "all_gather", "all_gather_into_tensor")}

        def wrap(name, key):
            def f(*a, **k):
                counts[key] += 1
                return orig[name](*a, **k)
            return f

        dist.all_reduce, dist.all_gather = wrap("all_reduce", "all_reduce"), wrap("all_gather", "all_gather")
        dist.all_gather_into_tensor = wrap("all_gather_into_tensor", "all_gather")
        shard = make_inputs(2, 64, 128, opt.frame_ids, seed=200 + rank, device=t.device)
        torch.manual_seed(300)
        np.random.seed(300)          # the same erase rectangle on every rank and in the big-batch run
        t.train_step(dict(shard))
        torch.cuda.synchronize()
        for k, v in orig.items():
            setattr(dist, k, v)
        grads = {n: (p.grad.detach().float().cpu().numpy().copy() if p.grad is not None else None) for n, p in _named(t)}
        weights = np.concatenate([p.detach().float().cpu().numpy().ravel() for _, p in _named(t)])
        counters_on_host = all(m.num_batches_tracked.device.type == "cpu" for m in sync_layers)
        q.put((rank, grads, _bn_buffers(t), weights, dict(counts), bn_calls["n"], len(t.grad_sync.buckets), len(sync_layers), n_plain,
               counters_on_host, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank,) + (None,) * 9 + (traceback.format_exc(),))


def _big_batch(peak=0.0, impl="torch"):
    """The same step in one process on the concatenated batch, plain BatchNorm: what synchronised statistics must reproduce."""
    sys.path.insert(0, ROOT)
    from movedepth_amd.options import MovedepthOptions
    from movedepth_amd.synthetic import make_inputs
    from movedepth_amd.trainer import Trainer

    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
    # impl hip: the big batch through the SAME normalisation kernels (--force_sync_bn: a group of one).  Two correct BatchNorm
    # implementations differ by ~1e-7 per layer, which is enough to tip per-pixel decisions further down (seen: the two-rank hip run
    # against the library-BatchNorm big batch at 5e-3 on the MVS encoder, every layer of it within 2e-7 of a float64 shadow of
    # itself, tools/diag/syncbn_trainer_shadow.py); with the same kernels on both sides what is left is exactly what this test is
    # about: the statistics exchange and the gradient reduction.
    extra = ["--force_sync_bn", "1", "--sync_bn_impl", "hip"] if impl == "hip" else []
    opt = MovedepthOptions().parse(ARGV + ["--batch_size", "4"] + extra)
    torch.manual_seed(50)            # rank 0's initial weights (the constructor's broadcast gives them to every rank)
    np.random.seed(50)
    t = Trainer(opt)
    _sharpen(t, peak)
    t.set_train()
    shards = [make_inputs(2, 64, 128, opt.frame_ids, seed=200 + r, device=t.device) for r in range(2)]
    batch = {k: torch.cat([s[k] for s in shards], 0) for k in shards[0]}
    torch.manual_seed(300)
    np.random.seed(300)
    t.train_step(batch)
    torch.cuda.synchronize()
    grads = {n: (p.grad.detach().float().cpu().numpy().copy() if p.grad is not None else None) for n, p in _named(t)}
    return grads, _bn_buffers(t)


@pytest.mark.parametrize("impl,peak", [("hip", 0.0), ("torch", 0.0), ("hip", PEAK)])
def test_default_ddp_path_with_synchronised_batchnorm_equals_the_big_batch_step(impl, peak):
    """impl = hip: networks.HipSyncBatchNorm on the kernels of csrc/syncbn.hip (the default); torch: torch.nn.SyncBatchNorm.
    Bounds.  The training step is not a continuous function of its weights: localmax's arg-max over the probability volume and the
    min over frames are per-pixel decisions.  Measured (tools/diag/step_sensitivity.py): scaling ONE weight tensor by 1 + 1e-7 moves
    the MVS encoder's gradient by 3.5e-3, the regulariser's by 1.6e-3 and, with peaked volumes (_sharpen), the up-sampling head's by
    1.5e-3 -- in a single process, no data parallelism involved.  Two correct runs that differ in a summation order are that far
    apart too, so the end-to-end comparison is held to 2e-2 per sub-network / 5e-3 overall (a wrong or missing statistics exchange
    moves EVERY gradient by tens of per cent); the tight check of the exchange itself is layer by layer, in
    test_every_batchnorm_call_of_a_two_rank_step_matches_its_float64_shadow below."""
    world = 2
    # Two attempts at the RENDEZVOUS only: a worker that dies of an infrastructure error (port taken between _free_port() and the
    # bind, a stale process group) is started again once and its traceback printed; every comparison below runs on whatever the
    # ranks return, without retry.
    for attempt in range(2):
        port = _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q, impl, peak)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda x: x[0])
        for p in procs:
            p.join(timeout=120)
        errs = [r[10] for r in res if r[10] is not None]
        if not errs or attempt == 1 or any("AssertionError" in e for e in errs):
            break
        print("rank launch failed, retrying once:\n" + "\n".join(errs))
    for r in res:
        assert r[10] is None, r[10]
    (_, ga, bna, wa, ca, calls_a, nb, nsync, nplain, host_a, _), (_, gb, bnb, wb, cb, calls_b, _, _, _, host_b, _) = res
    assert nplain == 0 and nsync >= 60, (nplain, nsync)                 # every BatchNorm layer of the networks is synchronised
    assert host_a and host_b                                             # --bn_counter_on_host survived the conversion
    assert np.array_equal(wa, wb), "weights differ between the ranks after the step"
    for n in ga:
        assert (ga[n] is None) == (gb[n] is None) and (ga[n] is None or np.array_equal(ga[n], gb[n])), n
    # collectives: one per gradient bucket, one per BatchNorm call forward (statistics) and one per call backward
    assert calls_a == calls_b and calls_a >= 100, calls_a                # 115 BatchNorm calls per step (shared encoders run 2-4 times)
    assert ca == cb, (ca, cb)
    assert ca["all_reduce"] + ca["all_gather"] == nb + 2 * calls_a, (ca, nb, calls_a)

    want_g, want_bn = _big_batch(peak, impl)

    def rel(a, b):
        return float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b.astype(np.float64)) + 1e-30))

    worst = []
    gmax = max(float(np.abs(v).max()) for v in want_g.values() if v is not None)
    for n, want in want_g.items():
        if want is None:
            assert ga[n] is None or float(np.abs(ga[n]).max()) == 0.0, n
            continue
        if float(np.abs(want).max()) < 1e-6 * gmax:
            continue                      # a gradient that is itself rounding noise (e.g. a bias in front of a BatchNorm)
        worst.append((rel(ga[n], want), n))
    worst.sort(reverse=True)
    print("worst parameter gradients vs the big-batch step:", [(n, "%.1e" % r) for r, n in worst[:5]])
    total = rel(np.concatenate([ga[n].ravel() for _, n in worst]), np.concatenate([want_g[n].ravel() for _, n in worst]))
    print("all gradients, norm-wise: %.2e" % total)
    by_model = {}
    for r_, n in worst:
        by_model.setdefault(n.split(".")[0], []).append(n)
    per_model = {m_: rel(np.concatenate([ga[n].ravel() for n in ns]), np.concatenate([want_g[n].ravel() for n in ns])) for m_, ns in by_model.items()}
    print("per sub-network:", {k: "%.2e" % v for k, v in per_model.items()})
    for r_, n in worst[:3]:
        if ga[n].size <= 64:
            print("  ", n, "\n     two ranks:", np.array2string(ga[n].ravel(), precision=5), "\n     big batch:", np.array2string(want_g[n].ravel(), precision=5))
    assert total <= 5e-3, (total, per_model)
    # Per sub-network (see the docstring for the bounds)
    by_model = {}
    for r_, n in worst:
        by_model.setdefault(n.split(".")[0], []).append(n)
    for mname, names in by_model.items():
        r_m = rel(np.concatenate([ga[n].ravel() for n in names]), np.concatenate([want_g[n].ravel() for n in names]))
        print("  %-14s %.2e" % (mname, r_m))
        assert r_m <= 2e-2, (mname, r_m)
    bn_worst = max((rel(bna[k], want_bn[k]), k) for k in want_bn)
    print("worst BatchNorm running statistic:", bn_worst)
    assert bn_worst[0] <= 1e-4, bn_worst


def _shadow_worker(rank, world, port, q):
    """Every HipSyncBatchNorm call of one two-rank training step, shadowed by the same layer written with torch ops in float64 and
    an all-reduce of its own: output, input gradient, and the layer's parameter gradients after the gradient reducer."""
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                          MD_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist

        from movedepth_amd import networks
        from movedepth_amd.options import MovedepthOptions
        from movedepth_amd.synthetic import make_inputs
        from movedepth_amd.trainer import Trainer

        torch.backends.cudnn.benchmark = False
        torch.backends.cudnn.deterministic = True
        opt = MovedepthOptions().parse(ARGV + ["--batch_size", "2", "--ddp"])
        torch.manual_seed(50 + rank)
        np.random.seed(50 + rank)
        t = Trainer(opt)
        t.set_train()
        report, pgrads = [], {}
        names = {m: mn + "." + n for mn, net in t.models.items() for n, m in net.named_modules() if isinstance(m, networks.HipSyncBatchNorm)}
        rel = lambda a, b: float((a.double() - b).norm() / (b.norm() + 1e-300))

        def fwd_hook(mod, inp, out):
            x = inp[0].detach()
            dims = [0] + list(range(2, x.dim()))
            xd = x.double()
            s = torch.stack([xd.sum(dims), (xd * xd).sum(dims)])
            dist.all_reduce(s)
            n = x.numel() // x.shape[1] * world
            mean, var = s[0] / n, s[1] / n - (s[0] / n) ** 2
            shp = [1, -1] + [1] * (x.dim() - 2)
            invstd = 1.0 / torch.sqrt(var + mod.eps)
            xh = (xd - mean.view(shp)) * invstd.view(shp)
            z = xh * mod.weight.detach().double().view(shp) + mod.bias.detach().double().view(shp)
            rec = {"name": names[mod], "fwd": rel(out.detach(), torch.relu(z) if mod.relu else z)}
            report.append(rec)
            if out.requires_grad and inp[0].requires_grad:
                store = {}
                live = (out.detach() > 0) if mod.relu else None   # the layer's own ReLU decisions (a pre-activation within float
                                                                   # rounding of zero may legitimately fall either way)

                def on_dy(g):
                    store["dy"] = g.detach().double()
                    return g

                def on_dx(g):
                    dz = store["dy"] * live if mod.relu else store["dy"]
                    r = torch.stack([dz.sum(dims), (dz * xh).sum(dims)])
                    acc = pgrads.setdefault(names[mod], torch.zeros_like(r))
                    acc += r
                    dist.all_reduce(r)
                    gi = (mod.weight.detach().double() * invstd).view(shp)
                    rec["bwd"] = rel(g.detach(), gi * (dz - (r[0] / n).view(shp) - xh * (r[1] / n).view(shp)))
                    return g

                out.register_hook(on_dy)
                inp[0].register_hook(on_dx)

        for m in names:
            m.register_forward_hook(fwd_hook)
        shard = make_inputs(2, 64, 128, opt.frame_ids, seed=200 + rank, device=t.device)
        torch.manual_seed(300)
        np.random.seed(300)
        t.train_step(dict(shard))
        torch.cuda.synchronize()
        perr = {}
        for mod, nm in names.items():
            if nm in pgrads:
                ref = pgrads[nm].clone()
                dist.all_reduce(ref)
                ref /= world                                    # the reducer's mean over the ranks of the per-rank sums
                perr[nm] = (rel(mod.bias.grad, ref[0]), rel(mod.weight.grad, ref[1]))
        q.put((rank, report, perr, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, None, None, traceback.format_exc()))


def test_every_batchnorm_call_of_a_two_rank_step_matches_its_float64_shadow():
    """The deterministic check of the statistics exchange (the end-to-end comparison above cannot be tight, see its docstring):
    inside a real two-rank step every one of the ~110 HipSyncBatchNorm calls -- forward AND backward -- is compared with the same
    layer in float64 torch ops with its own all-reduce: output 1e-5, input gradient 1e-4, and the gradients of the layer's weight /
    bias after the gradient reducer against the mean over the ranks of the shadow's per-rank sums, 1e-4.  A mis-scaled, missing or
    mis-ordered all-reduce of ANY single layer fails it."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shadow_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
    for rank, report, perr, err in res:
        assert err is None, err
        assert len(report) >= 100 and sum("bwd" in r for r in report) >= 100, len(report)
        worst_f = max(report, key=lambda r: r["fwd"])
        worst_b = max((r for r in report if "bwd" in r), key=lambda r: r["bwd"])
        worst_p = max(perr.items(), key=lambda kv: max(kv[1]))
        print("rank %d: %d calls; worst forward %s %.1e, worst input gradient %s %.1e, worst parameter gradient %s %s" % (
            rank, len(report), worst_f["name"], worst_f["fwd"], worst_b["name"], worst_b["bwd"], worst_p[0], ["%.1e" % v for v in worst_p[1]]))
        assert worst_f["fwd"] <= 1e-5, worst_f
        assert worst_b["bwd"] <= 1e-4, worst_b
        assert max(worst_p[1]) <= 1e-4, worst_p
