"""Data-parallel plumbing on CPU with the gloo backend, world_size 2 (the GPU box uses RCCL through the same
torch.distributed calls): bucketed gradient all-reduce == mean of per-shard gradients (what the reference's DDP
wrappers compute), weights broadcast from rank 0, rank-strided synthetic shards."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from movedepth_amd.dp import GradSync, broadcast_parameters
    from movedepth_amd.synthetic import SyntheticLoader

    torch.manual_seed(100 + rank)  # different initial weights per rank: broadcast must fix that
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 3, padding=1),
                              torch.nn.Flatten(), torch.nn.Linear(4 * 8 * 8, 5))
    broadcast_parameters([net])
    w0 = torch.cat([p.detach().flatten() for p in net.parameters()])
    sync = GradSync(list(net.parameters()), bucket_mb=0.002)  # tiny buckets: several all-reduces in flight
    loader = SyntheticLoader(2, 8, 8, (0, -1, 1), steps=2, rank=rank, world_size=world)
    grads, datas = [], []
    for inputs in loader:
        x = inputs[("color", 0, 0)]
        datas.append(x.clone())
        sync.zero_grad()
        net(x).square().mean().backward()
        sync.finish()
        grads.append(torch.cat([p.grad.flatten() for p in net.parameters()]).clone())
    # reference: every rank recomputes all shards' gradients locally and averages them
    ref = []
    for step in range(2):
        acc = 0
        for r in range(world):
            l2 = SyntheticLoader(2, 8, 8, (0, -1, 1), steps=2, rank=r, world_size=world)
            x = list(l2)[step][("color", 0, 0)]
            for p in net.parameters():
                p.grad = None
            net(x).square().mean().backward()
            acc = acc + torch.cat([p.grad.flatten() for p in net.parameters()])
        ref.append(acc / world)
    # plain numpy payloads: torch tensors would travel as shared-memory handles that die with this process
    q.put((rank, w0.numpy(), [g.numpy() for g in grads], [r.numpy() for r in ref], [d.numpy() for d in datas],
           len(sync.buckets)))
    dist.destroy_process_group()


def test_gradsync_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w_a, g_a, ref_a, d_a, nb), (_, w_b, g_b, ref_b, d_b, _) = res
    assert nb > 1
    import numpy as np

    assert np.array_equal(w_a, w_b), "weights must be identical after the rank-0 broadcast"
    assert not np.array_equal(d_a[0], d_b[0]), "ranks must see different shards"
    for step in range(2):
        assert np.array_equal(g_a[step], g_b[step]), "all ranks hold the same averaged gradient"
        assert np.allclose(g_a[step], ref_a[step], rtol=1e-5, atol=1e-7), "all-reduce == mean of per-shard gradients"


def test_gradsync_single_process_is_identity():
    sys.path.insert(0, ROOT)
    from movedepth_amd.dp import GradSync

    net = torch.nn.Linear(4, 3)
    sync = GradSync(list(net.parameters()), bucket_mb=1.0)
    x = torch.randn(5, 4)
    sync.zero_grad()
    net(x).sum().backward()
    sync.finish()
    g = net.weight.grad.clone()
    net.weight.grad = None
    net.bias.grad = None
    net(x).sum().backward()
    assert torch.allclose(g, net.weight.grad)


def _worker_unused(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from movedepth_amd.dp import GradSync

    torch.manual_seed(7)   # same weights on every rank
    used, unused = torch.nn.Linear(6, 4), torch.nn.Linear(6, 4)
    params = list(used.parameters()) + list(unused.parameters())
    sync = GradSync(params, bucket_mb=1e-5)   # one bucket per tensor: some complete in the hooks, the unused ones only in finish()
    out = []
    for step in range(2):
        x = torch.full((3, 6), float(rank + 1 + step))
        sync.zero_grad()
        assert all(p.grad is None for p in params)            # nothing pre-filled: autograd assigns, the hook packs
        used(x).sum().backward()
        sync.finish()
        assert all(p.grad is not None for p in params)
        # .grad are views into the flat buckets after the step
        assert all(p.grad.data_ptr() == v.data_ptr() for vs, (_, ps) in zip(sync.views, sync.buckets) for p, v in zip(ps, vs))
        out.append([p.grad.clone().numpy() for p in params])
    q.put((rank, out, len(sync.buckets)))
    dist.destroy_process_group()


def test_gradsync_packs_buckets_and_zeroes_unused_parameters():
    """the packed-bucket reducer: gradients of parameters that took part are the mean over ranks, parameters of an unused branch get
    zeros (and still take part in the collective: every rank issues the same all-reduces), two steps in a row"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_unused, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np

    (_, ga, nb), (_, gb, _) = res
    assert nb == 4
    for step in range(2):
        for a, b in zip(ga[step], gb[step]):
            assert np.array_equal(a, b)
        # d/dW of sum(W x + b) = x summed over the 3 rows, mean over ranks: 3 * mean(rank + 1 + step) = 3 * (1.5 + step); d/db = 3
        assert np.allclose(ga[step][0], 3.0 * (1.5 + step)) and np.allclose(ga[step][1], 3.0)
        assert not ga[step][2].any() and not ga[step][3].any()


def test_direct_rccl_path_declines_without_an_nccl_group():
    """movedepth_amd/rccl_direct.py: no process group, or a group on another backend (the gloo runs of this suite), -> None and the
    callers keep torch.distributed.all_reduce; ops._group_all_reduce / _group_size dispatch on what they are given"""
    sys.path.insert(0, ROOT)
    from movedepth_amd import rccl_direct

    assert not dist.is_initialized() and rccl_direct.make(None) is None
    port = _free_port()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert rccl_direct.make(None) is None
        from movedepth_amd import ops
        t = torch.ones(3)
        ops._group_all_reduce(t, dist.group.WORLD)
        assert ops._group_size(dist.group.WORLD) == 1 and torch.equal(t, torch.ones(3))

        class Fake:                      # what a DirectAllReduce looks like to the callers
            size, seen = 4, []
            def __call__(self, x):
                self.seen.append(x)
        f = Fake()
        ops._group_all_reduce(t, f)
        assert ops._group_size(f) == 4 and f.seen and f.seen[0] is t
    finally:
        dist.destroy_process_group()


def _stage_worker(rank, world, port, q, fail):
    """rccl_direct.run_stages over gloo: stage 'local_b' raises on the rank named by `fail` ('' = nobody).  The stage behind it is a
    collective on the group: a rank must reach it only if EVERY rank passed 'local_b'."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if fail:
        os.environ["MD_DIRECT_RCCL_FAIL"] = fail
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from movedepth_amd import rccl_direct

    seen = []

    def local_a(st):
        seen.append("local_a")
        st["x"] = torch.full((3,), float(rank + 1))

    def local_b(st):
        seen.append("local_b")

    def collective_c(st):                      # would block for ever if one rank had left after local_b
        seen.append("collective_c")
        dist.all_reduce(st["x"])

    st = rccl_direct.run_stages([("local_a", local_a), ("local_b", local_b), ("collective_c", collective_c)], dist.group.WORLD, None, "test")
    # ranks that fall back keep working on the parent group: one more collective proves nobody is stuck in a stage
    after = torch.ones(1)
    dist.all_reduce(after)
    q.put((rank, st is not None, seen, st["x"].tolist() if st is not None else None, float(after.item())))
    dist.destroy_process_group()


@pytest.mark.parametrize("fail", ["", "1:local_b", "0:local_a"])
def test_staged_agreement_one_rank_failure_is_a_group_wide_fallback(fail):
    """movedepth_amd/rccl_direct.run_stages (what make() is built from): an exception on ONE rank in any stage makes EVERY rank
    return None at that stage -- no rank enters the next stage's collective alone, no rank blocks.
    (A rank that dies INSIDE a collective stage, before its collective, cannot be covered by any such protocol -- its peers are
    already in the call; make()'s collective stages therefore contain nothing but the calls every rank makes, and bench.py's
    watchdog turns what is left into a non-zero exit code.)"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stage_worker, args=(r, world, port, q, fail)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    oks = [r[1] for r in res]
    assert oks[0] == oks[1], "the ranks must agree on the outcome"
    assert all(r[4] == 2.0 for r in res), "the parent group must stay usable after a fallback"
    if not fail:
        assert oks == [True, True] and all(r[3] == [3.0, 3.0, 3.0] for r in res)
        assert all(r[2] == ["local_a", "local_b", "collective_c"] for r in res)
        return
    assert oks == [False, False]
    frank, fstage = fail.split(":")
    order = ["local_a", "local_b", "collective_c"]
    reached = order[:order.index(fstage) + 1]
    for r in res:
        # the failing rank never ran the failing stage's body; the other ran it; nobody went further
        assert r[2] == (reached[:-1] if r[0] == int(frank) else reached), (fail, r)


def _direct_bucket_worker(rank, world, port, q):
    """GradSync(direct=...): the buckets go through the callable (rccl_direct.DirectAllReduce on the GPU box; here a stand-in with
    the same interface over gloo) in the order the hooks fire, and the result equals the torch-group path's."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from movedepth_amd.dp import GradSync

    class Direct:
        size, calls, sizes = world, 0, []
        def __call__(self, t):
            self.calls += 1
            self.sizes.append(t.numel())
            dist.all_reduce(t)
            return t

    out = []
    for use_direct in (False, True):
        torch.manual_seed(7)
        net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(), torch.nn.Linear(16, 2))
        d = Direct() if use_direct else None
        sync = GradSync(list(net.parameters()), bucket_mb=0.0005, direct=d)
        x = torch.full((4, 6), float(rank + 1))
        sync.zero_grad()
        net(x).square().mean().backward()
        sync.finish()
        out.append((torch.cat([p.grad.flatten() for p in net.parameters()]).numpy().copy(), len(sync.buckets), d.calls if d else 0, len(sync._handles)))
    q.put((rank, out))
    dist.destroy_process_group()


def test_gradsync_buckets_through_a_direct_all_reduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_direct_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    for _, out in res:
        (g_t, nb, _, _), (g_d, nb_d, calls, handles) = out
        assert nb == nb_d and nb > 1 and calls == nb and handles == 0   # one direct call per bucket, nothing queued on torch's group
        assert np.allclose(g_t, g_d, rtol=1e-6, atol=1e-8)
    assert np.array_equal(res[0][1][1][0], res[1][1][1][0])                # both ranks hold the same mean gradient
