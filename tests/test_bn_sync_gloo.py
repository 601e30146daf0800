"""Two ranks sharing cuda:0 over gloo: FusedBNReLU3d with a sync group must equal the single-process result on the
concatenated batch (SyncBatchNorm semantics: global statistics, local weight-gradient sums)."""
import os
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, xs, gys, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from movedepth_amd import networks
    cl = lambda t: t.contiguous(memory_format=torch.channels_last_3d)
    m = networks.FusedBNReLU3d(16).cuda()
    m.sync_group = dist.group.WORLD
    with torch.no_grad():
        m.weight.copy_(torch.linspace(0.5, 1.5, 16)); m.bias.copy_(torch.linspace(-0.3, 0.3, 16))
    x = cl(torch.from_numpy(xs[rank]).cuda()).requires_grad_(True)
    y = m(x)
    y.backward(cl(torch.from_numpy(gys[rank]).cuda()))
    q.put((rank, y.detach().cpu().numpy(), x.grad.cpu().numpy(), m.weight.grad.cpu().numpy(), m.bias.grad.cpu().numpy(),
           m.running_mean.cpu().numpy(), m.running_var.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_fused_bn_sync_two_ranks_equals_big_batch():
    rng = np.random.default_rng(5)
    xs = [rng.standard_normal((2, 16, 4, 6, 8)).astype(np.float32) + r for r in range(2)]   # different means per rank
    gys = [rng.standard_normal((2, 16, 4, 6, 8)).astype(np.float32) for _ in range(2)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:  # a free port: the suite may share the box with other runs
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ps = [ctx.Process(target=_worker, args=(r, 2, port, xs, gys, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in ps:
        p.join(60)
    # single-process reference on the concatenated batch, torch ops
    x = torch.from_numpy(np.concatenate(xs)).cuda().requires_grad_(True)
    w = torch.linspace(0.5, 1.5, 16).cuda().requires_grad_(True); b = torch.linspace(-0.3, 0.3, 16).cuda().requires_grad_(True)
    rm, rv = torch.zeros(16).cuda(), torch.ones(16).cuda()
    y = torch.relu(torch.nn.functional.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5))
    y.backward(torch.from_numpy(np.concatenate(gys)).cuda())
    rel = lambda a, c: np.linalg.norm(a - c) / np.linalg.norm(c)
    assert rel(np.concatenate([g[1] for g in got]), y.detach().cpu().numpy()) < 1e-5
    assert rel(np.concatenate([g[2] for g in got]), x.grad.cpu().numpy()) < 1e-4
    assert rel(got[0][3] + got[1][3], w.grad.cpu().numpy()) < 1e-4      # local sums add up to the big-batch gradient
    assert rel(got[0][4] + got[1][4], b.grad.cpu().numpy()) < 1e-4
    for g in got:
        assert rel(g[5], rm.cpu().numpy()) < 1e-5 and rel(g[6], rv.cpu().numpy()) < 1e-5
