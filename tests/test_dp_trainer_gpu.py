"""Data parallelism with the REAL trainer (SURVEY 8 row e; reference trainer.py:49, 69-135): two ranks, one process each,
both on cuda:0 (MD_SHARE_GPU=1: the GPU box has one GPU; the process group is gloo there, RCCL on a multi-GPU node -- the
same torch.distributed calls).  Each rank runs one Trainer.train_step on its own shard; checked:
  * every parameter's gradient after the step's all-reduce == the mean of the two shards' single-process gradients
    (DP semantics of the reference: each rank normalises its own loss);
  * the weights are identical on both ranks before and after the optimizer step;
  * the number of collectives in the step == the number of gradient buckets (no data-path collective).
This file runs with --sync_bn 0 (local BatchNorm statistics); the default --ddp --sync_bn 1 path is tests/test_dp_syncbn_gpu.py.
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), MD_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist

        from movedepth_amd.options import MovedepthOptions
        from movedepth_amd.synthetic import make_inputs
        from movedepth_amd.trainer import Trainer

        torch.backends.cudnn.benchmark = False
        torch.backends.cudnn.deterministic = True
        opt = MovedepthOptions().parse(["--height", "64", "--width", "128", "--num_depth_bins", "16", "--batch_size", "2",
                                        "--convex_up", "--weights_init", "scratch", "--miopen_find", "0", "--ddp", "--sync_bn", "0",
                                        "--automask_noise", "host", "--grad_bucket_mb", "8", "--learning_rate", "1e-3"])
        torch.manual_seed(50 + rank)       # different initial weights per rank: the constructor's broadcast must fix that
        np.random.seed(50 + rank)
        t = Trainer(opt)
        t.set_train()
        params = [p for m in t.models.values() for p in m.parameters()]

        def flat_w():
            return torch.cat([p.detach().flatten() for p in params]).cpu().numpy()

        def flat_g():
            return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().flatten() for p in params]).cpu().numpy()

        w_start = flat_w()
        shards = [make_inputs(2, 64, 128, opt.frame_ids, seed=200 + r, device=t.device) for r in range(world)]

        # ---- single-process gradients of BOTH shards on this rank (the reducer switched off)
        local = []
        for r in range(world):
            t.grad_sync.world = 1
            t.grad_sync.zero_grad()
            torch.manual_seed(300 + r)
            np.random.seed(300 + r)
            _, losses = t.process_batch(dict(shards[r]), is_train=True)
            losses["loss"].backward()
            local.append(flat_g())
        mean_local = sum(local) / world

        # ---- the real data-parallel step on this rank's shard, counting collectives
        t.grad_sync.world = world
        calls = {"n": 0}
        orig = dist.all_reduce

        def counting(*a, **k):
            calls["n"] += 1
            return orig(*a, **k)

        dist.all_reduce = counting
        torch.manual_seed(300 + rank)
        np.random.seed(300 + rank)
        # train_step = process_batch + backward (bucket all-reduces fire from the hooks) + finish + Adam
        t.train_step(dict(shards[rank]))
        dist.all_reduce = orig
        torch.cuda.synchronize()
        q.put((rank, w_start, mean_local, flat_g(), flat_w(), calls["n"], len(t.grad_sync.buckets), None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # surface the failure in the parent instead of a queue timeout
        import traceback
        q.put((rank, None, None, None, None, 0, 0, traceback.format_exc()))


def test_trainer_two_ranks_gradient_mean_and_weight_sync():
    world = 2
    # Two attempts at the RENDEZVOUS only: a worker that dies of an infrastructure error (port taken between _free_port() and the
    # bind, a stale process group) is started again once and its traceback printed; every comparison below runs on whatever the
    # ranks return, without retry.
    for attempt in range(2):
        port = _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda x: x[0])
        for p in procs:
            p.join(timeout=120)
        errs = [r[7] for r in res if r[7] is not None]
        if not errs or attempt == 1 or any("AssertionError" in e for e in errs):
            break
        print("rank launch failed, retrying once:\n" + "\n".join(errs))
    for r in res:
        assert r[7] is None, r[7]
    (_, w0a, ma, ga, w1a, na, nba, _), (_, w0b, mb, gb, w1b, nb, nbb, _) = res
    assert np.array_equal(w0a, w0b), "weights differ after the constructor's broadcast"

    def rel(a, b):
        return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64)))

    # both ranks hold the same reduced gradient, and it is the mean of the shards' gradients
    assert np.array_equal(ga, gb), "ranks hold different gradients after the all-reduce"
    # (each rank recomputed both shards itself; the two recomputations agree to kernel-level non-determinism)
    assert rel(ma, mb) <= 1e-4, rel(ma, mb)
    assert rel(ga, ma) <= 1e-4, ("reduced gradient vs mean of single-process gradients", rel(ga, ma))
    assert np.array_equal(w1a, w1b), "weights diverged after the optimizer step"
    assert not np.array_equal(w1a, w0a)
    # one collective per gradient bucket, nothing else (no data-path exchange; BatchNorm is local in this mode)
    assert na == nba == nb == nbb and nba >= 2, (na, nba, nb, nbb)
