"""The trainer's `--ddp --sync_bn 1` path on the backend the multi-GPU runs use: torch.distributed "nccl" (= RCCL on ROCm), here
with a group of ONE rank -- all a one-GPU box can host.  Every collective of a step goes through RCCL: the constructor's weight
broadcast and the gradient buckets' all-reduces (launched from autograd hooks) on torch's communicator stream, ordered against the
kernels by ProcessGroupNCCL's stream events; the 2C-double / 2C-float sums of every synchronised BatchNorm call, forward and
backward, as direct ncclAllReduce calls on the compute stream (movedepth_amd/rccl_direct.py).  The gloo tests
(tests/test_dp_syncbn_gpu.py) cannot see a mistake there: gloo collectives on GPU tensors are synchronous host copies.
A group of one makes every reduction the identity, so the step must reproduce the non-distributed step with the same normalisation
kernels (--force_sync_bn 1): loss, gradients, BatchNorm running statistics.  Not bit for bit -- the plane sweep's d_src sums are
float atomics between neighbouring tiles, and the step amplifies a 1e-7 difference to 1e-3 in some sub-networks
(tools/diag/step_sensitivity.py) -- so the bounds are those of tests/test_dp_syncbn_gpu.py; a collective reading its buffer before
the kernel in front of it has written it misses them by orders of magnitude.
Reference: trainer.py:49, 69-135."""
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
        "--miopen_find", "0", "--automask_noise", "host", "--grad_bucket_mb", "8", "--learning_rate", "1e-3", "--batch_size", "2"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _state(t):
    named = [(mn + "." + pn, p) for mn, m in t.models.items() for pn, p in m.named_parameters()]
    grads = {n: (p.grad.detach().cpu().numpy().copy() if p.grad is not None else None) for n, p in named}
    weights = {n: p.detach().cpu().numpy().copy() for n, p in named}
    stats = {mn + "." + bn: b.detach().cpu().numpy().copy() for mn, m in t.models.items() for bn, b in m.named_buffers()
             if bn.endswith(("running_mean", "running_var"))}
    return grads, weights, stats


def _run(ddp, steps=1):
    sys.path.insert(0, ROOT)
    from movedepth_amd.options import MovedepthOptions
    from movedepth_amd.synthetic import make_inputs
    from movedepth_amd.trainer import Trainer

    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
    opt = MovedepthOptions().parse(ARGV + (["--ddp"] if ddp else ["--force_sync_bn", "1"]))
    torch.manual_seed(50)
    np.random.seed(50)
    t = Trainer(opt)
    t.set_train()
    inputs = make_inputs(2, 64, 128, opt.frame_ids, seed=200, device=t.device)
    torch.manual_seed(300)
    np.random.seed(300)
    losses = []
    for _ in range(steps):
        _, l = t.train_step(dict(inputs))
        losses.append(float(l["loss"].detach()))
    torch.cuda.synchronize()
    return t, losses


def _worker(port, q, env):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                          HSA_ENABLE_IPC_MODE_LEGACY="0", MD_DP_FORCE_COLLECTIVES="1")
        for k in ("MD_SHARE_GPU", "MD_DIRECT_RCCL", "MD_DIRECT_RCCL_FAIL"):
            os.environ.pop(k, None)
        os.environ.update(env)
        import torch.distributed as dist

        counts = {"all_reduce": 0, "broadcast": 0}
        orig = {k: getattr(dist, k) for k in counts}
        for k in counts:
            def f(*a, _k=k, **kw):
                counts[_k] += 1
                return orig[_k](*a, **kw)
            setattr(dist, k, f)
        t, losses = _run(ddp=True)
        backend = dist.get_backend()
        from movedepth_amd import networks
        n_sync = sum(isinstance(m, networks.HipSyncBatchNorm) and m.sync_group is not None for net in t.models.values() for m in net.modules())
        direct = t.bn_group.calls if callable(t.bn_group) else -1
        q.put((_state(t), losses, backend, dict(counts), len(t.grad_sync.buckets), n_sync, direct, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((None, None, None, None, None, None, None, traceback.format_exc()))


_PLAIN = []


def _plain():
    if not _PLAIN:
        t, want_losses = _run(ddp=False)
        _PLAIN.append((_state(t), want_losses))
    return _PLAIN[0]


@pytest.mark.parametrize("mode", ["torch_group", "direct", "direct_failing_stage"])
def test_ddp_step_over_a_single_rank_rccl_group_equals_the_plain_step(mode):
    """torch_group: the default -- every collective through torch.distributed's group.  direct: MD_DIRECT_RCCL=1 -- the BatchNorm
    statistics AND the gradient buckets as ncclAllReduce on the compute stream through one communicator.  direct_failing_stage: the
    same with an exception injected into the stage of rccl_direct.make() that takes the communicator handle: the ranks agree to
    fall back (nothing blocks) and the step runs through torch's group."""
    env = {"torch_group": {}, "direct": {"MD_DIRECT_RCCL": "1"},
           "direct_failing_stage": {"MD_DIRECT_RCCL": "1", "MD_DIRECT_RCCL_FAIL": "0:handle"}}[mode]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q, env))
    p.start()
    state, losses, backend, counts, n_buckets, n_sync, direct, err = q.get(timeout=900)
    p.join(timeout=120)
    assert err is None, err
    assert backend == "nccl"
    assert n_sync >= 60                                       # every BatchNorm of the five networks talks to the group
    # one all-reduce per gradient bucket and two per BatchNorm call (>= 100 calls per step: shared encoders run 2-4 times)
    assert n_buckets >= 2
    if mode == "direct":
        assert direct >= 200 + n_buckets, (direct, n_buckets)   # statistics and buckets straight to RCCL on the compute stream
        assert counts["all_reduce"] <= 8, counts                 # torch's group only carried make()'s flag exchanges and its probe
    else:
        assert direct == -1, direct                              # no DirectAllReduce in use
        assert counts["all_reduce"] >= n_buckets + 200, (counts, n_buckets)
    assert counts["broadcast"] >= 100                          # the constructor's weight synchronisation, one call per tensor

    want, want_losses = _plain()
    assert abs(losses[0] - want_losses[0]) <= 1e-5 * abs(want_losses[0]), (losses, want_losses)

    def rel(keys, a, b):
        num = sum(float(np.sum((a[k].astype(np.float64) - b[k]) ** 2)) for k in keys)
        den = sum(float(np.sum(b[k].astype(np.float64) ** 2)) for k in keys)
        return (num / max(den, 1e-300)) ** 0.5

    g, wg = state[0], want[0]
    assert g.keys() == wg.keys()
    keys = [k for k in wg if wg[k] is not None]
    assert all(g[k] is not None for k in keys) and all(np.isfinite(g[k]).all() for k in keys)
    assert rel(keys, g, wg) <= 5e-3, rel(keys, g, wg)
    for net in sorted({k.split(".")[0] for k in keys}):
        sub = [k for k in keys if k.split(".")[0] == net]
        assert rel(sub, g, wg) <= 2e-2, (net, rel(sub, g, wg))
    assert state[2].keys() == want[2].keys()
    assert rel(list(want[2]), state[2], want[2]) <= 1e-4
