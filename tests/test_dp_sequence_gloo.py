"""The whole --ddp collective sequence of one training step on FOUR ranks over gloo: the eight sub-models of
the trainer (trainer.build_models: two ResNet-18 encoders, depth / pose decoders, FPN4, the 3-D regulariser, the mask network, the
convex up-sampler) converted to synchronised BatchNorm the way Trainer.__init__ converts them, called in process_batch's order
(reference trainer.py:297-442, :445-468), one backward through dp.GradSync -- every collective of the step (one all-reduce of 2C
sums per BatchNorm call and direction, one per gradient bucket) through ONE DirectAllReduce-shaped object, as with MD_DIRECT_RCCL=1,
or through torch's group, the default.  Asserted: every rank issues the identical sequence (index, element count, dtype), i.e. the
step has one total order of collectives on every rank -- what RCCL needs not to hang -- and the gradients come out as the mean over
the ranks.  No GPU here: the plane sweep, the losses and the BatchNorm KERNELS are replaced by torch stand-ins of the same data
flow (the collectives, their sizes and the autograd graph that orders them are the product's)."""
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


class RecordingAllReduce:
    """rccl_direct.DirectAllReduce's shape -- callable(tensor) in-place sum, .size, .group, .calls -- over gloo, with a log."""

    def __init__(self, group=None, kind=None):
        self.group = group
        self.size = dist.get_world_size(group)
        self.calls = 0
        self.log = []
        self.kind = kind if kind is not None else ["bn"]     # what the caller says it is reducing (set around GradSync._reduce)

    def __call__(self, t):
        assert t.is_contiguous()
        self.log.append((self.calls, self.kind[0], t.numel(), str(t.dtype)))
        dist.all_reduce(t, group=self.group)
        self.calls += 1
        return t


def _install_cpu_batchnorm(ops, networks):
    """torch stand-ins for the synchronised-BatchNorm kernels (csrc/syncbn.hip, bnrelu3d.hip): the same sums, the same ONE
    all-reduce per call and direction through ops._group_all_reduce, the same autograd structure."""
    class CpuSyncBN(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, group):
            C = x.shape[1]
            dims = [d for d in range(x.dim()) if d != 1]
            rows = x.numel() // C
            sums = torch.cat([x.double().sum(dims), (x.double() ** 2).sum(dims)])      # 2C doubles, as the kernel's
            n = rows
            if group is not None:
                ops._group_all_reduce(sums, group)
                n = rows * ops._group_size(group)
            mean = (sums[:C] / n).float()
            var = (sums[C:] / n).float() - mean ** 2
            invstd = (var + eps).rsqrt()
            shape = [1, C] + [1] * (x.dim() - 2)
            xh = (x - mean.view(shape)) * invstd.view(shape)
            y = xh * weight.view(shape) + bias.view(shape)
            if relu:
                y = y.clamp_min(0)
            ctx.save_for_backward(xh, invstd, weight, y)
            ctx.group, ctx.relu, ctx.n = group, relu, n
            return y

        @staticmethod
        def backward(ctx, dy):
            xh, invstd, weight, y = ctx.saved_tensors
            C = xh.shape[1]
            dims = [d for d in range(xh.dim()) if d != 1]
            shape = [1, C] + [1] * (xh.dim() - 2)
            if ctx.relu:
                dy = dy * (y > 0)
            sums = torch.cat([dy.sum(dims), (dy * xh).sum(dims)]).float()            # 2C floats: d_beta, d_gamma of this rank
            d_beta, d_gamma = sums[:C].clone(), sums[C:].clone()
            if ctx.group is not None:
                g = sums.clone()
                ops._group_all_reduce(g, ctx.group)
                sums = g
            dx = (dy - sums[:C].view(shape) / ctx.n - xh * sums[C:].view(shape) / ctx.n) * (weight * invstd).view(shape)
            return dx, d_gamma, d_beta, None, None, None, None, None, None

    ops._SyncBatchNorm = CpuSyncBN
    ops.sync_batch_norm_supported = lambda x: x.shape[1] % 4 == 0

    def fused_forward(self, x, res=None):       # FusedBNReLU3d on the CPU: BatchNorm + ReLU (+ skip) through the same stand-in
        if self.training:
            self.num_batches_tracked.add_(1)
        y = CpuSyncBN.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.momentum, self.eps, 1, self.sync_group)
        return y if res is None else y + res

    networks.FusedBNReLU3d.forward = fused_forward


def _step_worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from movedepth_amd import networks, ops
    from movedepth_amd.dp import GradSync, broadcast_parameters
    from movedepth_amd.options import MovedepthOptions
    from movedepth_amd.trainer import build_models

    _install_cpu_batchnorm(ops, networks)
    opt = MovedepthOptions().parse(["--height", "64", "--width", "128", "--num_depth_bins", "8", "--batch_size", "1", "--convex_up",
                                    "--weights_init", "scratch", "--ddp"])
    torch.manual_seed(100 + rank)                 # different weights per rank until the broadcast
    models, main, mvs = build_models(opt, 2)
    kind = ["bn"]
    rec = RecordingAllReduce(None, kind)
    log = rec.log
    if mode == "direct":                          # MD_DIRECT_RCCL=1: BatchNorm statistics and buckets through one object
        bn_group, direct = rec, rec
    else:                                         # the default: everything on torch's group; its calls are recorded at the API
        bn_group, direct = dist.group.WORLD, None
        orig_all_reduce = dist.all_reduce

        def logged_all_reduce(t, *a, **k):
            log.append((len(log), kind[0], t.numel(), str(t.dtype)))
            return orig_all_reduce(t, *a, **k)
        dist.all_reduce = logged_all_reduce
    for k in list(models):                        # Trainer.__init__'s conversion (trainer.py here: :118-143)
        models[k] = networks.convert_hip_sync_batchnorm(models[k], bn_group, fuse_relu=True)
        for mod in models[k].modules():
            if isinstance(mod, networks.FusedBNReLU3d):
                mod.sync_group = bn_group
        models[k].train()
    leftovers = [type(m).__name__ for net in models.values() for m in net.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    broadcast_parameters(models.values())
    n_start = len(log)
    params = [p for k in main for p in models[k].parameters()] + [p for k in mvs for p in models[k].parameters()]
    sync = GradSync(params, bucket_mb=8.0, direct=direct)
    reduce_bucket = sync._reduce

    def tagged_reduce(bi):
        kind[0] = "bucket"
        try:
            reduce_bucket(bi)
        finally:
            kind[0] = "bn"
    sync._reduce = tagged_reduce
    bn_calls = [0]
    for net in models.values():
        for mod in net.modules():
            if isinstance(mod, (networks.HipSyncBatchNorm, networks.FusedBNReLU3d)):
                mod.register_forward_hook(lambda *_: bn_calls.__setitem__(0, bn_calls[0] + 1))

    # ---- one step in process_batch's order, this rank's own shard of the batch
    g = torch.Generator().manual_seed(7 + rank)
    B, H, W, D = 1, opt.height, opt.width, opt.num_depth_bins
    img = {f: torch.rand(B, 3, H, W, generator=g) for f in opt.frame_ids}
    sync.zero_grad()
    loss = 0.0
    for f in opt.frame_ids[1:]:                   # predict_poses
        pair = [img[f], img[0]] if f < 0 else [img[0], img[f]]
        axisangle, translation = models["pose"]([models["pose_encoder"](torch.cat(pair, 1))])
        loss = loss + axisangle.square().mean() + translation.square().mean()
    ref_match, ref_ctx = models["mvs_encoder"](img[0])
    src_match = [models["mvs_encoder"](img[f])[0] for f in opt.matching_ids[1:]]
    disps = models["mono_depth"](models["mono_encoder"](img[0]), no_match=False)
    loss = loss + sum(v.mean() for k, v in disps.items() if k[0] == "disp")

    def mvs_branch(ref_feat):                     # stand-in for plane sweep + fusion: a (B,16,D,h,w) volume of the two feature maps
        Bc, C, h, w = ref_feat.shape
        prod = (ref_feat * src_match[0]).reshape(Bc, C // 16, 16, h, w).mean(1)
        vol = prod[:, None] * torch.linspace(0.5, 1.5, D).view(1, D, 1, 1, 1)    # logical (B,D,G,h,w), as ops.fuse_volumes returns it
        logits = models["reg3d"](vol.contiguous())                   # B D h w
        prob = logits.softmax(1)
        depth = (prob * torch.arange(D).view(1, D, 1, 1)).sum(1)
        ent = -(prob * (prob + 1e-6).log()).sum(1, keepdim=True)
        return depth, ent
    depth_mvs, ent = mvs_branch(ref_match)
    trust = models["mask_cnn"](ent)
    masked = img[0].clone()
    masked[:, :, 8:24, 16:48] = 0
    depth_aug, _ = mvs_branch(models["mvs_encoder"](masked)[0])
    up = models["up"](depth_mvs, ref_ctx)
    loss = loss + trust.mean() + (depth_aug - depth_mvs).abs().mean() + up.mean()
    n_fwd = len(log)
    loss.backward()
    sync.finish()
    n_end = len(log)

    bucket_sizes = sorted(flat.numel() for flat, _ in sync.buckets)
    step_log = [(i - n_start, k, n, dt) for i, k, n, dt in log[n_start:]]
    grads = torch.cat([p.grad.flatten() for p in params])
    # the mean over the ranks of what every rank would compute alone is not available without the other shards; what must hold on
    # every rank is that all of them now hold the SAME gradient vector: compared by the parent through a checksum and a lattice
    q.put((rank, step_log, n_fwd - n_start, n_end - n_fwd, bn_calls[0], bucket_sizes, leftovers,
           float(grads.double().sum()), grads[:: max(1, grads.numel() // 997)].numpy().copy(), bool(torch.isfinite(grads).all())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["direct", "torch_group"])
def test_whole_step_collective_sequence_four_ranks(mode):
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_step_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    res = []
    for _ in range(6000):
        try:
            res.append(q.get(timeout=0.1))
        except queue.Empty:
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), "a rank died: %s" % [p.exitcode for p in procs]
        if len(res) == world:
            break
    assert len(res) == world
    res.sort(key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    import numpy as np
    _, log0, nf0, nb0, bn0, buckets0, left0, sum0, lat0, fin0 = res[0]
    assert left0 == [], left0                       # every BatchNorm of the shipped networks runs on the synchronised kernels' path
    assert bn0 == 115, bn0                          # BatchNorm calls per step at the default frame set (115 + 115 collectives)
    assert nf0 == bn0                               # forward: exactly one all-reduce per BatchNorm call (2C sums)
    nbuckets = len(buckets0)
    assert nbuckets > 1 and nb0 == bn0 + nbuckets   # backward: one per BatchNorm call + one per gradient bucket
    assert all(k == "bn" and n <= 4096 and n % 8 == 0 for _, k, n, _ in log0[:nf0])        # 2C sums, C a multiple of 4
    bwd = log0[nf0:]
    assert sorted(n for _, k, n, _ in bwd if k == "bucket") == buckets0     # each bucket exactly once, interleaved with the layers'
    assert sum(k == "bn" for _, k, _, _ in bwd) == bn0
    first_bucket = next(i for i, e in enumerate(bwd) if e[1] == "bucket")
    last_bn = max(i for i, e in enumerate(bwd) if e[1] == "bn")
    assert first_bucket < last_bn                   # a bucket leaves BEFORE the backward's last BatchNorm collective: overlap is real
    for rank, log, nf, nb, bn, buckets, left, s, lat, fin in res[1:]:
        assert log == log0, "rank %d issued a different collective sequence than rank 0" % rank
        assert (nf, nb, bn, buckets) == (nf0, nb0, bn0, buckets0)
        assert fin and fin0
        assert np.array_equal(lat, lat0) and s == sum0, "rank %d holds a different averaged gradient" % rank
    print("%s: %d collectives per step and rank: %d BatchNorm forward + %d backward + %d buckets of %s elements"
          % (mode, len(log0), nf0, nb0 - nbuckets, nbuckets, buckets0))
