"""Data parallelism: one process per GPU, gradients averaged with bucketed all-reduce over RCCL/xGMI
(torch.distributed backend "nccl" is RCCL on ROCm).

The reference wraps each of its 8 sub-models in its own DistributedDataParallel (trainer.py:133-135), i.e. 8
reducers.  Here ONE reducer owns every parameter: gradients are views into a few large flat buckets
(so autograd accumulates straight into the communication buffers: no copies), and a bucket's all-reduce is
launched from a post-accumulate hook the moment its last gradient lands, overlapping with the rest of
backward.  xGMI is point-to-point (7 links/GPU): ring collectives are per-link bound, so the 112.9 MB of fp32
gradients go out as a few ~32 MB buckets rather than DDP's default 25 MB x 8 wrappers.
The hot path itself has no exchange step: every kernel is per-sample.
"""
import os

import torch
import torch.distributed as dist


def _force():
    """MD_DP_FORCE_COLLECTIVES=1 (tests on a one-GPU box): a group of one rank still issues every collective, so that the calls, their
    dtypes and their ordering against the kernels run through RCCL (tests/test_dp_rccl_single_rank_gpu.py)."""
    return os.environ.get("MD_DP_FORCE_COLLECTIVES", "0") == "1"


class GradSync:
    """One reducer for every parameter.  Gradients are left where autograd produces them (p.grad = None before a backward: the
    engine then hands each parameter its gradient tensor without a kernel); the moment the last gradient of a bucket has arrived, a
    post-accumulate hook packs the bucket's gradients into its flat buffer with ONE multi-tensor copy, re-points the parameters'
    .grad at views of that buffer and starts the bucket's all-reduce, which overlaps the rest of backward.  (First version:
    .grad were views of pre-zeroed buffers from the start, so autograd ADDED into them -- one add kernel per parameter and
    backward, 226 launches = 0.70 ms of a 44 ms step, plus 113 MB of zero fill; profiles/r04_ddp_one_rank.txt.)"""

    def __init__(self, params, bucket_mb=32.0, process_group=None, direct=None):
        self.pg = process_group
        # rccl_direct.DirectAllReduce (opt-in): the buckets go through the SAME communicator and stream as the BatchNorm statistics
        # -- ncclAllReduce on the compute stream, in the program order of the thread that runs backward -- instead of torch's
        # group and its stream.  One communicator, one stream: the collectives of a step have one total order on every rank
        # (two communicators on two streams is the documented RCCL hang).  The price is that a bucket's all-reduce no longer
        # overlaps the backward kernels behind it (4 x 28 MB per step: well under a millisecond over xGMI).
        self.direct = direct
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # reverse order ~ order in which backward produces gradients
        self.params = [p for p in reversed(list(params)) if p.requires_grad]
        self.buckets = []  # (flat tensor, [params])
        self.views = []    # per bucket: the parameters' gradient views into the flat tensor
        cap = int(bucket_mb * 1024 * 1024 / 4)
        cur, cur_n = [], 0
        for p in self.params:
            if cur and cur_n + p.numel() > cap:
                self._close(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self._close(cur)
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._hooks = []
        self.reduce_calls = self.reduce_bytes = 0   # bucket all-reduces issued so far, and their bytes (bench.py config.collectives)
        for bi, (_, ps) in enumerate(self.buckets):
            for p in ps:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self.zero_grad()

    @property
    def active(self):
        """collectives are issued: more than one rank (tests switch the reducer off by setting .world = 1), or forced"""
        return dist.is_initialized() and (self.world > 1 or _force())

    def _close(self, ps):
        n = sum(p.numel() for p in ps)
        flat = torch.zeros(n, dtype=ps[0].dtype, device=ps[0].device)
        views, off = [], 0
        for p in ps:
            # same (dense) strides as the parameter, e.g. channels_last_3d weights
            views.append(flat[off:off + p.numel()].as_strided(p.size(), p.stride()))
            off += p.numel()
        self.buckets.append((flat, ps))
        self.views.append(views)

    def _pack(self, bi):
        """the bucket's gradients -> its flat buffer (one multi-tensor copy), .grad -> views of it; a parameter that received no
        gradient contributes zeros (what the pre-zeroed buffers of the first version gave)"""
        _, ps = self.buckets[bi]
        views = self.views[bi]
        dst, src = [], []
        with torch.no_grad():
            for p, v in zip(ps, views):
                if p.grad is None:
                    v.zero_()
                elif p.grad is not v:
                    dst.append(v)
                    src.append(p.grad)
            if dst:
                torch._foreach_copy_(dst, src)
        for p, v in zip(ps, views):
            p.grad = v

    def _make_hook(self, bi):
        def hook(_p):
            self._pending[bi] -= 1
            if self._pending[bi] == 0 and self.active:
                self._pack(bi)
                self._reduce(bi)
        return hook

    def _reduce(self, bi):
        flat = self.buckets[bi][0]
        self.reduce_calls += 1
        self.reduce_bytes += flat.numel() * flat.element_size()
        if self.direct is not None:
            self.direct(flat)        # on the current (compute) stream; ordered by the stream, nothing to wait for
        else:
            self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def reset(self):
        """Call before every backward (after zero_grad)."""
        self._pending = [len(ps) for _, ps in self.buckets]
        self._handles = []

    def zero_grad(self):
        for p in self.params:
            p.grad = None      # nothing to fill: autograd assigns, _pack copies
        self.reset()

    def finish(self):
        """Wait for the in-flight buckets, pack and reduce any bucket whose parameters did not all receive a gradient this step
        (unused branches), and turn sums into means (DDP semantics: mean of per-rank gradients)."""
        if not self.active:
            return
        for bi, n in enumerate(self._pending):
            if n > 0:
                self._pack(bi)
                self._reduce(bi)
        for h in self._handles:
            h.wait()
        inv = 1.0 / self.world
        if inv != 1.0:
            torch._foreach_mul_([flat for flat, _ in self.buckets], inv)
        self._handles = []


def broadcast_parameters(modules, src=0, process_group=None):
    """One-time weight/buffer sync from rank 0 (what DDP's constructor does, trainer.py:135)."""
    if not dist.is_initialized() or (dist.get_world_size(process_group) == 1 and not _force()):
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            if t.device.type == "cpu" and dist.get_backend(process_group) == "nccl":
                # host-resident BatchNorm counters (--bn_counter_on_host): RCCL moves device memory only, so they travel through
                # a device copy -- rank 0's value wins, also after a load_model() that only it performed
                tmp = t.data.to(torch.device("cuda", torch.cuda.current_device()))
                dist.broadcast(tmp, src=src, group=process_group)
                t.data.copy_(tmp)
                continue
            dist.broadcast(t.data, src=src, group=process_group)
