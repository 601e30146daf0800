"""Small all-reduces issued straight to RCCL on the CALLING stream.

Why.  A synchronised-BatchNorm step has 230 all-reduces of a few hundred bytes, each between two kernels that depend on it.
torch.distributed runs a collective on the process group's own stream: an event recorded on the compute stream, a wait on the
communicator's stream, the collective, an event back, a wait on the compute stream -- two cross-stream dependencies, i.e. two
round trips through the command processor with the GPU idle.  Measured with a group of ONE rank (no bytes to move, RCCL launches
nothing): 43.5 ms per step against 41.6 without collectives, with identical kernel time (profiles/r04_ddp_one_rank.txt): 2 ms of
idle gaps = 230 x 8.5 us.  ncclAllReduce on the compute stream itself has no such gaps.

How.  A dedicated process group (its own communicator: one communicator is driven from one stream only -- the gradient buckets
stay on torch's group and stream) is created by torch, its communicator handle taken from ProcessGroupNCCL._comm_ptr(), and
ncclAllReduce called through ctypes from the RCCL library torch itself has loaded (the same instance that owns the communicator).
Every rank issues the same calls in the same order (program order of one thread).  Anything missing -- another backend (the gloo
tests), an older torch without _comm_ptr, MD_DIRECT_RCCL=0 -- and callers fall back to torch.distributed.all_reduce."""
import ctypes
import os

import torch
import torch.distributed as dist

_NCCL_DTYPE = {torch.float16: 6, torch.float32: 7, torch.float64: 8, torch.bfloat16: 9, torch.int32: 2, torch.int64: 4}
_NCCL_SUM = 0


class DirectAllReduce:
    """callable(tensor): in-place sum over the group on torch's current stream; .size = ranks; .calls = all-reduces issued"""

    def __init__(self, group, comm_ptr, fn):
        self.group, self.comm, self._fn = group, ctypes.c_void_p(comm_ptr), fn
        self.size = dist.get_world_size(group)
        self.calls = 0

    def __call__(self, t):
        if not t.is_contiguous():
            raise RuntimeError("DirectAllReduce: contiguous tensors only")
        p = ctypes.c_void_p(t.data_ptr())
        rc = self._fn(p, p, ctypes.c_size_t(t.numel()), _NCCL_DTYPE[t.dtype], _NCCL_SUM, self.comm,
                      ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(t.device.index)))
        if rc != 0:
            raise RuntimeError("ncclAllReduce failed with code %d" % rc)
        self.calls += 1
        return t


def make(group=None):
    """A DirectAllReduce over a NEW group with the ranks of `group` (default: all), or None when the direct path is unavailable.
    Collective: every rank of the group must call it."""
    if os.environ.get("MD_DIRECT_RCCL", "1") == "0" or not dist.is_initialized() or dist.get_backend(group) != "nccl":
        return None
    try:
        ranks = dist.get_process_group_ranks(group if group is not None else dist.group.WORLD)
        g = dist.new_group(ranks=ranks, backend="nccl")
        dev = torch.device("cuda", torch.cuda.current_device())
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe, group=g)          # creates the communicator
        torch.cuda.synchronize()
        backend = g._get_backend(dev)
        comm = backend._comm_ptr()
        lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
        fn = lib.ncclAllReduce
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        d = DirectAllReduce(g, comm, fn)
        # the first direct call, checked: sum of ones = number of ranks, on every rank, or nobody uses the path
        # (in the two element types the BatchNorm layers use, with rank-dependent addends: 1 + rank summed over the ranks)
        rank = dist.get_rank(g)
        want = float(d.size * (d.size + 1) // 2)
        good = True
        for dt in (torch.float64, torch.float32):
            check = torch.full((6,), float(rank + 1), device=dev, dtype=dt)
            d(check)
            torch.cuda.synchronize()
            good = good and bool((check == want).all())
        ok = torch.tensor([1.0 if good else 0.0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=g)
        d.calls = 0
        return d if float(ok.item()) == 1.0 else None
    except Exception as e:   # an older torch (no _comm_ptr), a library that is not where torch usually ships it, ...
        import sys
        print("movedepth_amd: direct RCCL all-reduce unavailable (%s: %s): torch.distributed is used" % (type(e).__name__, e), file=sys.stderr)
        return None
