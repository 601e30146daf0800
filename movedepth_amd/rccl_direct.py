"""Small all-reduces issued straight to RCCL on the CALLING stream.  OPT-IN: --direct_rccl 1 (default: MD_DIRECT_RCCL=1 of the launcher).

Why.  A synchronised-BatchNorm step has 230 all-reduces of a few hundred bytes, each between two kernels that depend on it.
torch.distributed runs a collective on the process group's own stream: an event recorded on the compute stream, a wait on the
communicator's stream, the collective, an event back, a wait on the compute stream -- two cross-stream dependencies, i.e. two
round trips through the command processor with the GPU idle.  Measured with a group of ONE rank (no bytes to move, RCCL launches
nothing): 43.5 ms per step against 41.6 without collectives, with identical kernel time (profiles/r04_ddp_one_rank.txt): 2 ms of
idle gaps = 230 x 8.5 us.  ncclAllReduce on the compute stream itself has no such gaps.

How.  A dedicated process group is created by torch, its communicator handle taken from ProcessGroupNCCL._comm_ptr(), and
ncclAllReduce called through ctypes from the RCCL library torch itself has loaded (the same instance that owns the communicator).
Every rank issues the same calls in the same order (program order of one thread).

One communicator, one stream.  Two communicators driven concurrently from two streams may execute in different orders on
different ranks -- the documented NCCL / RCCL hang.  When the direct path is on, EVERY collective of the training step goes
through it: the BatchNorm statistics and the gradient buckets (dp.GradSync(direct=...)), all on the compute stream, so that their
order is the program order of one thread on every rank.  When it is off, every collective goes through torch's group (its own
stream, one communicator) -- again one total order.  Off is the default: the direct path has run with a group of one rank only
(tests/test_dp_rccl_single_rank_gpu.py; no box with two GPUs was available to this build), and what has not run with N > 1 is not
what a first multi-GPU job should depend on.

make() is itself collective.  It is built from stages; after EVERY stage all ranks exchange a success flag over the PARENT group
(torch's own, known-good communicator) and stop together at the first stage any of them failed -- so an exception on one rank (an
older torch without _comm_ptr, a library that is not where torch usually ships it) becomes "None on every rank", never some ranks
inside a collective the others will not join.  A stage is either purely local or a collective that every rank enters; the flag
exchange sits between them.  tests/test_dp_gloo.py runs the protocol over gloo (world size 2) with a failure injected on one rank."""
import ctypes
import os
import sys

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


class Skip(Exception):
    """raised by a stage that declines quietly (the path is switched off on this rank): no message"""


def agree(ok, group=None, device=None):
    """True on every rank iff `ok` is true on every rank of `group` (one MIN all-reduce on the group's own backend)."""
    flag = torch.tensor([1.0 if ok else 0.0], device=device if device is not None else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item() == 1.0)


def run_stages(stages, group=None, device=None, label="rccl_direct"):
    """Run `stages` -- (name, fn(state)) pairs -- in order on every rank of `group`, exchanging a success flag after each one
    (agree()).  Returns the state dict when every stage succeeded on every rank, else None ON EVERY RANK: the ranks leave at the
    same stage, so none of them enters a later stage's collective alone.  A stage that raises is reported on stderr by the rank
    it failed on.  MD_DIRECT_RCCL_FAIL="<rank>:<stage name>" injects a failure (tests)."""
    state = {}
    inject = os.environ.get("MD_DIRECT_RCCL_FAIL", "")
    rank = dist.get_rank(group)
    for name, fn in stages:
        ok = True
        try:
            if inject == "%d:%s" % (rank, name):
                raise RuntimeError("injected failure (MD_DIRECT_RCCL_FAIL)")
            fn(state)
        except Skip:
            ok = False
        except Exception as e:
            ok = False
            print("movedepth_amd: %s: stage '%s' failed on rank %d (%s: %s)" % (label, name, rank, type(e).__name__, e), file=sys.stderr)
        if not agree(ok, group, device):
            if ok:
                print("movedepth_amd: %s: stage '%s' failed on another rank; rank %d falls back with it" % (label, name, rank), file=sys.stderr)
            return None
    return state


ENABLED = None   # set by the trainer from --direct_rccl; None: the launcher's MD_DIRECT_RCCL


def enabled():
    return bool(ENABLED) if ENABLED is not None else os.environ.get("MD_DIRECT_RCCL", "0") == "1"


def make(group=None):
    """A DirectAllReduce over a NEW group with the ranks of `group` (default: all), or None -- on every rank alike -- when the direct
    path is switched off (the default), unavailable, or failed anywhere.  Collective: every rank of the group must call it."""
    if not dist.is_initialized() or dist.get_backend(group) != "nccl":
        return None        # the same on every rank by construction (one backend per group)
    dev = torch.device("cuda", torch.cuda.current_device())
    parent = group if group is not None else dist.group.WORLD

    def s_want(st):        # local: the environment may differ between ranks -- it is part of what is agreed on
        if not enabled():
            raise Skip()

    def s_group(st):       # collective on the parent: every rank is here (agreed above)
        st["g"] = dist.new_group(ranks=dist.get_process_group_ranks(parent), backend="nccl")
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe, group=st["g"])          # creates the communicator
        torch.cuda.synchronize()

    def s_handle(st):      # local: private torch API, the library file
        comm = st["g"]._get_backend(dev)._comm_ptr()
        lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
        fn = lib.ncclAllReduce
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        st["d"] = DirectAllReduce(st["g"], comm, fn)

    def s_check(st):       # collective through the new path: every rank holds a handle (agreed above).  Sums of 1 + rank over
        d = st["d"]        # the ranks, in the element types the BatchNorm layers and the gradient buckets use
        rank = dist.get_rank(st["g"])
        want = float(d.size * (d.size + 1) // 2)
        for dt in (torch.float64, torch.float32):
            check = torch.full((6,), float(rank + 1), device=dev, dtype=dt)
            d(check)
            torch.cuda.synchronize()
            if not bool((check == want).all()):
                raise RuntimeError("direct all-reduce returned %s, expected %s" % (check.tolist(), want))
        d.calls = 0

    # (also when the path is off, the default: one flag exchange on torch's own group at start-up, so that ranks whose
    # environments differ still leave together)
    st = run_stages([("want", s_want), ("group", s_group), ("handle", s_handle), ("check", s_check)], parent, dev)
    return st["d"] if st is not None else None
