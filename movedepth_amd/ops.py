"""torch.autograd.Function wrappers over the C ABI (include/movedepth_hip.h).

PyTorch is plumbing here: it owns device memory, the current HIP stream and autograd bookkeeping; every
forward/backward below is one or two launches of a hand-written gfx950 kernel.  No CPU fallback exists:
tensors must live on the GPU and the HIP library must load, otherwise these raise.
"""
import ctypes

import torch

from . import _lib

_TYPES = {"inverse": 0, "linear": 1, "log": 2}

# Optional per-launch timing with HIP events on the stream the kernels are launched on (torch's current
# stream).  bench.py switches it on for the roofline figure; off (None) it costs nothing.
KERNEL_EVENTS = None  # or dict: entry-point name -> list of (start_event, end_event)


def enable_kernel_timing(names):
    global KERNEL_EVENTS
    KERNEL_EVENTS = {n: [] for n in names}


TIME_ROOFLINE, TIME_PHOTOMETRIC, TIME_BATCHNORM = 2, 4, 8   # classes of md_kernel_timing_enable (include/movedepth_hip.h)


def enable_library_kernel_timing(on=True):
    """HIP events recorded INSIDE the library, directly around its kernel launches (md_kernel_timing_*).  True: every class;
    an int: a mask of TIME_* classes (a timed dispatch costs its stream ~5 us, so a training step times only what it reports)."""
    _lib.call("md_kernel_timing_enable", int(on))


def library_kernel_times_us(names):
    out = {}
    for n in names:
        avg, mn, cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        _lib.call("md_kernel_timing_read", n.encode(), ctypes.byref(avg), ctypes.byref(mn), ctypes.byref(cnt))
        if cnt.value:
            buf = (ctypes.c_double * cnt.value)()
            _lib.load().md_kernel_timing_list(n.encode(), buf, cnt.value)
            ts = sorted(buf)
            out[n] = {"avg_us": avg.value, "min_us": mn.value, "median_us": ts[len(ts) // 2], "launches": cnt.value,
                      "all_us": list(buf)}
    return out


def kernel_times_us():
    """Average duration per timed entry point (call after torch.cuda.synchronize())."""
    out = {}
    for n, evs in (KERNEL_EVENTS or {}).items():
        if evs:
            ts = [a.elapsed_time(b) * 1e3 for a, b in evs]
            out[n] = {"avg_us": sum(ts) / len(ts), "min_us": min(ts), "launches": len(ts)}
    return out


def _timed_call(name, *args):
    if KERNEL_EVENTS is not None and name in KERNEL_EVENTS:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.call(name, *args)
        b.record()
        KERNEL_EVENTS[name].append((a, b))
    else:
        _lib.call(name, *args)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _prep(t, name):
    """float32, contiguous, on the GPU -- or raise (never silently move work to the CPU)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.MovedepthHipError("%s must be a GPU tensor (got %s): the HIP path has no CPU fallback" % (name, t.device))
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# --------------------------------------------------------------------------- schedule
def schedule_depth_range(prior_depth, ndepth, scale_fac, z_trans=None, type="inverse"):
    """schedule_depth_rangev2 / schedule_depth_range_zv2 (reference layers.py:256-284 / 370-398). no_grad."""
    if type not in _TYPES:
        raise NotImplementedError(type)
    with torch.no_grad():
        prior = _prep(prior_depth, "prior_depth")
        B, _, h, w = prior.shape
        zt = None
        if z_trans is not None:
            zt = _prep(z_trans, "z_trans").reshape(-1)
            if zt.numel() != B:
                # the reference broadcast only works for one lookup frame
                raise RuntimeError("z_trans must have one value per sample (got %d for B=%d)" % (zt.numel(), B))
        out = torch.empty(B, ndepth, h, w, device=prior.device, dtype=torch.float32)
        _lib.call("md_schedule_depth_range", _p(prior), _p(zt), B, h, w, ndepth, float(scale_fac), _TYPES[type], _p(out),
                  _stream())
    return out


# --------------------------------------------------------------------------- cost volume
def _vol_alloc(layout, B, D, G, h, w, device, dtype=torch.float32):
    """Storage + (sb, sd, sg, sp) strides + the logical (B,D,G,h,w) view for a grouped volume.
       'bdg'  : (B,D,G,h,w) contiguous -- the reference's layout;
       'bgd'  : (B,G,D,h,w) contiguous -- what reg3d's permute asks for (NCDHW);
       'ndhwc': (B,D,h,w,G) contiguous -- channels_last_3d for reg3d, the layout MIOpen's fast 3-D convs take."""
    if layout == "bgd":
        store = torch.empty(B, G, D, h, w, device=device, dtype=dtype)
        return store, (store.stride(0), store.stride(2), store.stride(1), 1), store.permute(0, 2, 1, 3, 4)
    if layout == "ndhwc":
        store = torch.empty(B, D, h, w, G, device=device, dtype=dtype)
        return store, (store.stride(0), store.stride(1), 1, G), store.permute(0, 1, 4, 2, 3)
    if layout == "bdg":
        store = torch.empty(B, D, G, h, w, device=device, dtype=dtype)
        return store, (store.stride(0), store.stride(1), store.stride(2), 1), store
    raise ValueError("unknown volume layout %r" % (layout,))


def _vol_as_layout(t, layout):
    """Logical (B,D,G,h,w) tensor -> contiguous storage in `layout` (no copy when it already is) + strides."""
    if layout == "bgd":
        g = t.permute(0, 2, 1, 3, 4).contiguous()
        return g, (g.stride(0), g.stride(2), g.stride(1), 1)
    if layout == "ndhwc":
        g = t.permute(0, 1, 3, 4, 2).contiguous()
        return g, (g.stride(0), g.stride(1), 1, g.shape[4])
    g = t.contiguous()
    return g, (g.stride(0), g.stride(1), g.stride(2), 1)


def _vol_logical(store, layout):
    if layout == "bgd":
        return store.permute(0, 2, 1, 3, 4)
    if layout == "ndhwc":
        return store.permute(0, 1, 4, 2, 3)
    return store


_HALF_SUFFIX = {torch.bfloat16: "_bf16", torch.float16: "_f16"}

# False: always hand the plane-sweep kernels planar [B,C,h,w] feature maps (a layout copy when the encoder runs channels_last)
FEATURES_CHANNELS_LAST = True


def _is_cl(t):
    """4-D tensor stored [B,h,w,C] (and not also [B,C,h,w]-contiguous, as any tensor with C == 1 or h == w == 1 is)."""
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()


class BackwardPolicy:
    """Per-launch choices of md_costvol_bwd* that depend on the POSES, made on the host without a synchronisation and without anything
    read from the environment (ABI 17: `flags`, `census`, `shares`, `cost`).  Every backward launch leaves, in one small device
    buffer, its census -- steps walked in gather mode, all steps, windows staged, segments -- and the shader cycles it spent per work item
    (16 x 4-pixel tile of one sample); an asynchronous copy brings both to pinned host memory behind the launch, and the NEXT launch
    of the same shape reads whatever has landed by then (this step's earlier volume or the previous step's: poses and depths drift
    over hundreds of steps).  From them:
      * MD_CV_GATHER_TABLE when more than `threshold` of the steps were gathered (wild poses of an untrained pose network: ~65 % at
        config 2's shape; moderate ~27 %, driving scene < 1 %, sane 0): the cell-table build pays only there;
      * a partition of the launch's work with equal COST per workgroup instead of equal steps when the items' costs are very uneven
        (max / mean above `imbalance`) and next to nothing is gathered: all workgroups of the backward are resident at once, so
        nothing else evens out what parallax makes uneven -- a driving-scene launch lasts as long as its slowest tile
        (csrc/costvol.hip launch_cl_inst; 107.9 -> 98.4 us at 1 m per frame, profiles/r06_costvol_spec.txt).  The measure is noisy --
        equal tiles differ by 1.5-1.8x in cycles with where they ran -- so the bar is high (2.0; sane launches read 1.2-1.8 and stay on the library's
        own partition, d_ref stored without a fill), and launches with gathered sub-slices are left alone: their tail is the chip's
        atomic rate, which no partition changes (moderate poses: 122.9 -> 142.8 us balanced).
      * MD_CV_FINE_SLICES for the FORWARD of the same shape (fp32) when the last backward staged more than `fine_above` windows per
        segment or gathered anything: under parallax twice the slices per item even the forward out (moderate 66 -> 62 us).
    `force_table` / `force_balance` / `force_fine` = True / False pin a choice (tests, A/B runs)."""

    RING = 4   # pinned upload buffers per shape: one is rewritten only after its copy has completed (event), else the launch goes unbalanced

    def __init__(self, device, threshold=0.45, imbalance=2.0, max_gathered_for_balance=0.05):
        self.device = device
        self.threshold, self.imbalance, self.max_gathered = float(threshold), float(imbalance), float(max_gathered_for_balance)
        self.force_table = self.force_balance = self.force_fine = None
        self.fine_above = 1.005
        self.launches = self.table_launches = self.balanced_launches = self.fine_launches = 0
        self._shapes = {}
        self._last = None

    # -- per-shape state
    def _state(self, key):
        st = self._shapes.get(key)
        if st is None:
            import ctypes as C
            B, Cc, G, h, w, D, fcl, cl_volume = key
            items, nwg = C.c_int(0), C.c_int(0)
            if cl_volume:
                _lib.call("md_costvol_bwd_plan", B, Cc, G, h, w, D, int(fcl), C.byref(items), C.byref(nwg))
            st = type("S", (), {})()
            st.items, st.nwg, st.D = items.value, nwg.value, D
            n = max(st.items, 1)
            st.dev = torch.zeros(2 + n, dtype=torch.int32, device=self.device)       # [census word (lo, hi), cost per item ...]
            st.host = torch.zeros(2 + n, dtype=torch.int32).pin_memory()
            st.np = st.host.numpy()          # numpy views of the pinned words, made once: a read costs no tensor indexing
            st.census_np = st.np[:2].view("uint64")      # the census word
            st.cost_np = st.np[2:2 + st.items]
            st.census_ok = st.items * D < (1 << 20) and st.nwg < (1 << 13)     # (18-bit step counts in units of 4, 14-bit window / segment counts)
            st.shares_dev = torch.zeros(max(st.nwg, 1), 2, dtype=torch.int64, device=self.device)
            st.ring = [(torch.zeros(max(st.nwg, 1), 2, dtype=torch.int64).pin_memory(), torch.cuda.Event()) for _ in range(self.RING)]
            st.ring_used = [False] * self.RING
            st.slot = 0
            self._shapes[key] = st
        return st

    @staticmethod
    def _census(word):
        """(gathered steps, all steps, windows staged, segments) of a census word (include/movedepth_hip.h)"""
        v = int(word[0])
        return ((v >> 46) & 0x3FFFF) * 4, ((v >> 28) & 0x3FFFF) * 4, (v >> 14) & 0x3FFF, v & 0x3FFF

    def gathered_share(self, key=None):
        """Share of the last landed census' hypothesis steps that ran in gather mode (plain read of pinned memory: never waits)."""
        st = self._shapes.get(key) if key is not None else self._last
        if st is None:
            return 0.0
        if not st.census_ok:
            return 0.0
        g, t, _, _ = self._census(st.census_np)
        return g / t if t else 0.0

    def costs(self, key=None):
        """Per-item shader cycles of the last landed launch (numpy view of pinned memory)."""
        st = self._shapes.get(key) if key is not None else self._last
        return None if st is None or st.items == 0 else st.cost_np

    def partition(self, cost, nwg, D, quantum=8):
        """[lo, hi) per workgroup in step units (item * D + d): equal cost per workgroup, boundaries on multiples of `quantum` steps
        inside an item (a sub-slice shorter than that is all fixed cost)."""
        import numpy as np
        c = np.maximum(cost.astype(np.float64), 1.0)
        acc = np.concatenate([[0.0], np.cumsum(c)])
        x = acc[-1] * np.arange(nwg + 1, dtype=np.float64) / nwg
        i = np.clip(np.searchsorted(acc, x, side="right") - 1, 0, len(c) - 1)
        step = i * D + np.minimum(np.round((x - acc[i]) / c[i] * D / quantum) * quantum, D).astype(np.int64)
        step[0], step[-1] = 0, len(c) * D
        step = np.maximum.accumulate(step)
        return np.stack([step[:-1], step[1:]], 1)

    def before_launch(self, key):
        """-> (flags, census pointer, shares pointer, n_shares, cost pointer) for the launch about to be made on the current stream"""
        st = self._last = self._state(key)
        self.launches += 1
        table = self.force_table if self.force_table is not None else self.gathered_share(key) > self.threshold
        self.table_launches += int(bool(table))
        flags = _lib.CV_GATHER_TABLE if table else 0
        base = st.dev.data_ptr()
        if st.items == 0:
            return flags, base, None, 0, None
        shares, n = None, 0
        cost = st.cost_np
        want = self.force_balance
        if want is None:
            tot = float(cost.sum())
            want = tot > 0 and float(cost.max()) * st.items > self.imbalance * tot and self.gathered_share(key) < self.max_gathered
        if want and float(cost.sum()) > 0:
            buf, ev = st.ring[st.slot]
            if not st.ring_used[st.slot] or ev.query():     # (never waits: a busy slot means this launch goes unbalanced)
                buf.numpy()[:] = self.partition(cost, st.nwg, st.D)
                st.shares_dev.copy_(buf, non_blocking=True)
                ev.record()
                st.ring_used[st.slot] = True
                st.slot = (st.slot + 1) % self.RING
                shares, n = st.shares_dev.data_ptr(), st.nwg
                self.balanced_launches += 1
        return flags, base, shares, n, base + 8

    def windows_per_segment(self, key=None):
        """Windows the last landed backward staged per segment: 1 when every tile's sweep fits its window (sane poses), more under parallax."""
        st = self._shapes.get(key) if key is not None else self._last
        if st is None:
            return 0.0
        _, _, wnd, seg = self._census(st.census_np)
        return wnd / seg if seg else 0.0

    def forward_flags(self, key, fp32):
        """MD_CV_FINE_SLICES for the forward of this shape when the last landed backward of the same shape shows parallax (windows per
        segment off 1 by more than `fine_above` - 1, or anything gathered): twice the slices even out what the tiles' unequal sweeps make uneven
        (csrc/costvol.hip launch_cl_inst); fp32 only (the 2-byte forward loses more to the second staging than it gains)."""
        if self.force_fine is not None:
            return _lib.CV_FINE_SLICES if self.force_fine else 0
        if not fp32 or key not in self._shapes:
            return 0
        wps = self.windows_per_segment(key)   # exactly 1 when every tile's sweep fits its window; sub-slices raise it, gathered ones lower it
        on = wps > 0 and (abs(wps - 1.0) > self.fine_above - 1.0 or self.gathered_share(key) > 0.005)
        self.fine_launches += int(on)
        return _lib.CV_FINE_SLICES if on else 0

    def after_launch(self, key):
        st = self._shapes[key]
        st.host.copy_(st.dev, non_blocking=True)   # stream-ordered behind the launch; the host does not wait for it

    def last_census(self):
        """(gathered steps, all steps, windows staged, segments) of the most recent launch (steps counted in units of 4 per workgroup), read from the DEVICE
        (synchronises: tests and tools only)."""
        return self._census(self._last.dev[:2].cpu().numpy().view("uint64"))

    def reset(self):
        """forget every shape's history (tests)"""
        torch.cuda.synchronize()
        self._shapes.clear()
        self._last = None
        self.force_table = self.force_balance = self.force_fine = None

    # round-6 name of the first half of this object (tests, tools)
    @property
    def force(self):
        return self.force_table

    @force.setter
    def force(self, v):
        self.force_table = v


GatherTablePolicy = BackwardPolicy
_BACKWARD_POLICIES = {}


def backward_policy(device=None):
    """The process' policy object of `device` (created on first use)."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx not in _BACKWARD_POLICIES:
        _BACKWARD_POLICIES[idx] = BackwardPolicy(torch.device("cuda", idx))
    return _BACKWARD_POLICIES[idx]


gather_table_policy = backward_policy


class _CostVolume(torch.autograd.Function):
    """Grouped plane-sweep volume; returns a tensor of logical shape (B,D,G,h,w) over `layout` storage."""

    @staticmethod
    def forward(ctx, ref, src, K, invK, pose, hyp, prior, ztrans, scale_fac, sched_type, G, D, layout):
        # bf16 / fp16 feature maps (mixed-precision configs): the 2-byte build of the kernel reads them and writes a volume
        # of the same type; everything else (and any other dtype) is fp32
        io = ref.dtype if (ref.dtype in _HALF_SUFFIX and src.dtype == ref.dtype) else torch.float32
        for t, nm in ((ref, "ref"), (src, "src")):
            if not t.is_cuda:
                raise _lib.MovedepthHipError("%s must be a GPU tensor (got %s): the HIP path has no CPU fallback" % (nm, t.device))
        if io == torch.float32:
            ref, src = ref.float(), src.float()
        B, C, h, w = ref.shape
        # Feature maps in channels_last (memory [B,h,w,C], what FPN4 produces when the 2-D networks run in that format) go to
        # the kernels as they are -- and d_ref / d_src come back in the same format -- whenever the channels-last-volume
        # kernels serve the problem; anything else is taken as [B,C,h,w] (one layout copy if it is not).
        fcl = (layout == "ndhwc" and G in (8, 16) and C % G == 0 and C // G in (1, 2, 4) and FEATURES_CHANNELS_LAST
               and _is_cl(ref) and (_is_cl(src) or src.is_contiguous()))
        if fcl:
            src = src.contiguous(memory_format=torch.channels_last)
        else:
            ref, src = ref.contiguous(), src.contiguous()
        sfx = _HALF_SUFFIX.get(io, "")
        K, invK, pose = _prep(K, "K"), _prep(invK, "invK"), _prep(pose, "pose")
        hyp, prior, ztrans = _prep(hyp, "depth_priors"), _prep(prior, "prior"), _prep(ztrans, "z_trans")
        store, (sb, sd, sg, sp), out = _vol_alloc(layout, B, D, G, h, w, ref.device, io)
        fflags = backward_policy(ref.device).forward_flags((B, C, G, h, w, D, bool(fcl), layout == "ndhwc"), io == torch.float32)
        _timed_call("md_costvol_fwd" + sfx, _p(ref), _p(src), _p(K), _p(invK), _p(pose), _p(hyp), _p(prior), _p(ztrans),
                    float(scale_fac), int(sched_type), B, C, G, h, w, D, int(fcl), _p(store), sb, sd, sg, sp, fflags, _stream())
        ctx.sfx, ctx.io, ctx.fcl = sfx, io, fcl
        ctx.save_for_backward(ref, src, K, invK, pose, hyp if hyp is not None else torch.empty(0),
                              prior if prior is not None else torch.empty(0),
                              ztrans if ztrans is not None else torch.empty(0))
        ctx.meta = (float(scale_fac), int(sched_type), G, D, layout, hyp is not None, prior is not None,
                    ztrans is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        ref, src, K, invK, pose, hyp, prior, ztrans = ctx.saved_tensors
        scale_fac, sched_type, G, D, layout, has_hyp, has_prior, has_z = ctx.meta
        B, C, h, w = ref.shape
        g, (sb, sd, sg, sp) = _vol_as_layout(gout.to(ctx.io), layout)  # no copy when the consumer kept the layout
        # fp32 accumulation (atomics); one allocation, d_ref then d_src: the library zeroes them with a single fill
        if ctx.fcl:   # gradients in the features' own format: (B,C,h,w) views of [B,h,w,C] storage
            d_both = torch.empty((2, B, h, w, C), device=ref.device, dtype=torch.float32).permute(0, 1, 4, 2, 3)
        else:
            d_both = torch.empty((2,) + tuple(ref.shape), device=ref.device, dtype=torch.float32)
        d_ref, d_src = d_both[0], d_both[1]
        pol = backward_policy(ref.device)
        key = (B, C, G, h, w, D, bool(ctx.fcl), layout == "ndhwc")   # (other volume layouts: census only, the library's own partition)
        flags, census, shares, n_shares, cost = pol.before_launch(key)
        _timed_call("md_costvol_bwd" + ctx.sfx, _p(g), sb, sd, sg, sp, _p(ref), _p(src), _p(K), _p(invK), _p(pose),
                    _p(hyp if has_hyp else None), _p(prior if has_prior else None), _p(ztrans if has_z else None),
                    scale_fac, sched_type, B, C, G, h, w, D, int(ctx.fcl), _p(d_ref), _p(d_src), flags, census, shares, n_shares, cost,
                    _stream())
        pol.after_launch(key)
        return (d_ref.to(ctx.io), d_src.to(ctx.io)) + (None,) * 11


def costvol_grouped(ref, src, K, invK, pose, G, depth_priors=None, prior=None, ndepth=None, scale_fac=0.3,
                    z_trans=None, type="inverse", layout="bgd"):
    """Plane-sweep volume with the group mean fused in.  Either `depth_priors` (B,D,h,w) or `prior` (B,1,h,w) +
    schedule parameters (the schedule is then evaluated inside the kernel, same arithmetic as
    schedule_depth_range).  pose: (B,4,4).  Returns logical (B,D,G,h,w)."""
    if depth_priors is not None:
        D = depth_priors.shape[1]
        prior = None
    else:
        D = int(ndepth)
    zt = None if (z_trans is None or depth_priors is not None) else z_trans.reshape(-1)
    return _CostVolume.apply(ref, src, K, invK, pose.reshape(-1, 4, 4), depth_priors, prior, zt, scale_fac,
                             _TYPES[type], int(G), D, layout)


class _FuseVolumes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, layout, *vols):
        N = len(vols)
        B, D, G, h, w = vols[0].shape
        vs, strides = [], None
        for v in vols:
            st, strides = _vol_as_layout(v.float(), layout)   # the fusion kernels are fp32 (2-byte volumes are widened here)
            vs.append(st)
        sb, sd, sg, sp = strides
        store = torch.empty_like(vs[0])
        weights = torch.empty(N, B, h, w, device=store.device, dtype=torch.float32)
        arr = (ctypes.c_void_p * N)(*[v.data_ptr() for v in vs])
        _lib.call("md_fuse_fwd", arr, N, B, D, G, h * w, sb, sd, sg, sp, 0, _p(store), _p(weights), _stream())
        ctx.save_for_backward(*vs)
        ctx.meta = (layout, N, B, D, G, h, w, sb, sd, sg, sp)
        ctx.mark_non_differentiable(weights)
        return _vol_logical(store, layout), weights

    @staticmethod
    def backward(ctx, gout, _gw):
        vs = ctx.saved_tensors
        layout, N, B, D, G, h, w, sb, sd, sg, sp = ctx.meta
        g, _ = _vol_as_layout(gout.float(), layout)
        ds = [torch.empty_like(v) for v in vs]
        arr = (ctypes.c_void_p * N)(*[v.data_ptr() for v in vs])
        darr = (ctypes.c_void_p * N)(*[d.data_ptr() for d in ds])
        _lib.call("md_fuse_bwd", _p(g), arr, N, B, D, G, h * w, sb, sd, sg, sp, darr, _stream())
        return (None,) + tuple(_vol_logical(d, layout) for d in ds)


def fuse_volumes(vols, layout="bgd", exact_single_frame=False):
    """Confidence-weighted fusion over lookup frames (reference trainer.py:349-363) -> (cor_feats, weights|None).

    With one lookup frame the reference's result equals its input to 1.6e-7 relative (w >= 1/G against the
    1e-8 guard) and the gradient through w is O(1e-8); the kernel is then skipped unless exact_single_frame."""
    if len(vols) == 1 and not exact_single_frame:
        return vols[0], None
    for v in vols:
        if not v.is_cuda:
            raise _lib.MovedepthHipError("cost volumes must be GPU tensors: the HIP path has no CPU fallback")
    return _FuseVolumes.apply(layout, *[v.float() for v in vols])


def fuse_volumes_eval(vols, layout="bgd"):
    """The evaluation script's fusion over lookup frames (reference evaluate_depth.py:225-243): the confidence weight is the
    max over D of softmax_D(mean over G) -- not the training one -- one kernel, forward only -> (cor_feats, weights)."""
    if len(vols) == 1:
        return vols[0], None
    with torch.no_grad():
        N = len(vols)
        B, D, G, h, w = vols[0].shape
        vs, strides = [], None
        for v in vols:
            if not v.is_cuda:
                raise _lib.MovedepthHipError("cost volumes must be GPU tensors: the HIP path has no CPU fallback")
            st, strides = _vol_as_layout(v.float(), layout)
            vs.append(st)
        sb, sd, sg, sp = strides
        store = torch.empty_like(vs[0])
        weights = torch.empty(N, B, h, w, device=store.device, dtype=torch.float32)
        arr = (ctypes.c_void_p * N)(*[v.data_ptr() for v in vs])
        _lib.call("md_fuse_fwd", arr, N, B, D, G, h * w, sb, sd, sg, sp, 1, _p(store), _p(weights), _stream())
    return _vol_logical(store, layout), weights


# --------------------------------------------------------------------------- photometric warp
class _WarpBorder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, depth, K, invK, T, want_pix, want_mask):
        img, depth = _prep(img, "img"), _prep(depth, "depth")
        K, invK, T = _prep(K, "K"), _prep(invK, "invK"), _prep(T, "T")
        B, Ci, H, W = img.shape
        if depth.numel() != B * H * W:
            raise RuntimeError("depth has %d elements, expected B*H*W=%d" % (depth.numel(), B * H * W))
        out = torch.empty_like(img)
        pix = torch.empty(B, H, W, 2, device=img.device, dtype=torch.float32) if want_pix else None
        mask = torch.empty(B, H, W, device=img.device, dtype=torch.uint8) if want_mask else None
        _lib.call("md_warp_fwd", _p(img), _p(depth), _p(K), _p(invK), _p(T), B, Ci, H, W, _p(pix), _p(out), _p(mask),
                  _stream())
        ctx.save_for_backward(img, depth, K, invK, T)
        if want_pix:
            ctx.mark_non_differentiable(pix)
        if want_mask:
            ctx.mark_non_differentiable(mask)
        return out, pix, mask

    @staticmethod
    def backward(ctx, gout, _gp, _gm):
        img, depth, K, invK, T = ctx.saved_tensors
        B, Ci, H, W = img.shape
        g = gout.contiguous().float()
        d_depth = torch.empty_like(depth)
        d_T = torch.empty(B, 4, 4, device=img.device, dtype=torch.float32)
        ws = _ws(_lib.load().md_warp_bwd_ws_bytes(B, H, W), img.device)
        _lib.call("md_warp_bwd", _p(g), _p(img), _p(depth), _p(K), _p(invK), _p(T), B, Ci, H, W, _p(d_depth), _p(d_T),
                  _p(ws), _stream())
        return None, d_depth, None, None, d_T.reshape(T.shape), None, None


def warp_border(img, depth, K, invK, T, want_pix=False, want_mask=False):
    """backproject(depth, invK) -> project(K, T) -> grid_sample(img, border, align_corners=True), fused.
    Gradients flow to `depth` and `T` (not to the image, as in the reference).  Returns (warped, pix, oob_mask)."""
    return _WarpBorder.apply(img, depth, K, invK, T, bool(want_pix), bool(want_mask))


class _DispToDepthUp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, H, W, min_depth, max_depth):
        disp = _prep(disp, "disp")
        B, _, h, w = disp.shape
        depth = torch.empty(B, 1, H, W, device=disp.device, dtype=torch.float32)
        _lib.call("md_disp_to_depth_up_fwd", _p(disp), B, h, w, H, W, float(min_depth), float(max_depth), _p(depth),
                  _stream())
        ctx.save_for_backward(disp)
        ctx.meta = (H, W, float(min_depth), float(max_depth))
        return depth

    @staticmethod
    def backward(ctx, g):
        (disp,) = ctx.saved_tensors
        H, W, mn, mx = ctx.meta
        B, _, h, w = disp.shape
        g = g.contiguous().float()
        d = torch.empty_like(disp)
        _lib.call("md_disp_to_depth_up_bwd", _p(g), _p(disp), B, h, w, H, W, mn, mx, _p(d), _stream())
        return d, None, None, None, None


def disp_to_depth_up(disp, H, W, min_depth, max_depth):
    """F.interpolate(disp, [H,W], bilinear, align_corners=False) then disp_to_depth(...)[1] (trainer.py:512-514)."""
    return _DispToDepthUp.apply(disp, int(H), int(W), min_depth, max_depth)


# --------------------------------------------------------------------------- SSIM + L1
def ssim_map(x, y):
    """SSIM.forward (reference layers.py:663-677), forward only (the trainer uses reprojection_loss)."""
    x, y = _prep(x, "x"), _prep(y, "y")
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    _lib.call("md_ssim", _p(x), _p(y), B, C, H, W, _p(out), _stream())
    return out


class _ReprojLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, ssim_w, no_ssim):
        pred, target = _prep(pred, "pred"), _prep(target, "target")
        B, C, H, W = pred.shape
        out = torch.empty(B, 1, H, W, device=pred.device, dtype=torch.float32)
        _lib.call("md_reproj_loss_fwd", _p(pred), _p(target), B, C, H, W, float(ssim_w), int(no_ssim), _p(out), _stream())
        ctx.save_for_backward(pred, target)
        ctx.meta = (float(ssim_w), int(no_ssim))
        return out

    @staticmethod
    def backward(ctx, g):
        pred, target = ctx.saved_tensors
        ssim_w, no_ssim = ctx.meta
        B, C, H, W = pred.shape
        g = g.contiguous().float()
        d = torch.empty_like(pred)
        _lib.call("md_reproj_loss_bwd", _p(g), _p(pred), _p(target), B, C, H, W, ssim_w, no_ssim, _p(d), _stream())
        return d, None, None, None


def reprojection_loss(pred, target, ssim_w=0.85, no_ssim=False):
    """compute_reprojection_loss (reference trainer.py:535-550) -> (B,1,H,W); gradient to `pred` only."""
    return _ReprojLoss.apply(pred, target, ssim_w, no_ssim)


# --------------------------------------------------------------------------- min / automask / masked mean
class _MaskedMin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, reproj, ident, noise, ext_mask, mvs_mode):
        reproj = _prep(reproj, "reprojection_losses")
        ident, noise, ext_mask = _prep(ident, "identity"), _prep(noise, "noise"), _prep(ext_mask, "mask")
        B, N, H, W = reproj.shape
        mn = torch.empty(B, 1, H, W, device=reproj.device, dtype=torch.float32)
        mask = torch.empty_like(mn)
        loss2 = torch.empty(2, device=reproj.device, dtype=torch.float32)
        ws = _ws(_lib.load().md_masked_min_ws_bytes(B, H, W), reproj.device)
        _lib.call("md_masked_min_fwd", _p(reproj), _p(ident), _p(noise), _p(ext_mask), B, N, H, W, int(mvs_mode), _p(mn),
                  _p(mask), _p(loss2), _p(ws), _stream())
        ctx.save_for_backward(reproj, mask, loss2)
        ctx.mark_non_differentiable(mn, mask)
        return loss2[0].clone(), mn, mask

    @staticmethod
    def backward(ctx, gloss, _g1, _g2):
        reproj, mask, loss2 = ctx.saved_tensors
        B, N, H, W = reproj.shape
        gl = gloss.reshape(1).contiguous().float()
        d = torch.empty_like(reproj)
        _lib.call("md_masked_min_bwd", _p(gl), _p(reproj), _p(mask), _p(loss2), B, N, H, W, _p(d), _stream())
        return d, None, None, None, None


def masked_min_loss(reproj, ident=None, noise=None, ext_mask=None, mvs_mode=False):
    """min over frames + automask + masked mean (reference trainer.py:687-709 / 630-662).
    reproj, ident: (B,N,H,W).  Returns (loss scalar, min map (B,1,H,W), mask (B,1,H,W))."""
    return _MaskedMin.apply(reproj, ident, noise, ext_mask, bool(mvs_mode))


# --------------------------------------------------------------------------- the photometric chain, fused
def _ptr(t):
    return None if t is None else t.data_ptr()


def is_packed(img):
    """(B,H,W,4) contiguous float32 RGBx: the image layout of the fused photometric kernels.  Told from a planar (B,3,H,W) frame
    by shape; the one shape that reads both ways -- (B,3,X,4): a planar frame 4 pixels wide or a packed one 3 rows high -- is
    rejected instead of guessed."""
    if img.dim() == 4 and img.shape[-1] == 4 and img.shape[1] == 3:
        raise _lib.MovedepthHipError("image of shape %s is ambiguous (planar (B,3,H,4) or packed (B,3,W,4)): frames must be "
                                     "wider than 4 pixels and higher than 3" % (tuple(img.shape),))
    return img.dim() == 4 and img.shape[-1] == 4 and img.dtype == torch.float32 and img.is_contiguous()


def packed_view(p):
    """(B,3,H,W) view of a packed (B,H,W,4) image: strides (4HW, 1, 4W, 4), no copy"""
    return p[..., :3].permute(0, 3, 1, 2)


def pack_rgbx(imgs):
    """(B,3,H,W) frames -> packed (B,H,W,4) RGBx, up to five images per launch (md_pack_rgbx).  no_grad: the frames are inputs."""
    imgs = list(imgs)
    out = []
    with torch.no_grad():
        for i in range(0, len(imgs), 5):
            chunk = [_prep(t, "image") for t in imgs[i:i + 5]]
            B, Ci, H, W = chunk[0].shape
            if Ci != 3 or any(tuple(t.shape) != (B, 3, H, W) for t in chunk):
                raise _lib.MovedepthHipError("pack_rgbx: images must be (B,3,H,W) of one size")
            res = [torch.empty(B, H, W, 4, device=chunk[0].device, dtype=torch.float32) for _ in chunk]
            n = len(chunk)
            _lib.call("md_pack_rgbx", (ctypes.c_void_p * n)(*[t.data_ptr() for t in chunk]), n, B, H, W,
                      (ctypes.c_void_p * n)(*[t.data_ptr() for t in res]), _stream())
            out += res
    return out


def _packed(imgs):
    """accept planar (B,3,H,W) or packed (B,H,W,4) images; pack the planar ones (one launch)"""
    imgs = list(imgs)
    todo = [i for i, t in enumerate(imgs) if not is_packed(t)]
    if todo:
        for i, p in zip(todo, pack_rgbx([imgs[i] for i in todo])):
            imgs[i] = p
    return imgs


def _fill_desc(d, cfg, target, srcs, Ts, K, invK, dzs, ident_min, noise, ext_mask):
    B, H, W, _ = target.shape     # packed (B,H,W,4)
    d.B, d.H, d.W, d.F, d.S = B, H, W, len(srcs), len(dzs)
    d.is_disp, d.identity, d.mvs_mode, d.no_ssim = int(cfg["is_disp"]), int(cfg["identity"]), int(cfg["mvs_mode"]), int(cfg["no_ssim"])
    d.ssim_w, d.min_depth, d.max_depth = float(cfg["ssim_w"]), float(cfg["min_depth"]), float(cfg["max_depth"])
    d.target, d.K, d.invK = _ptr(target), _ptr(K), _ptr(invK)
    d.ident_min, d.noise, d.ext_mask = _ptr(ident_min), _ptr(noise), _ptr(ext_mask)
    for f, (im, T) in enumerate(zip(srcs, Ts)):
        d.src[f], d.T[f] = _ptr(im), _ptr(T)
    for si, z in enumerate(dzs):
        if z is None:
            continue
        d.dz[si] = _ptr(z)
        d.dh[si], d.dw[si] = (z.shape[-2], z.shape[-1]) if cfg["is_disp"] else (H, W)


class _PhotoLoss(torch.autograd.Function):
    """md_photo_fwd / md_photo_bwd: one launch each way for all scales and frames of one group of photometric losses."""

    @staticmethod
    def forward(ctx, cfg, target, K, invK, ident_min, noise, ext_mask, *rest):
        F, S = cfg["F"], cfg["S"]
        srcs = list(rest[:F])             # packed (B,H,W,4), see photometric_loss
        Ts = [_prep(t, "T").reshape(-1, 4, 4) for t in rest[F:2 * F]]
        dzs = [_prep(t, "disp / depth") for t in rest[2 * F:2 * F + S]]
        K, invK = _prep(K, "K"), _prep(invK, "inv_K")
        ident_min, noise, ext_mask = _prep(ident_min, "identity loss"), _prep(noise, "noise"), _prep(ext_mask, "mask")
        B, H, W, _ = target.shape
        dev, f32 = target.device, torch.float32
        for z in dzs:
            if not cfg["is_disp"] and z.numel() != B * H * W:
                raise RuntimeError("depth has %d elements, expected B*H*W=%d" % (z.numel(), B * H * W))
        if noise is not None and noise.numel() != S * B * H * W:
            raise RuntimeError("noise must have S*B*H*W = %d elements (one draw per scale, trainer.py:698)" % (S * B * H * W))
        d = _lib.PhotoDesc()
        _fill_desc(d, cfg, target, srcs, Ts, K, invK, dzs, ident_min, noise, ext_mask)
        warped = torch.empty(S, F, B, H, W, 4, device=dev, dtype=f32)   # packed; handed out as (B,3,H,W) views
        pix = torch.empty(S, F, B, H, W, 2, device=dev, dtype=f32) if cfg["want_pix"] else None
        oob = torch.empty(F, B, H, W, device=dev, dtype=torch.uint8) if cfg["want_oob"] else None
        depth_out = torch.empty(S, B, 1, H, W, device=dev, dtype=f32) if cfg["is_disp"] else None
        mn = torch.empty(S, B, 1, H, W, device=dev, dtype=f32)
        mask = torch.empty(S, B, 1, H, W, device=dev, dtype=f32) if cfg["want_mask"] else None
        sel = torch.empty(S, B, H, W, device=dev, dtype=torch.uint8)
        loss2 = torch.empty(S, 2, device=dev, dtype=f32)
        for si in range(S):
            for f in range(F):
                d.warped[si][f] = warped[si, f].data_ptr()
                if pix is not None:
                    d.pix[si][f] = pix[si, f].data_ptr()
            if depth_out is not None:
                d.depth_out[si] = depth_out[si].data_ptr()
            d.mn[si], d.sel[si] = mn[si].data_ptr(), sel[si].data_ptr()
            if mask is not None:
                d.mask[si] = mask[si].data_ptr()
        if oob is not None:
            for f in range(F):
                d.oob[f] = oob[f].data_ptr()
        d.loss = loss2.data_ptr()
        ws = _ws(_lib.load().md_photo_fwd_ws_bytes(B, S, H, W), dev)
        _timed_call("md_photo_fwd", ctypes.byref(d), _p(ws), _stream())
        ctx.cfg = cfg
        ctx.shapes = [t.shape for t in rest[F:2 * F]], [t.shape for t in rest[2 * F:2 * F + S]]
        saved = [target, K, invK, warped, sel, loss2] + srcs + Ts + dzs
        ctx.has_mask = mask is not None and ext_mask is not None
        if ctx.has_mask:
            saved.append(mask)
        ctx.save_for_backward(*saved)
        ctx.set_materialize_grads(False)
        aux = [t for t in (warped, pix, oob, depth_out, mn, mask) if t is not None]
        ctx.mark_non_differentiable(*aux)
        ctx.n_aux = 6
        return tuple(loss2[si, 0] for si in range(S)) + (warped, pix, oob, depth_out, mn, mask)

    @staticmethod
    def backward(ctx, *grads):
        cfg = ctx.cfg
        F, S = cfg["F"], cfg["S"]
        saved = ctx.saved_tensors
        target, K, invK, warped, sel, loss2 = saved[:6]
        srcs, Ts, dzs = saved[6:6 + F], saved[6 + F:6 + 2 * F], saved[6 + 2 * F:6 + 2 * F + S]
        mask = saved[6 + 2 * F + S] if ctx.has_mask else None
        B, H, W, _ = target.shape
        d = _lib.PhotoDesc()
        _fill_desc(d, cfg, target, srcs, Ts, K, invK, dzs, None, None, None)
        gl = [None if g is None else g.reshape(1).contiguous().float() for g in grads[:S]]
        d_dz = [torch.empty_like(z) for z in dzs]
        need_T = [ctx.needs_input_grad[7 + F + f] for f in range(F)]
        d_T = [torch.empty(B, 4, 4, device=target.device, dtype=torch.float32) if need_T[f] else None for f in range(F)]
        for si in range(S):
            for f in range(F):
                d.warped[si][f] = warped[si, f].data_ptr()
            d.sel[si] = sel[si].data_ptr()
            if mask is not None:
                d.mask[si] = mask[si].data_ptr()
            d.gloss[si] = _ptr(gl[si])
            d.d_dz[si] = d_dz[si].data_ptr()
        for f in range(F):
            d.d_T[f] = _ptr(d_T[f])
        d.loss = loss2.data_ptr()
        ws = _ws(_lib.load().md_photo_bwd_ws_bytes(B, S, F, H, W, int(cfg["is_disp"])), target.device)
        _timed_call("md_photo_bwd", ctypes.byref(d), _p(ws), _stream())
        t_shapes, z_shapes = ctx.shapes
        out_T = [None if d_T[f] is None else d_T[f].reshape(t_shapes[f]) for f in range(F)]
        out_z = [d_dz[si].reshape(z_shapes[si]) if ctx.needs_input_grad[7 + 2 * F + si] else None for si in range(S)]
        return (None,) * 7 + (None,) * F + tuple(out_T) + tuple(out_z)


def photometric_loss(target, srcs, Ts, K, invK, depths, is_disp=False, min_depth=0.1, max_depth=100.0, ssim_w=0.85,
                     no_ssim=False, ident_min=None, noise=None, ext_mask=None, mvs_mode=False, want_pix=False,
                     want_oob=False, want_mask=False):
    """generate_images_pred + compute_losses' photometric part for one group of losses, in one launch each way
    (reference trainer.py:491-532 with 675-709 [mono, every scale], 498-509 with 621-662 [MVS], 569-612 [fused depth]).

    target (B,3,H,W) or packed (B,H,W,4) (pack_rgbx: the kernels read packed images; planar ones are packed here, so pack the
    frames once per step when several calls share them); srcs: F source frames, likewise; Ts: F poses (B,4,4); depths: S tensors -- disparity pyramid levels
    (B,1,h_s,w_s) when is_disp (up-sampled to HxW and converted with min/max_depth inside) or depth maps (B,[1,]H,W).
    ident_min (B,1,H,W): the identity loss of identity_loss(); noise (S,B,1,H,W): the 1e-5-scaled tie-break noise.
    Returns a dict: loss (list of S scalars, differentiable w.r.t. depths and Ts), warped[s][f], pix[s][f] | None,
    oob[f] | None, depth[s] | None (is_disp only), min[s] (B,1,H,W), mask[s] | None."""
    F, S = len(srcs), len(depths)
    if not (1 <= F <= _lib.PHOTO_MAX_FRAMES and 1 <= S <= _lib.PHOTO_MAX_SCALES):
        raise _lib.MovedepthHipError("photometric_loss: %d source frames / %d scales (1..4 each)" % (F, S))
    cfg = dict(F=F, S=S, is_disp=bool(is_disp), identity=False, mvs_mode=bool(mvs_mode), no_ssim=bool(no_ssim) or ssim_w == 0,
               ssim_w=float(ssim_w), min_depth=float(min_depth), max_depth=float(max_depth), want_pix=bool(want_pix),
               want_oob=bool(want_oob), want_mask=bool(want_mask) or ext_mask is not None)
    if not target.is_cuda:
        raise _lib.MovedepthHipError("photometric_loss: target must be a GPU tensor (got %s): the HIP path has no CPU fallback" % target.device)
    packed = _packed([target] + list(srcs))
    out = _PhotoLoss.apply(cfg, packed[0], K, invK, ident_min, noise, ext_mask, *packed[1:], *Ts, *depths)
    warped, pix, oob, depth_out, mn, mask = out[S:]
    return {"loss": list(out[:S]),
            "warped": [[packed_view(warped[s, f]) for f in range(F)] for s in range(S)],
            "pix": None if pix is None else [[pix[s, f] for f in range(F)] for s in range(S)],
            "oob": None if oob is None else [oob[f] for f in range(F)],
            "depth": None if depth_out is None else [depth_out[s] for s in range(S)],
            "min": [mn[s] for s in range(S)],
            "mask": None if mask is None else [mask[s] for s in range(S)]}


def identity_loss(target, srcs, ssim_w=0.85, no_ssim=False):
    """min over frames of compute_reprojection_loss(src_f, target) (reference trainer.py:690-696 / 592-599): the identity
    loss the auto-mask compares against, evaluated once per step (it does not depend on the scale) -> (B,1,H,W).  no_grad:
    its inputs are the input frames."""
    with torch.no_grad():
        if not target.is_cuda:
            raise _lib.MovedepthHipError("identity_loss: target must be a GPU tensor (got %s)" % target.device)
        packed = _packed([target] + list(srcs))
        target, srcs = packed[0], packed[1:]
        B, H, W, _ = target.shape
        cfg = dict(is_disp=False, identity=True, mvs_mode=False, no_ssim=bool(no_ssim) or ssim_w == 0, ssim_w=float(ssim_w),
                   min_depth=0.1, max_depth=100.0)
        d = _lib.PhotoDesc()
        _fill_desc(d, cfg, target, srcs, [None] * len(srcs), None, None, [None], None, None, None)
        d.S = 1
        out = torch.empty(B, 1, H, W, device=target.device, dtype=torch.float32)
        d.mn[0] = out.data_ptr()
        _timed_call("md_photo_fwd", ctypes.byref(d), None, _stream())
    return out


# --------------------------------------------------------------------------- smoothness
class _Smooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, img, normalize):
        disp, img = _prep(disp, "disp"), _prep(img, "img")
        B, Ci, h, w = img.shape
        loss = torch.empty(1, device=disp.device, dtype=torch.float32)
        ws = _ws(_lib.load().md_smooth_ws_bytes(B, h, w), disp.device)
        _lib.call("md_smooth_fwd", _p(disp), _p(img), B, Ci, h, w, int(normalize), _p(loss), _p(ws), _stream())
        ctx.save_for_backward(disp, img)
        ctx.normalize = int(normalize)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        disp, img = ctx.saved_tensors
        B, Ci, h, w = img.shape
        gl = g.reshape(1).contiguous().float()
        d = torch.empty_like(disp)
        ws = _ws(_lib.load().md_smooth_ws_bytes(B, h, w), disp.device)
        _lib.call("md_smooth_bwd", _p(gl), _p(disp), _p(img), B, Ci, h, w, ctx.normalize, _p(d), _p(ws), _stream())
        return d, None, None


def smooth_loss(disp, img, normalize=True):
    """get_smooth_loss(disp / (mean_hw(disp) + 1e-7), img) (reference trainer.py:712-714, layers.py:630-643);
    normalize=False is the bare get_smooth_loss.  Gradient to `disp` only."""
    return _Smooth.apply(disp, img, bool(normalize))


class _SmoothMulti(torch.autograd.Function):
    """get_smooth_loss of every pyramid level in one launch per pass (md_smooth_multi_*)."""

    @staticmethod
    def forward(ctx, normalize, S, *tensors):
        disps = [_prep(t, "disp") for t in tensors[:S]]
        imgs = [_prep(t, "img") for t in tensors[S:]]
        B, Ci = imgs[0].shape[:2]
        hs = (ctypes.c_int * S)(*[im.shape[2] for im in imgs])
        ws_ = (ctypes.c_int * S)(*[im.shape[3] for im in imgs])
        loss = torch.empty(S, device=disps[0].device, dtype=torch.float32)
        ws = _ws(_lib.load().md_smooth_multi_ws_bytes(B, S), disps[0].device)
        dp = (ctypes.c_void_p * S)(*[d.data_ptr() for d in disps])
        ip = (ctypes.c_void_p * S)(*[i.data_ptr() for i in imgs])
        _lib.call("md_smooth_multi_fwd", dp, ip, hs, ws_, S, B, Ci, int(normalize), _p(loss), _p(ws), _stream())
        ctx.save_for_backward(*disps, *imgs)
        ctx.meta = (int(normalize), S)
        ctx.set_materialize_grads(False)
        return tuple(loss[s] for s in range(S))

    @staticmethod
    def backward(ctx, *grads):
        normalize, S = ctx.meta
        saved = ctx.saved_tensors
        disps, imgs = saved[:S], saved[S:]
        B, Ci = imgs[0].shape[:2]
        gl = [None if g is None else g.reshape(1).contiguous().float() for g in grads]
        dd = [torch.empty_like(d) for d in disps]
        hs = (ctypes.c_int * S)(*[im.shape[2] for im in imgs])
        ws_ = (ctypes.c_int * S)(*[im.shape[3] for im in imgs])
        ws = _ws(_lib.load().md_smooth_multi_ws_bytes(B, S), disps[0].device)
        gp = (ctypes.c_void_p * S)(*[_ptr(g) for g in gl])
        dp = (ctypes.c_void_p * S)(*[d.data_ptr() for d in disps])
        ip = (ctypes.c_void_p * S)(*[i.data_ptr() for i in imgs])
        op = (ctypes.c_void_p * S)(*[d.data_ptr() for d in dd])
        _lib.call("md_smooth_multi_bwd", gp, dp, ip, hs, ws_, S, B, Ci, normalize, op, _p(ws), _stream())
        return (None, None) + tuple(dd) + (None,) * S


def smooth_losses(disps, imgs, normalize=True):
    """smooth_loss(disps[s], imgs[s]) for every pyramid level s (reference trainer.py:712-714 in its scale loop), one launch
    per pass for all of them.  Returns a list of scalars."""
    S = len(disps)
    if not (1 <= S <= _lib.PHOTO_MAX_SCALES) or len(imgs) != S:
        raise _lib.MovedepthHipError("smooth_losses: %d disparity levels / %d images (1..4, equal)" % (S, len(imgs)))
    return list(_SmoothMulti.apply(bool(normalize), S, *disps, *imgs))


# --------------------------------------------------------------------------- post-volume regression
class _SoftmaxEntropyLocalmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, min_inv, max_inv, radius, want_prob):
        logits = _prep(logits, "logits")
        min_inv, max_inv = _prep(min_inv, "min_depth_inverse"), _prep(max_inv, "max_depth_inverse")
        B, D, h, w = logits.shape
        prob = torch.empty_like(logits) if want_prob else None
        ent = torch.empty(B, 1, h, w, device=logits.device, dtype=torch.float32)
        depth = torch.empty(B, h, w, device=logits.device, dtype=torch.float32)
        _lib.call("md_softmax_entropy_localmax_fwd", _p(logits), B, D, h, w, int(radius), _p(min_inv), _p(max_inv),
                  _p(prob), _p(ent), _p(depth), _stream())
        ctx.save_for_backward(logits, min_inv, max_inv)
        ctx.radius = int(radius)
        if want_prob:
            ctx.mark_non_differentiable(prob)
        return depth, ent, prob

    @staticmethod
    def backward(ctx, g_depth, g_ent, _gp):
        logits, min_inv, max_inv = ctx.saved_tensors
        B, D, h, w = logits.shape
        gd = None if g_depth is None else g_depth.contiguous().float()
        ge = None if g_ent is None else g_ent.contiguous().float()
        d = torch.empty_like(logits)
        _lib.call("md_softmax_entropy_localmax_bwd", _p(gd), _p(ge), _p(logits), B, D, h, w, ctx.radius, _p(min_inv),
                  _p(max_inv), _p(d), _stream())
        return d, None, None, None, None


def softmax_entropy_localmax(logits, min_depth_inverse, max_depth_inverse, radius=1, want_prob=False):
    """F.softmax(logits, 1) -> entropy(dim=1, keepdim) + localmax(...) in one pass (reference trainer.py:367-371).
    Returns (depth (B,h,w), entropy (B,1,h,w), prob (B,D,h,w) or None [detached])."""
    return _SoftmaxEntropyLocalmax.apply(logits, min_depth_inverse, max_depth_inverse, radius, want_prob)


# --------------------------------------------------------------------------- convex upsample / standalone geometry
class _ConvexUpsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, mask, scale):
        depth, mask = _prep(depth, "depth"), _prep(mask, "mask")
        B, h, w = depth.shape[0], depth.shape[-2], depth.shape[-1]
        s = 2 ** scale
        out = torch.empty(B, s * h, s * w, device=depth.device, dtype=torch.float32)
        _lib.call("md_convex_upsample_fwd", _p(depth), _p(mask), B, h, w, int(scale), _p(out), _stream())
        ctx.save_for_backward(depth, mask)
        ctx.scale = int(scale)
        return out

    @staticmethod
    def backward(ctx, g):
        depth, mask = ctx.saved_tensors
        B, h, w = depth.shape[0], depth.shape[-2], depth.shape[-1]
        g = g.contiguous().float()
        d_depth, d_mask = torch.empty_like(depth), torch.empty_like(mask)
        ws = _ws(_lib.load().md_convex_upsample_bwd_ws_bytes(B, h, w), depth.device)
        _lib.call("md_convex_upsample_bwd", _p(g), _p(depth), _p(mask), B, h, w, ctx.scale, _p(d_depth), _p(d_mask),
                  _p(ws), _stream())
        return d_depth, d_mask, None


def convex_upsample(depth, mask, scale=2):
    """reference layers.py:200-214: depth (B,h,w) or (B,1,h,w), mask (B,9*4**scale,h,w) -> (B, 2**scale*h, 2**scale*w)."""
    return _ConvexUpsample.apply(depth, mask, scale)


# --------------------------------------------------------------------------- reg3d's last layer (one output channel)
CONV3D_C1_CHANNELS = (8, 16)


def _c1_weight_strides(w):
    """(tap stride, channel stride) of a [1,C,3,3,3] weight whose 27 taps are evenly spaced in memory."""
    s = w.stride()
    if s[3] != 3 * s[4] or s[2] != 9 * s[4]:
        raise _lib.MovedepthHipError("conv3d_c1: weight strides %s are not tap-regular" % (tuple(s),))
    return s[4], s[1]


class _Conv3dC1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        if not x.is_cuda:
            raise _lib.MovedepthHipError("conv3d_c1: x must be a GPU tensor (got %s)" % x.device)
        B, C, D, H, W = x.shape
        if tuple(weight.shape) != (1, C, 3, 3, 3):
            raise _lib.MovedepthHipError("conv3d_c1: weight %s does not match x %s" % (tuple(weight.shape), tuple(x.shape)))
        x = x.float().contiguous(memory_format=torch.channels_last_3d)      # no copy when already NDHWC
        weight = weight.float()
        wsk, wsc = _c1_weight_strides(weight)
        y = torch.empty(B, 1, D, H, W, device=x.device, dtype=torch.float32)
        _timed_call("md_conv3d_c1_fwd", _p(x), _p(weight), wsk, wsc, _p(y), B, C, D, H, W, _stream())
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        B, C, D, H, W = x.shape
        gy = gy.float().contiguous()
        dx = dw = None
        # Weight gradient FIRST: it streams x (283 MB at config 2) from HBM, and launched behind the data gradient -- which has
        # just written as much, leaving the caches full of dirty lines to write back -- it measured 94 us in the training step
        # against 63 us stand-alone (profiles/r02_conv_c1_*).  The data gradient only writes and does not care who ran before.
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)                                   # preserves the weight's strides
            dsk, dsc = _c1_weight_strides(dw)
            nbytes = _lib.load().md_conv3d_c1_bwd_weight_ws_bytes(B, C, D, H, W)
            ws = _ws(nbytes, x.device)
            _timed_call("md_conv3d_c1_bwd_weight", _p(x), _p(gy), _p(dw), dsk, dsc, _p(ws), int(nbytes), B, C, D, H, W,
                        _stream())
        if ctx.needs_input_grad[0]:
            wsk, wsc = _c1_weight_strides(weight)
            dx = torch.empty_like(x, memory_format=torch.channels_last_3d)
            _timed_call("md_conv3d_c1_bwd_data", _p(gy), _p(weight), wsk, wsc, _p(dx), B, C, D, H, W, _stream())
        return dx, dw


def conv3d_c1(x, weight):
    """nn.Conv3d(C, 1, 3, stride=1, padding=1, bias=False) -- reg3d.prob, reference networks/resnet_encoder.py:254,277.
    x (B,C,D,H,W), C in CONV3D_C1_CHANNELS, read as channels_last_3d storage; weight (1,C,3,3,3) in either memory
    format -> (B,1,D,H,W).  Gradients to x (channels_last_3d) and weight."""
    return _Conv3dC1.apply(x, weight)


# --------------------------------------------------------------------------- reg3d's first layer (16 -> 16): weight gradient
_CONV_ARGS = ([1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1)  # stride, padding, dilation, transposed, out pad, groups


def _c16_weight_strides(w):
    s = w.stride()
    if s[3] != 3 * s[4] or s[2] != 9 * s[4]:
        raise _lib.MovedepthHipError("conv3d_16: weight strides %s are not tap-regular" % (tuple(s),))
    return s[0], s[1], s[4]


def _is_planar(t):
    """plain contiguous [B,C,D,H,W] (the cost volume's `bgd` storage) as opposed to channels_last_3d"""
    return t.is_contiguous() and not t.is_contiguous(memory_format=torch.channels_last_3d)


def _w_for_forward(w):
    """The weight with its OUTPUT-channel stride 1 (storage order ci, tap, co).  A workgroup of the bf16 x 3 kernels gathers its 120 weights
    per lane straight from memory, lane = output channel (forward) or input channel (data gradient): with the lane index strided by
    Ci * 27 floats that prologue cost the forward 28 us of 387 at 16 -> 16 and 80 of 318 at 32 -> 32 (profiles/r05_conv3d_channel_blocks.txt); a
    transposed copy of the 27-110 KB tensor is one small kernel."""
    if w.stride(0) == 1:
        return w
    return w.permute(1, 0, 2, 3, 4).contiguous(memory_format=torch.channels_last_3d).permute(1, 0, 2, 3, 4)


def _w_for_dgrad(w):
    """... and with its INPUT-channel stride 1 (channels_last_3d storage: what a channels-last module holds anyway)."""
    return w if w.stride(1) == 1 else w.contiguous(memory_format=torch.channels_last_3d)


class _Conv3d16(torch.autograd.Function):
    """reg3d.conv0's convolution on the MFMA kernels of csrc/conv3d_c16.hip.  x may be planar (contiguous
    [B,16,D,H,W], what md_costvol_fwd writes fastest) or channels_last_3d; y is channels_last_3d; dx has x's layout.
    `lib_fwd_dgrad` keeps the forward and the data gradient on the library convolution (A/B switch; the weight
    gradient is always the hand-written one)."""

    @staticmethod
    def forward(ctx, x, weight, lib_fwd_dgrad):
        ctx.lib = bool(lib_fwd_dgrad)
        planar = _is_planar(x) and not ctx.lib
        if not planar:
            x = x.contiguous(memory_format=torch.channels_last_3d)
        ctx.planar = planar
        ctx.save_for_backward(x, weight)
        if ctx.lib:
            return torch.ops.aten.convolution(x, weight, None, *_CONV_ARGS)
        B, C, D, H, W = x.shape
        y = torch.empty((B, weight.shape[0], D, H, W), device=x.device, dtype=torch.float32,
                        memory_format=torch.channels_last_3d)
        wf = _w_for_forward(weight)
        _timed_call("md_conv3d_c16_fwd", _p(x), int(planar), _p(wf), *_c16_weight_strides(wf), _p(y), B, C,
                    weight.shape[0], D, H, W, _stream())
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        B, C, D, H, W = x.shape
        gy = gy.float().contiguous(memory_format=torch.channels_last_3d)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if ctx.lib:
                dx = torch.ops.aten.convolution_backward(gy, x, weight, None, *_CONV_ARGS, [True, False, False])[0]
            else:
                dx = torch.empty_like(x, memory_format=torch.contiguous_format if ctx.planar else torch.channels_last_3d)
                wd = _w_for_dgrad(weight)
                _timed_call("md_conv3d_c16_bwd_data", _p(gy), _p(wd), *_c16_weight_strides(wd), _p(dx),
                            int(ctx.planar), B, C, weight.shape[0], D, H, W, _stream())
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)  # keeps the weight's strides
            nbytes = _lib.load().md_conv3d_c16_bwd_weight_ws_bytes(B, D, H, W)
            ws = _ws(nbytes, x.device)
            _timed_call("md_conv3d_c16_bwd_weight", _p(x), int(ctx.planar), _p(gy), _p(dw), *_c16_weight_strides(dw), _p(ws),
                        int(nbytes), B, C, weight.shape[0], D, H, W, _stream())
        return dx, dw, None


def conv3d_16(x, weight, lib_fwd_dgrad=False):
    """nn.Conv3d(16, 16, 3, stride=1, padding=1, bias=False) -- reg3d.conv0's convolution (reference
    networks/resnet_encoder.py:231,258).  x (B,16,D,H,W) on the GPU, planar (contiguous) or channels_last_3d, read in
    place; weight (16,16,3,3,3) in either memory format; returns a channels_last_3d (B,16,D,H,W) tensor."""
    if not x.is_cuda or tuple(weight.shape) != (16, 16, 3, 3, 3) or x.shape[1] != 16:
        raise _lib.MovedepthHipError("conv3d_16: needs a GPU tensor with 16 channels and a (16,16,3,3,3) weight, got %s %s %s"
                                     % (x.device, tuple(x.shape), tuple(weight.shape)))
    return _Conv3d16.apply(x.float(), weight.float(), lib_fwd_dgrad)


# --------------------------------------------------------------------------- reg3d's interior layers (Ci, Co multiples of 16)
class _Conv3dCB(torch.autograd.Function):
    """3 x 3 x 3, stride 1, padding 1, no bias, Ci and Co multiples of 16, on the bf16 x 3 kernels of csrc/conv3d_c16.hip as sums
    over 16 x 16 channel blocks (md_conv3d_cb_*).  Channels_last_3d in and out."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous(memory_format=torch.channels_last_3d)
        ctx.save_for_backward(x, weight)
        B, C, D, H, W = x.shape
        y = torch.empty((B, weight.shape[0], D, H, W), device=x.device, dtype=torch.float32, memory_format=torch.channels_last_3d)
        wf = _w_for_forward(weight)
        _timed_call("md_conv3d_cb_fwd", _p(x), _p(wf), *_c16_weight_strides(wf), _p(y), B, C, weight.shape[0], D, H, W, _stream())
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        B, C, D, H, W = x.shape
        gy = gy.float().contiguous(memory_format=torch.channels_last_3d)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x, memory_format=torch.channels_last_3d)
            wd = _w_for_dgrad(weight)
            _timed_call("md_conv3d_cb_bwd_data", _p(gy), _p(wd), *_c16_weight_strides(wd), _p(dx), B, C, weight.shape[0], D, H, W,
                        _stream())
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)  # keeps the weight's strides
            nbytes = _lib.load().md_conv3d_cb_bwd_weight_ws_bytes(B, C, weight.shape[0], D, H, W)
            ws = _ws(nbytes, x.device)
            _timed_call("md_conv3d_cb_bwd_weight", _p(x), _p(gy), _p(dw), *_c16_weight_strides(dw), _p(ws), int(nbytes), B, C,
                        weight.shape[0], D, H, W, _stream())
        return dx, dw


def conv3d_cb_supported(x, weight):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.dim() == 5
            and tuple(weight.shape[2:]) == (3, 3, 3) and weight.shape[0] % 16 == 0 and weight.shape[1] % 16 == 0
            and weight.shape[0] <= 256 and weight.shape[1] <= 256 and x.shape[1] == weight.shape[1])


def conv3d_cb(x, weight):
    """nn.Conv3d(Ci, Co, 3, stride=1, padding=1, bias=False) with Ci, Co multiples of 16 -- reg3d's conv2 / conv4 / conv6 (reference
    networks/resnet_encoder.py:233-239, applied :260-262).  x (B,Ci,D,H,W) fp32 on the GPU (read as channels_last_3d), weight (Co,Ci,3,3,3) in either memory
    format; returns a channels_last_3d (B,Co,D,H,W) tensor.  Products on the bf16 matrix pipe with three-piece operands: fp32 results to
    ~4e-7 relative."""
    if not conv3d_cb_supported(x, weight):
        raise _lib.MovedepthHipError("conv3d_cb: needs fp32 GPU tensors, a (Co,Ci,3,3,3) weight with Ci, Co multiples of 16 (<= 256), got %s %s %s"
                                     % (x.device, tuple(x.shape), tuple(weight.shape)))
    return _Conv3dCB.apply(x, weight)


# --------------------------------------------------------------------------- BatchNorm + ReLU (+ residual), 16 channels
class _BnRelu3d(torch.autograd.Function):
    """Training-mode BatchNorm3d + ReLU (+ residual add) over a channels_last_3d (B,16,D,H,W) tensor in 3 + 5 passes
    (csrc/bnrelu3d.hip).  `group`: a process group to all-reduce the statistics over (SyncBatchNorm semantics) or None."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, running_mean, running_var, momentum, eps, group):
        import torch.distributed as dist
        x = x.contiguous(memory_format=torch.channels_last_3d)
        if res is not None:
            res = res.contiguous(memory_format=torch.channels_last_3d)
        C = x.shape[1]
        nvox = x.numel() // C
        ws = _ws(_lib.load().md_bn_relu_ws_bytes(), x.device)
        sums = torch.empty(2 * C, device=x.device, dtype=torch.float64)
        _timed_call("md_bn_relu_stats", _p(x), nvox, C, _p(sums), _p(ws), _stream())
        n_total = nvox
        if group is not None:
            _group_all_reduce(sums, group)
            n_total = nvox * _group_size(group)
        stat = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        mean, invstd = stat[:C], stat[C:]
        rm = running_mean if (running_mean is not None and running_mean.dtype == torch.float32) else None
        rv = running_var if (running_var is not None and running_var.dtype == torch.float32) else None
        _lib.call("md_bn_relu_finalize", _p(sums), n_total, C, float(eps), float(momentum), _p(mean), _p(invstd), _p(rm), _p(rv),
                  _stream())
        y = torch.empty_like(x, memory_format=torch.channels_last_3d)
        w, b = weight.float().contiguous(), bias.float().contiguous()
        _timed_call("md_bn_relu_apply", _p(x), _p(mean), _p(invstd), _p(w), _p(b), _p(res), nvox, C, _p(y), _stream())
        ctx.save_for_backward(x, mean, invstd, w, b)
        ctx.group, ctx.has_res = group, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        x, mean, invstd, w, b = ctx.saved_tensors
        C = x.shape[1]
        nvox = x.numel() // C
        dy = dy.float().contiguous(memory_format=torch.channels_last_3d)
        ws = _ws(_lib.load().md_bn_relu_ws_bytes(), x.device)
        sums = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        _timed_call("md_bn_relu_bwd_reduce", _p(dy), _p(x), _p(mean), _p(invstd), _p(w), _p(b), nvox, C, _p(sums), _p(ws),
                    _stream())
        d_beta, d_gamma = sums[:C], sums[C:]                             # local sums (the gradient reducer averages them)
        n_total = nvox
        if ctx.group is not None:
            local = sums
            sums = local.clone()
            _group_all_reduce(sums, ctx.group)
            n_total = nvox * _group_size(ctx.group)
        dx = torch.empty_like(x, memory_format=torch.channels_last_3d)
        _timed_call("md_bn_relu_bwd_dx", _p(dy), _p(x), _p(mean), _p(invstd), _p(w), _p(b), _p(sums), n_total, nvox, C, _p(dx),
                    _stream())
        return dx, d_gamma, d_beta, (dy if ctx.has_res else None), None, None, None, None, None


BN_RELU_CHANNELS = (16, 32, 64)


def bn_relu_3d(x, weight, bias, res=None, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, group=None):
    """relu(batch_norm(x)) [+ res] in training mode (batch statistics; running statistics updated in place):
    reference networks/resnet_encoder.py:231 (ConvBnReLU3D), :249-252 + :264 (conv11 and the skip connection).
    x (B,16,D,H,W) on the GPU, read as channels_last_3d."""
    if not x.is_cuda or x.shape[1] not in BN_RELU_CHANNELS:
        raise _lib.MovedepthHipError("bn_relu_3d: needs a GPU tensor with 16, 32 or 64 channels, got %s %s" % (x.device, tuple(x.shape)))
    return _BnRelu3d.apply(x.float(), weight, bias, res, running_mean, running_var, float(momentum), float(eps), group)


# --------------------------------------------------------------------------- BatchNorm with synchronised statistics
# The model calls this 115 times per step and direction and the step is nearly host-bound (kernel time ~ wall time): everything
# below is written for few Python operations per call -- raw pointers as ints (ctypes converts them by the argtypes), one
# allocation for the per-call scalars, no dtype / layout conversions when the tensor already is what the kernels read.
_BN_WS = {}
_BN_FN = None
_CL_FORMAT = {4: torch.channels_last, 5: torch.channels_last_3d}


def _bn_fns():
    global _BN_FN
    if _BN_FN is None:
        lib = _lib.load()
        _BN_FN = (lib.md_bn_stats, lib.md_bn_apply, lib.md_bn_bwd_reduce, lib.md_bn_bwd_dx)
    return _BN_FN


def _bn_ws(device, stream):
    """The reduction kernels' workspace: zeroed once, left zeroed by every launch (one per device and stream) -> raw pointer"""
    key = (device.index, stream)
    ws = _BN_WS.get(key)
    if ws is None:
        t = torch.zeros(int(_lib.load().md_bn_ws_bytes()), dtype=torch.uint8, device=device)
        ws = _BN_WS[key] = (t, t.data_ptr())
    return ws[1]


GROUP_ALL_REDUCE = [0, 0]   # statistics all-reduces issued by the BatchNorm layers of this process: calls, bytes (bench.py config.collectives)


def _group_all_reduce(t, group):
    """in-place sum over `group`: a rccl_direct.DirectAllReduce (RCCL on the current stream) or a torch.distributed group"""
    GROUP_ALL_REDUCE[0] += 1
    GROUP_ALL_REDUCE[1] += t.numel() * t.element_size()
    if callable(group):
        group(t)
    else:
        import torch.distributed as dist
        dist.all_reduce(t, group=group)


def _group_size(group):
    if callable(group):
        return group.size
    import torch.distributed as dist
    return dist.get_world_size(group)


_BN_DTYPE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}   # `dtype` of md_bn_* (include/movedepth_hip.h)


def _rows_channels_last(x, dtype=None):
    """x (N,C,*spatial) -> (tensor stored channels-last, rows, C); float32, bf16 and fp16 (what torch.autocast hands a BatchNorm) stay
    as they are -- or become `dtype` --, anything else becomes float32; a copy only if it is not stored that way already"""
    want = dtype if dtype is not None else (x.dtype if x.dtype in _BN_DTYPE else torch.float32)
    if x.dtype is not want:
        x = x.to(want)
    nd = x.dim()
    if nd == 2:
        if not x.is_contiguous():
            x = x.contiguous()
    else:
        fmt = _CL_FORMAT.get(nd)
        if fmt is None:
            raise _lib.MovedepthHipError("sync_batch_norm: %d-D input unsupported (2-D, 4-D or 5-D)" % nd)
        if not x.is_contiguous(memory_format=fmt):
            x = x.contiguous(memory_format=fmt)
    C = x.shape[1]
    return x, x.numel() // C, C


def _f32c(t):
    return t if (t.dtype is torch.float32 and t.is_contiguous()) else t.float().contiguous()


class _SyncBatchNorm(torch.autograd.Function):
    """Training-mode batch normalisation over the global batch of a process group (csrc/syncbn.hip): statistics -> ONE all-reduce
    of 2C doubles -> apply, and in the backward two sums -> ONE all-reduce of 2C floats -> dx.  group None: this process only."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, group):
        x, rows, C = _rows_channels_last(x)
        dev = x.device
        stream = torch._C._cuda_getCurrentRawStream(dev.index)
        f_stats, f_apply, _, _ = _bn_fns()
        # one allocation: 2C doubles (the sums) followed by 2C floats (mean, invstd for the backward)
        buf = torch.empty(3 * C, device=dev, dtype=torch.float64)
        sums = buf[:2 * C]
        stat = buf[2 * C:].view(torch.float32)
        dt = _BN_DTYPE[x.dtype]
        rc = f_stats(x.data_ptr(), dt, rows, C, sums.data_ptr(), _bn_ws(dev, stream), stream)
        if rc:
            _lib.check(rc, "md_bn_stats")
        n_total = rows
        if group is not None:
            import torch.distributed as dist
            # every rank contributes the same number of rows (the per-GPU batch is fixed: the loaders drop the last batch)
            _group_all_reduce(sums, group)
            n_total = rows * _group_size(group)
        y = torch.empty_like(x)   # preserves the channels-last strides
        w, b = _f32c(weight), _f32c(bias)
        rm = running_mean.data_ptr() if (running_mean is not None and running_mean.dtype is torch.float32) else None
        rv = running_var.data_ptr() if (running_var is not None and running_var.dtype is torch.float32) else None
        rc = f_apply(x.data_ptr(), dt, sums.data_ptr(), n_total, eps, momentum, w.data_ptr(), b.data_ptr(), relu, rm, rv, stat.data_ptr(),
                     y.data_ptr(), rows, C, stream)
        if rc:
            _lib.check(rc, "md_bn_apply")
        ctx.save_for_backward(x, stat, w, b)
        ctx.group, ctx.relu = group, relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stat, w, b = ctx.saved_tensors
        dy, rows, C = _rows_channels_last(dy, x.dtype)
        dt = _BN_DTYPE[x.dtype]
        dev = x.device
        stream = torch._C._cuda_getCurrentRawStream(dev.index)
        _, _, f_reduce, f_dx = _bn_fns()
        # 2C sums of this rank (= d_beta, d_gamma: the gradient reducer averages parameter gradients) and, in a group, a second
        # copy written by the same kernel for the all-reduce
        grouped = ctx.group is not None
        buf = torch.empty(4 * C if grouped else 2 * C, device=dev, dtype=torch.float32)
        sums = buf[:2 * C]
        gsums = buf[2 * C:] if grouped else sums
        rc = f_reduce(dy.data_ptr(), x.data_ptr(), dt, stat.data_ptr(), w.data_ptr(), b.data_ptr(), ctx.relu, rows, C, sums.data_ptr(),
                      gsums.data_ptr() if grouped else None, _bn_ws(dev, stream), stream)
        if rc:
            _lib.check(rc, "md_bn_bwd_reduce")
        d_beta, d_gamma = sums[:C], sums[C:]
        n_total = rows
        if grouped:
            import torch.distributed as dist
            _group_all_reduce(gsums, ctx.group)
            n_total = rows * _group_size(ctx.group)
        sums = gsums
        dx = torch.empty_like(x)
        rc = f_dx(dy.data_ptr(), x.data_ptr(), dt, stat.data_ptr(), w.data_ptr(), b.data_ptr(), ctx.relu, sums.data_ptr(), n_total, rows, C,
                  dx.data_ptr(), stream)
        if rc:
            _lib.check(rc, "md_bn_bwd_dx")
        return dx, d_gamma, d_beta, None, None, None, None, None, None


def sync_batch_norm_supported(x):
    return x.is_cuda and x.dim() in (2, 4, 5) and x.shape[1] % 4 == 0 and 4 <= x.shape[1] <= 4096 and \
        (256 % (x.shape[1] // 4) == 0 or x.shape[1] // 4 <= 1024)


def sync_batch_norm(x, weight, bias, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, relu=False, group=None):
    """F.batch_norm(training=True) [+ ReLU] with the statistics taken over `group`'s global batch (torch.nn.SyncBatchNorm,
    which the reference's --ddp path converts every BatchNorm to: trainer.py:69-135).  x: (N,C,*spatial) on the GPU; the
    result is stored channels-last.  Running statistics are updated in place."""
    if not sync_batch_norm_supported(x):
        raise _lib.MovedepthHipError("sync_batch_norm: needs a GPU tensor (N,C,...) with C a multiple of 4 <= 4096, got %s %s" % (x.device, tuple(x.shape)))
    return _SyncBatchNorm.apply(x, weight, bias, running_mean, running_var, float(momentum), float(eps), int(bool(relu)), group)


def batch_norm_eval(x, weight, bias, running_mean, running_var, eps=1e-5, relu=False):
    """evaluation-mode BatchNorm [+ ReLU] from the running statistics, one kernel; no gradient"""
    with torch.no_grad():
        x, rows, C = _rows_channels_last(x)
        y = torch.empty_like(x)
        _lib.call("md_bn_eval", _p(x), _BN_DTYPE[x.dtype], _p(running_mean.float()), _p(running_var.float()), float(eps), _p(weight.float().contiguous()),
                  _p(bias.float().contiguous()), int(relu), _p(y), rows, C, _stream())
    return y


# --------------------------------------------------------------------------- pose parameters -> 4x4
class _PoseMatrix(torch.autograd.Function):
    @staticmethod
    def forward(ctx, axisangle, translation, invert):
        aa, tr = _prep(axisangle, "axisangle").reshape(-1, 3), _prep(translation, "translation").reshape(-1, 3)
        if aa.shape != tr.shape:
            raise _lib.MovedepthHipError("pose_matrix: axisangle %s and translation %s differ" % (tuple(axisangle.shape),
                                                                                               tuple(translation.shape)))
        B = aa.shape[0]
        T = torch.empty(B, 4, 4, device=aa.device, dtype=torch.float32)
        _lib.call("md_pose_matrix_fwd", _p(aa), _p(tr), B, int(bool(invert)), _p(T), _stream())
        ctx.save_for_backward(aa, tr)
        ctx.invert = int(bool(invert))
        ctx.shapes = (axisangle.shape, translation.shape)
        return T

    @staticmethod
    def backward(ctx, gT):
        aa, tr = ctx.saved_tensors
        gT = gT.contiguous().float()
        d_aa, d_tr = torch.empty_like(aa), torch.empty_like(tr)
        _lib.call("md_pose_matrix_bwd", _p(gT), _p(aa), _p(tr), aa.shape[0], ctx.invert, _p(d_aa), _p(d_tr), _stream())
        return d_aa.reshape(ctx.shapes[0]), d_tr.reshape(ctx.shapes[1]), None


def pose_matrix(axisangle, translation, invert=False):
    """transformation_from_parameters (reference layers.py:412-429): (B,1,3) or (B,3) each -> (B,4,4), differentiable."""
    return _PoseMatrix.apply(axisangle, translation, invert)


def backproject(depth, inv_K, batch_size, height, width):
    """BackprojectDepth.forward (reference layers.py:581-586), forward only -> (Bs,4,h*w)."""
    with torch.no_grad():
        depth, inv_K = _prep(depth, "depth"), _prep(inv_K, "inv_K")
        if depth.numel() != batch_size * height * width:
            raise RuntimeError("depth has %d elements, module was built for %d x %d x %d" %
                               (depth.numel(), batch_size, height, width))
        nk = inv_K.reshape(-1, 16).shape[0]
        out = torch.empty(batch_size, 4, height * width, device=depth.device, dtype=torch.float32)
        _lib.call("md_backproject", _p(depth), _p(inv_K), batch_size, nk, height, width, _p(out), _stream())
    return out


def project3d(points, K, T, batch_size, height, width, eps=1e-7):
    """Project3D.forward (reference layers.py:601-621), forward only -> (Bs,h,w,2)."""
    with torch.no_grad():
        points, K, T = _prep(points, "points"), _prep(K, "K"), _prep(T, "T")
        nk = K.reshape(-1, 16).shape[0]
        if T.reshape(-1, 16).shape[0] != nk:
            T = T.expand(nk, 4, 4).contiguous() if T.reshape(-1, 16).shape[0] == 1 else T
        out = torch.empty(batch_size, height, width, 2, device=points.device, dtype=torch.float32)
        _lib.call("md_project3d", _p(points), _p(K), _p(T), batch_size, nk, height, width, float(eps), _p(out), _stream())
    return out
