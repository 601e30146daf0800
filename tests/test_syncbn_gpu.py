"""networks.HipSyncBatchNorm / ops.sync_batch_norm (csrc/syncbn.hip): training-mode BatchNorm2d / BatchNorm3d with statistics
over the global batch -- what the reference's --ddp path gets from torch.nn.SyncBatchNorm (trainer.py:69-135).
(1) one process: output, every gradient and the running statistics against the same torch ops in float64;
(2) two ranks sharing cuda:0 over gloo: equal to the single-process result on the concatenated batch;
(3) the state_dict of a converted model is the unconverted model's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import assert_close, assert_close_knife_edge

pytestmark = pytest.mark.gpu


def host(t):
    return t.detach().float().cpu().numpy()


def _fmt(shape):
    return {2: torch.contiguous_format, 4: torch.channels_last, 5: torch.channels_last_3d}[len(shape)]


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("shape,stored_cl", [((2, 8, 5, 7), True), ((3, 64, 24, 40), True), ((2, 512, 6, 20), True), ((2, 2048, 3, 5), True),
                                             ((2, 16, 5, 7, 9), True), ((4, 12, 3, 3), True), ((2, 384, 4, 4), True), ((6, 64, 96, 320), True),
                                             ((2, 32, 9, 11), False), ((5, 24), True)])
def test_sync_batch_norm_vs_torch_fp64(shape, stored_cl, relu):
    from movedepth_amd import ops

    torch.manual_seed(3)
    C = shape[1]
    x = torch.randn(*shape, device="cuda") * 1.5 + 0.3
    if stored_cl:
        x = x.contiguous(memory_format=_fmt(shape))
    gy = torch.randn(*shape, device="cuda")
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.2
    rm_a, rv_a = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    xa, ga, ba = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ya = ops.sync_batch_norm(xa, ga, ba, rm_a, rv_a, 0.1, 1e-5, relu=relu)
    assert ya.shape == x.shape
    if len(shape) > 2:
        assert ya.is_contiguous(memory_format=_fmt(shape))
    ya.backward(gy)
    xb, gb, bb = (t.double().clone().requires_grad_(True) for t in (x, gamma, beta))
    rm_b, rv_b = torch.zeros(C, device="cuda", dtype=torch.float64), torch.ones(C, device="cuda", dtype=torch.float64)
    yb = torch.nn.functional.batch_norm(xb, rm_b, rv_b, gb, bb, True, 0.1, 1e-5)
    if relu:
        yb = torch.relu(yb)
    yb.backward(gy.double())
    n = x.numel() // C
    assert_close(host(ya), host(yb), what="y")
    # a ReLU pre-activation within float rounding of zero may fall on the other side: a 1e-6 fraction of outliers at 12 M elements
    assert_close_knife_edge(host(xa.grad), host(xb.grad), rtol=2e-4 if n < 100 else 1e-4, max_outlier_frac=1e-6 if (relu and n > 1e5) else 0.0, what="d_x")
    rt = 5e-4 if (relu and n > 1e5) else 1e-4
    assert_close(host(ga.grad), host(gb.grad), rtol=rt, what="d_gamma")
    assert_close(host(ba.grad), host(bb.grad), rtol=rt, what="d_beta")
    assert_close(host(rm_a), host(rm_b), rtol=1e-5, what="running_mean")
    assert_close(host(rv_a), host(rv_b), rtol=1e-5, what="running_var")
    # the same launch again on the same workspace (the reduction kernels must leave their ticket counter at zero)
    xc = x.clone().requires_grad_(True)
    yc = ops.sync_batch_norm(xc, gamma, beta, None, None, 0.1, 1e-5, relu=relu)
    assert torch.equal(yc, ya)


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(3, 64, 24, 40), (2, 16, 5, 7, 9), (2, 8, 64, 128), (2, 512, 6, 20)])
def test_sync_batch_norm_half_io_vs_torch_fp64(shape, dtype, relu):
    """bf16 / fp16 activations and gradients (what torch.autocast hands a BatchNorm layer): the kernels read and write the 2-byte
    type directly, statistics and parameter gradients in fp32 / double.  Reference: the same op in float64 ON THE ROUNDED INPUTS;
    outputs within one rounding step of the 2-byte type (8 bits of mantissa for bf16, 11 for fp16), sums within 1e-4."""
    from movedepth_amd import ops

    torch.manual_seed(5)
    C = shape[1]
    x = (torch.randn(*shape, device="cuda") * 1.5 + 0.3).to(dtype).contiguous(memory_format=_fmt(shape))
    gy = torch.randn(*shape, device="cuda").to(dtype)
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.2
    rm_a, rv_a = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    xa, ga, ba = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ya = ops.sync_batch_norm(xa, ga, ba, rm_a, rv_a, 0.1, 1e-5, relu=relu)
    assert ya.dtype is dtype and ya.shape == x.shape and ya.is_contiguous(memory_format=_fmt(shape))
    ya.backward(gy)
    assert xa.grad.dtype is dtype and ga.grad.dtype is torch.float32
    xb, gb, bb = (t.double().clone().requires_grad_(True) for t in (x, gamma, beta))
    rm_b, rv_b = torch.zeros(C, device="cuda", dtype=torch.float64), torch.ones(C, device="cuda", dtype=torch.float64)
    yb = torch.nn.functional.batch_norm(xb, rm_b, rv_b, gb, bb, True, 0.1, 1e-5)
    if relu:
        yb = torch.relu(yb)
    yb.backward(gy.double())
    step = 2.0 ** -8 if dtype is torch.bfloat16 else 2.0 ** -11
    n = x.numel() // C
    # element-wise: |got - want| <= one rounding step of |want| (+ a floor for values near zero)
    for got, want, what in ((ya, yb, "y"), (xa.grad, xb.grad, "d_x")):
        g_, w_ = got.detach().double(), want.detach()
        bad = (g_ - w_).abs() > step * w_.abs() + step * 1e-2 * float(w_.abs().max())
        # a ReLU pre-activation within rounding of zero may fall on the other side in d_x
        assert int(bad.sum()) <= (bad.numel() * 1e-4 if (relu and what == "d_x") else 0), (what, int(bad.sum()))
    rt = 2e-3 if relu else 1e-4
    assert_close(host(ga.grad), host(gb.grad), rtol=rt if n >= 100 else 5e-4, what="d_gamma")
    assert_close(host(ba.grad), host(bb.grad), rtol=rt if n >= 100 else 5e-4, what="d_beta")
    assert_close(host(rm_a), host(rm_b), rtol=1e-5, what="running_mean")
    assert_close(host(rv_a), host(rv_b), rtol=1e-5, what="running_var")


def test_autocast_step_keeps_batchnorm_in_half_precision():
    """under torch.autocast a converted model's BatchNorm layers run on the 2-byte kernels: no float32 copies of the activations"""
    import copy
    from movedepth_amd import networks

    torch.manual_seed(2)
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 16, 3, padding=1, bias=False), torch.nn.BatchNorm2d(16), torch.nn.ReLU(),
                              torch.nn.Conv2d(16, 8, 3, padding=1, bias=False), torch.nn.BatchNorm2d(8)).cuda().to(memory_format=torch.channels_last)
    conv = networks.convert_hip_sync_batchnorm(copy.deepcopy(ref))
    x = torch.randn(4, 3, 20, 24, device="cuda").contiguous(memory_format=torch.channels_last)
    seen = []
    conv[1].register_forward_hook(lambda m, i, o: seen.append((i[0].dtype, o.dtype)))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ya, yb = conv(x), ref(x)
    assert seen == [(torch.bfloat16, torch.bfloat16)]
    assert ya.dtype is yb.dtype
    assert_close(host(ya), host(yb), rtol=3e-2, what="autocast forward")     # two bf16 pipelines: a few rounding steps apart
    gw = torch.randn_like(ya, dtype=torch.float32)     # (a loss like mean(y^2) of a normalised output has no gradient to compare)
    (ya.float() * gw).sum().backward()
    (yb.float() * gw).sum().backward()
    assert_close(host(conv[0].weight.grad), host(ref[0].weight.grad), rtol=5e-2, what="autocast d_weight")


def test_eval_mode_and_state_dict_compat():
    from movedepth_amd import networks, ops

    torch.manual_seed(1)
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 16, 3, padding=1, bias=False), torch.nn.BatchNorm2d(16), torch.nn.ReLU(),
                              torch.nn.Conv2d(16, 6, 1), torch.nn.BatchNorm2d(6)).cuda()   # 6 channels: not a multiple of 4, left alone
    with torch.no_grad():
        ref[1].running_mean.normal_(); ref[1].running_var.uniform_(0.5, 2.0); ref[1].weight.uniform_(0.5, 1.5); ref[1].bias.normal_()
    import copy
    conv = networks.convert_hip_sync_batchnorm(copy.deepcopy(ref))
    assert isinstance(conv[1], networks.HipSyncBatchNorm) and isinstance(conv[4], torch.nn.BatchNorm2d)
    assert list(conv.state_dict().keys()) == list(ref.state_dict().keys())
    conv.load_state_dict(ref.state_dict())
    x = torch.randn(2, 3, 10, 12, device="cuda")
    ref.eval(); conv.eval()
    assert_close(host(conv(x)), host(ref(x)), rtol=1e-6, what="eval forward")
    # the one-kernel evaluation path
    y = ops.batch_norm_eval(x := torch.randn(2, 16, 5, 6, device="cuda"), ref[1].weight, ref[1].bias, ref[1].running_mean, ref[1].running_var, relu=True)
    assert_close(host(y), host(torch.relu(ref[1](x))), rtol=1e-6, what="batch_norm_eval")
    # training: batch statistics, running statistics updated like BatchNorm2d's
    ref.train(); conv.train()
    x = torch.randn(4, 3, 10, 12, device="cuda")
    assert_close(host(conv(x)), host(ref(x)), rtol=1e-5, what="train forward")
    assert_close(host(conv[1].running_mean), host(ref[1].running_mean), rtol=1e-5)
    assert_close(host(conv[1].running_var), host(ref[1].running_var), rtol=1e-5)
    assert int(conv[1].num_batches_tracked) == int(ref[1].num_batches_tracked)


def _worker(rank, world, port, xs, gys, relu, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from movedepth_amd import networks
    C = xs[0].shape[1]
    m = networks.HipSyncBatchNorm(C, relu=relu).cuda()
    m.sync_group = dist.group.WORLD
    with torch.no_grad():
        m.weight.copy_(torch.linspace(0.5, 1.5, C)); m.bias.copy_(torch.linspace(-0.3, 0.3, C))
    x = torch.from_numpy(xs[rank]).cuda().contiguous(memory_format=_fmt(xs[rank].shape)).requires_grad_(True)
    y = m(x)
    y.backward(torch.from_numpy(gys[rank]).cuda())
    q.put((rank, y.detach().cpu().numpy(), x.grad.cpu().numpy(), m.weight.grad.cpu().numpy(), m.bias.grad.cpu().numpy(),
           m.running_mean.cpu().numpy(), m.running_var.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("shape", [(3, 64, 6, 10), (2, 16, 4, 6, 8), (2, 128, 2, 2, 4), (2, 32, 8, 8, 16), (2, 8, 64, 128)])
def test_two_ranks_equal_the_big_batch(shape, relu):
    rng = np.random.default_rng(5)
    C = shape[1]
    xs = [rng.standard_normal(shape).astype(np.float32) + r for r in range(2)]   # different means per rank
    gys = [rng.standard_normal(shape).astype(np.float32) + 0.3 for _ in range(2)]      # a non-zero mean gradient: sum dz matters
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ps = [ctx.Process(target=_worker, args=(r, 2, port, xs, gys, relu, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in ps:
        p.join(60)
    x = torch.from_numpy(np.concatenate(xs)).cuda().double().requires_grad_(True)
    w = torch.linspace(0.5, 1.5, C).cuda().double().requires_grad_(True)
    b = torch.linspace(-0.3, 0.3, C).cuda().double().requires_grad_(True)
    rm, rv = torch.zeros(C, dtype=torch.float64).cuda(), torch.ones(C, dtype=torch.float64).cuda()
    y = torch.nn.functional.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5)
    if relu:
        y = torch.relu(y)
    y.backward(torch.from_numpy(np.concatenate(gys)).cuda().double())
    rel = lambda a, c: np.linalg.norm(a - c) / np.linalg.norm(c)
    assert rel(np.concatenate([g[1] for g in got]), host(y)) < 1e-5
    assert rel(np.concatenate([g[2] for g in got]), host(x.grad)) < 1e-4
    assert rel(got[0][3] + got[1][3], host(w.grad)) < 1e-4      # the ranks' local sums add up to the big-batch gradient
    assert rel(got[0][4] + got[1][4], host(b.grad)) < 1e-4
    for g in got:
        assert rel(g[5], host(rm)) < 1e-5 and rel(g[6], host(rv)) < 1e-5
    assert np.array_equal(got[0][5], got[1][5]) and np.array_equal(got[0][6], got[1][6])   # identical on both ranks
