"""GPU: BatchNorm (training) forward + backward per layer shape, MIOpen against torch's native kernels (the ones
torch.nn.SyncBatchNorm is built from), NCHW and channels_last.  Why: a data-parallel run with SyncBatchNorm leaves MIOpen's
BatchNorm for the native kernels on every rank."""
import torch, time
import torch.nn.functional as F

def ev(fn, n=20, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

shapes = [(6, 64, 96, 320), (6, 64, 48, 160), (6, 128, 24, 80), (6, 256, 12, 40), (6, 512, 6, 20), (12, 32, 48, 160), (6, 16, 96, 320)]
for shp in shapes:
    for cl in (False, True):
        x = torch.randn(*shp, device="cuda")
        if cl: x = x.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True)
        C = shp[1]
        w = torch.ones(C, device="cuda", requires_grad=True); b = torch.zeros(C, device="cuda", requires_grad=True)
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        gy = torch.randn_like(x)
        def fb():
            y = F.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5)
            y.backward(gy)
        t_mi = ev(fb)
        with torch.backends.cudnn.flags(enabled=False):
            t_na = ev(fb)
        # SyncBN's primitives
        def sync_like():
            mean, invstd = torch.batch_norm_stats(x, 1e-5)
            y = torch.batch_norm_elemt(x, w, b, mean, invstd, 1e-5)
            sum_dy, sum_dy_xmu, gw, gb = torch.batch_norm_backward_reduce(gy, x, mean, invstd, w, True, True, True)
            dx = torch.batch_norm_backward_elemt(gy, x, mean, invstd, w, sum_dy, sum_dy_xmu, torch.tensor([x.numel() // C], device="cuda", dtype=torch.int32))
        t_sy = ev(sync_like)
        print("%-22s %-13s MIOpen %7.1f us   native %7.1f us   SyncBN primitives %7.1f us" % (shp, "channels_last" if cl else "NCHW", t_mi, t_na, t_sy), flush=True)
