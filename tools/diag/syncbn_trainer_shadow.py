"""Debug: two ranks of the Trainer (--ddp, HipSyncBatchNorm) on cuda:0 over gloo; every HipSyncBatchNorm call is shadowed by the same
layer written with torch ops in float64 + all_reduce, forward and backward; prints the calls whose output / input gradient differ."""
import os, sys, socket
import numpy as np
import torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ARGV = ["--height", "64", "--width", "128", "--num_depth_bins", "16", "--convex_up", "--weights_init", "scratch",
        "--miopen_find", "0", "--automask_noise", "host", "--grad_bucket_mb", "8", "--learning_rate", "1e-3", "--disable_automasking"]


def worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MD_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from movedepth_amd import networks
    from movedepth_amd.options import MovedepthOptions
    from movedepth_amd.synthetic import make_inputs
    from movedepth_amd.trainer import Trainer
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
    opt = MovedepthOptions().parse(ARGV + ["--batch_size", "2", "--ddp"])
    torch.manual_seed(50 + rank); np.random.seed(50 + rank)
    t = Trainer(opt)
    t.set_train()
    report = []
    pgrads = {}
    names = {m: mn + "." + n for mn, net in t.models.items() for n, m in net.named_modules() if isinstance(m, networks.HipSyncBatchNorm)}
    rel = lambda a, b: float((a.double() - b).norm() / (b.norm() + 1e-300))

    def fwd_hook(mod, inp, out):
        x = inp[0].detach()
        dims = [0] + list(range(2, x.dim()))
        xd = x.double()
        s = torch.stack([xd.sum(dims), (xd * xd).sum(dims)])
        dist.all_reduce(s)
        n = x.numel() // x.shape[1] * world
        mean, var = s[0] / n, s[1] / n - (s[0] / n) ** 2
        shp = [1, -1] + [1] * (x.dim() - 2)
        invstd = 1.0 / torch.sqrt(var + mod.eps)
        xh = (xd - mean.view(shp)) * invstd.view(shp)
        z = xh * mod.weight.detach().double().view(shp) + mod.bias.detach().double().view(shp)
        y = torch.relu(z) if mod.relu else z
        e = rel(out.detach(), y)
        call = len(report)
        rec = {"name": names[mod], "call": call, "fwd": e, "shape": tuple(x.shape), "stride": tuple(x.stride())}
        report.append(rec)
        if out.requires_grad and inp[0].requires_grad:
            store = {}

            def on_dy(g):
                store["dy"] = g.detach().double()
                return g

            def on_dx(g):
                dy = store["dy"]
                dz = dy * (z > 0) if mod.relu else dy
                r = torch.stack([dz.sum(dims), (dz * xh).sum(dims)])
                acc = pgrads.setdefault(names[mod], torch.zeros_like(r))
                acc += r            # this rank's d_beta / d_gamma of this call
                dist.all_reduce(r)
                gi = (mod.weight.detach().double() * invstd).view(shp)
                dx = gi * (dz - (r[0] / n).view(shp) - xh * (r[1] / n).view(shp))
                rec["bwd"] = rel(g.detach(), dx)
                rec["dy_stride"] = tuple(dy.stride())
                return g

            out.register_hook(on_dy)
            inp[0].register_hook(on_dx)

    for m in names:
        m.register_forward_hook(fwd_hook)
    shard = make_inputs(2, 64, 128, opt.frame_ids, seed=200 + rank, device=t.device)
    torch.manual_seed(300); np.random.seed(300)
    t.train_step(dict(shard))
    torch.cuda.synchronize()
    # BatchNorm parameter gradients after the reducer: mean over the ranks of the per-rank sums
    perr = []
    for mod, nm in names.items():
        if nm in pgrads:
            ref = pgrads[nm].clone()
            dist.all_reduce(ref)
            ref /= world
            perr.append((max(rel(mod.bias.grad, ref[0]), rel(mod.weight.grad, ref[1])), nm, rel(mod.bias.grad, ref[0]), rel(mod.weight.grad, ref[1])))
    perr.sort(reverse=True)
    print("rank", rank, "worst BatchNorm parameter gradients vs the shadow sums:", [(n, "%.1e" % b, "%.1e" % w) for _, n, b, w in perr[:6]], flush=True)
    q.put((rank, report))
    dist.barrier()
    dist.destroy_process_group()


def main():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    ps = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    got = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    [p.join(60) for p in ps]
    for rank, rep in got:
        bad = [r for r in rep if r["fwd"] > 1e-5 or r.get("bwd", 0) > 1e-4]
        print("rank", rank, "calls", len(rep), "with backward", sum("bwd" in r for r in rep), "bad", len(bad))
        for r in bad[:40]:
            print("  ", r)
        worst = sorted(rep, key=lambda r: -r.get("bwd", 0))[:5]
        print("  worst bwd:", [(r["name"], r["call"], "%.1e" % r.get("bwd", 0)) for r in worst])


if __name__ == "__main__":
    main()
