"""Debug: FPN4 (3 calls per step, like the trainer) with HipSyncBatchNorm on two gloo ranks sharing cuda:0 against the single-process
big-batch run with plain BatchNorm."""
import os, sys, socket
import numpy as np
import torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


NET = os.environ.get("NET", "fpn4")


def build(seed=0):
    from movedepth_amd import networks
    torch.manual_seed(seed)
    if NET == "reg3d":
        return networks.reg3d(16, 16, down_size=3, fused_bn=os.environ.get("FUSED3D", "1") == "1")
    return networks.FPN4(8, scale=2)


def run(net, xs, gs, calls):
    outs = []
    for c in range(calls):
        o = net(xs[c])
        outs.append(o[0] if isinstance(o, tuple) else o)
    loss = sum((o * g).sum() for o, g in zip(outs, gs))
    loss.backward()
    return {n: p.grad.detach().cpu().numpy().copy() for n, p in net.named_parameters()}, [o.detach().cpu().numpy() for o in outs]


def worker(rank, world, port, xs, gs, calls, fuse, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from movedepth_amd import networks
    net = build()
    if os.environ.get("IMPL", "hip") == "torch":
        net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
    else:
        net = networks.convert_hip_sync_batchnorm(net, dist.group.WORLD, fuse_relu=fuse)
    net = net.cuda().to(memory_format=torch.channels_last_3d if NET == "reg3d" else torch.channels_last).train()
    if NET == "reg3d":
        net.find_convs = False
    x = [torch.from_numpy(a[rank]).cuda() for a in xs]
    g = [torch.from_numpy(a[rank]).cuda() for a in gs]
    grads, outs = run(net, x, g, calls)
    q.put((rank, grads, outs))
    dist.barrier()
    dist.destroy_process_group()


def main():
    calls = int(os.environ.get("CALLS", "3"))
    fuse = os.environ.get("FUSE", "1") == "1"
    rng = np.random.default_rng(0)
    if NET == "reg3d":
        xs = [[rng.standard_normal((2, 16, 16, 16, 32)).astype(np.float32) for _ in range(2)] for _ in range(calls)]
        gs = [[rng.standard_normal((2, 16, 16, 32)).astype(np.float32) for _ in range(2)] for _ in range(calls)]
    else:
        xs = [[rng.random((2, 3, 64, 128), dtype=np.float32) for _ in range(2)] for _ in range(calls)]
        gs = [[rng.standard_normal((2, 32, 16, 32)).astype(np.float32) for _ in range(2)] for _ in range(calls)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    ps = [ctx.Process(target=worker, args=(r, 2, port, xs, gs, calls, fuse, q)) for r in range(2)]
    [p.start() for p in ps]
    got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    [p.join(60) for p in ps]
    net = build().cuda().to(memory_format=torch.channels_last_3d if NET == "reg3d" else torch.channels_last).train()
    if NET == "reg3d":
        net.find_convs = False
    x = [torch.from_numpy(np.concatenate(a)).cuda() for a in xs]
    g = [torch.from_numpy(np.concatenate(a)).cuda() for a in gs]
    want, wouts = run(net, x, g, calls)
    rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
    for c in range(calls):
        print("call", c, "output rel", rel(np.concatenate([got[0][2][c], got[1][2][c]]), wouts[c]))
    for n in want:
        print("%-28s %.2e" % (n, rel(got[0][1][n] + got[1][1][n], want[n])))


if __name__ == "__main__":
    main()
