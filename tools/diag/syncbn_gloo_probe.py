"""Does torch.nn.SyncBatchNorm run over gloo with CUDA tensors (two ranks sharing cuda:0)?"""
import os, sys, socket
import torch, torch.distributed as dist, torch.multiprocessing as mp


def worker(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8))).cuda()
    m.train()
    x = torch.randn(2, 3, 16, 16, device="cuda", generator=torch.Generator("cuda").manual_seed(rank))
    try:
        y = m(x)
        y.square().mean().backward()
        torch.cuda.synchronize()
        print(rank, "ok", float(y.mean()), float(m[1].running_var.mean()))
    except Exception as e:
        print(rank, "FAILED", type(e).__name__, str(e)[:300])
    dist.destroy_process_group()


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(port,), nprocs=2)
