"""MIOpen solver choices for the library convolutions.

MIOpen picks a convolution's solver by heuristic or from the result of a measured search kept in its user find-db.  The
search matters here: the heuristic choice for the 3-D regulariser's fp32 convolutions is a naive kernel (75x slower), and
for the 2-D networks the searched choices are worth 4 ms of a 47 ms step -- but a full search of the bench workload takes
about 9 minutes per process.  `movedepth_amd/miopen_cache/` therefore ships the search results (find-db) and the kernels the
search compiled, produced on the GPU box by tools/make_miopen_cache.sh.

    use_shipped_cache(rank)      point this process at a PRIVATE copy of that cache (MIOpen's files are not meant to be shared by
                                 concurrent writers); call before the first convolution, no-op if the user set the paths
    find_db_hits(device_index)   True if this MIOpen honours the shipped find-db (same build, same device): two find calls for
                                 convolutions of the bench workload answer in tens of milliseconds on a hit and run the search
                                 for that problem (about half a second) on a miss
A launcher that gets True passes `--miopen_find 2` (every convolution in search mode); otherwise the trainer's default
`--miopen_find 1` searches the regulariser's convolutions only, as it always did.
"""
import atexit
import os
import shutil
import sys
import tempfile
import time

import torch

CACHE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_cache")


def use_shipped_cache(rank=0):
    if not os.path.isdir(CACHE_DIR) or "MIOPEN_USER_DB_PATH" in os.environ:
        return False
    try:
        priv = tempfile.mkdtemp(prefix="movedepth_miopen_rank%d_" % rank)   # private, unpredictable name, mode 0700
        shutil.copytree(CACHE_DIR, priv, dirs_exist_ok=True)
    except OSError:
        return False
    os.environ["MIOPEN_USER_DB_PATH"] = os.path.join(priv, "db")
    os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = os.path.join(priv, "cache")
    atexit.register(shutil.rmtree, priv, ignore_errors=True)  # 2 MB per process: do not pile up in the temp directory
    return True


def find_db_hits(device_index=0, limit_s=0.3, log=sys.stderr):
    """Measured on MI355X: 0.05-0.11 s per call on a hit (the latter with two ranks sharing a GPU), 0.49-0.54 s on a miss (an
    empty user db).  The two problems are ResNet-18 layer1 / layer2 at 192x640, batch 6 (both in the shipped db)."""
    dev = torch.device("cuda", device_index)
    prev = torch.backends.cudnn.benchmark
    worst = 0.0
    try:
        torch.backends.cudnn.benchmark = False
        torch.nn.functional.conv2d(torch.randn(1, 64, 8, 8, device=dev), torch.randn(64, 64, 3, 3, device=dev), padding=1)
        torch.cuda.synchronize(dev)  # library start-up, not timed
        torch.backends.cudnn.benchmark = True
        for c, h, w in ((64, 48, 160), (128, 24, 80)):
            x = torch.randn(6, c, h, w, device=dev).contiguous(memory_format=torch.channels_last)
            wt = torch.randn(c, c, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
            torch.cuda.synchronize(dev)
            t0 = time.time()
            torch.nn.functional.conv2d(x, wt, padding=1)
            torch.cuda.synchronize(dev)
            worst = max(worst, time.time() - t0)
    except Exception as e:  # noqa: BLE001 -- any failure means "do not rely on the db"
        if log:
            log.write("miopen_setup: find-db probe failed (%s): solver search for the 3-D convolutions only\n" % e)
        return False
    finally:
        torch.backends.cudnn.benchmark = prev
    hit = worst < limit_s
    if log:
        log.write("miopen_setup: find-db probe %.2f s -> %s\n" % (worst, "shipped solver choices for all convolutions" if hit else
                                                                  "miss: solver search for the 3-D convolutions only"))
    return hit
