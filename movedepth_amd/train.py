"""`python -m movedepth_amd.train ...` -- same CLI as the reference's movedepth/train.py (whose shipped import of
`MovedepthOptions` fails; both names exist here).  Launch one process per GPU:

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m movedepth_amd.train \\
        --height 192 --width 640 --prior_scale 2 --ddp --batch_size 6 --convex_up --learning_rate 2e-4
"""
import os
import random

import numpy as np
import torch

from .options import MovedepthOptions
from .trainer import Trainer


def seed_all(seed):
    """reference train.py:8-19"""
    if not seed:
        seed = 1
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def main():
    import sys

    from . import miopen_setup

    argv = sys.argv[1:]
    rank = int(os.environ.get("RANK", "0"))
    miopen_setup.use_shipped_cache(rank)  # MIOpen's search results shipped in-tree (miopen_setup.py)
    # `--miopen_find 2` (every convolution on its searched solver) is left to the user: the shipped db covers the bench workload
    # (192x640, ResNet-18, batch 6 per GPU); for any other shape the first step would spend minutes searching
    opts = MovedepthOptions().parse(argv)
    seed_all(opts.pytorch_random_seed)
    Trainer(opts).train()


if __name__ == "__main__":
    main()
