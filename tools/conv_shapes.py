#!/usr/bin/env python3
"""Which 2-D convolutions the step spends its library time in: one training step of BASELINE config 2 under torch.profiler with
shapes; convolution forward / backward ops grouped by (input, weight) shape with their device time.  GPU only.
usage: python tools/conv_shapes.py [trainer args...]"""
import contextlib
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from movedepth_amd import miopen_setup
    miopen_setup.use_shipped_cache(0)
    from movedepth_amd.options import MovedepthOptions
    from movedepth_amd.synthetic import make_inputs
    from movedepth_amd.trainer import Trainer
    argv = ["--height", "192", "--width", "640", "--num_depth_bins", "96", "--batch_size", "6", "--res_arch", "18", "--prior_scale", "2",
            "--convex_up", "--weights_init", "scratch", "--learning_rate", "2e-4", "--local_rank", "0",
            "--miopen_find", "2" if miopen_setup.find_db_hits(0) else "1"] + sys.argv[1:]
    opt = MovedepthOptions().parse(argv)
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(sys.stderr):
        trainer = Trainer(opt)
    trainer.set_train()
    inputs = make_inputs(opt.batch_size, opt.height, opt.width, opt.frame_ids, seed=0, device=trainer.device)
    for _ in range(8):
        trainer.train_step(dict(inputs))
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(3):
            trainer.train_step(dict(inputs))
        torch.cuda.synchronize()
    agg = defaultdict(lambda: [0, 0.0])
    for e in prof.key_averages(group_by_input_shape=True):
        if "conv" in e.key.lower() and e.device_time_total > 0 and ("cudnn" in e.key or "miopen" in e.key or "convolution_backward" == e.key.split("::")[-1]
                                                                   or e.key in ("aten::convolution_backward", "aten::miopen_convolution", "aten::cudnn_convolution")):
            shp = [s for s in e.input_shapes if s]
            agg[(e.key, str(shp[:3]))][0] += e.count
            agg[(e.key, str(shp[:3]))][1] += e.self_device_time_total
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print("convolution ops by shape, 3 steps: total self device time %.2f ms per step" % (tot / 3e3))
    for (k, shp), (cnt, us) in rows[:40]:
        print("  %-34s %-72s x%-3d %8.1f us per step  (%.1f us each)" % (k, shp, cnt // 3, us / 3, us / max(cnt, 1)))


if __name__ == "__main__":
    main()
