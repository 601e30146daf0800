"""Command-line options of the training hot path.  Flag names, types and defaults follow the reference's
MonodepthOptions (movedepth/options.py:7-350) for every flag the path reads ("live flags");
flags of subsystems that are out of scope (dataset paths, tensorboard, evaluation) are accepted and ignored so
existing launch lines keep working."""
import argparse
import os


class MonodepthOptions:
    def __init__(self):
        p = argparse.ArgumentParser(description="MOVEDepth (MI355X hot path) options")
        # paths / bookkeeping
        p.add_argument("--data_path", type=str, default="synthetic",
                       help="'synthetic' (default: seeded synthetic KITTI-shaped frames) or the root of a KITTI raw tree "
                            "(<date>/<drive>_sync/image_0{2,3}/data/*.jpg), read by movedepth_amd.datasets")
        p.add_argument("--train_files", type=str, default=None,
                       help="split file ('<folder> <frame> <l|r>' per line) when --data_path is a KITTI tree; default "
                            "<data_path>/splits/<split>/train_files.txt")
        p.add_argument("--log_dir", type=str, default=os.path.join(os.path.expanduser("~"), "tmp"))
        p.add_argument("--model_name", type=str, default="mdp")
        p.add_argument("--split", type=str, default="eigen_zhou")
        p.add_argument("--dataset", type=str, default="kitti")
        p.add_argument("--png", action="store_true")
        # model / hot path
        p.add_argument("--num_layers", type=int, default=18)
        p.add_argument("--res_arch", type=int, default=18, choices=[18, 34, 50])
        p.add_argument("--num_depth_bins", type=int, default=16)
        p.add_argument("--ztrans_start_epc", type=int, default=8)
        p.add_argument("--depth_bin_fac", type=float, default=0.3)
        p.add_argument("--ssim_lw", type=float, default=0.85)
        p.add_argument("--mask_lw", type=float, default=10)
        p.add_argument("--photo_conf", type=float, default=0.2)
        p.add_argument("--height", type=int, default=192)
        p.add_argument("--width", type=int, default=640)
        p.add_argument("--disparity_smoothness", type=float, default=1e-3)
        p.add_argument("--scales", nargs="+", type=int, default=[0, 1, 2, 3])
        p.add_argument("--min_depth", type=float, default=0.1)
        p.add_argument("--max_depth", type=float, default=100.0)
        p.add_argument("--frame_ids", nargs="+", type=int, default=[0, -1, 1])
        p.add_argument("--matching_ids", nargs="+", type=int, default=[0, -1])
        p.add_argument("--reg3d_c", type=int, default=16)
        p.add_argument("--prior_scale", type=int, default=2)
        p.add_argument("--norm_radius", type=int, default=1)
        p.add_argument("--schedule_type", type=str, default="inverse", choices=["inverse", "linear", "log"])
        p.add_argument("--z_scale", type=float, default=30)
        p.add_argument("--dist_thres", type=float, default=0)
        p.add_argument("--convex_up", action="store_true")
        p.add_argument("--load_pose", action="store_true")
        p.add_argument("--mask_mvs_conf", action="store_true")
        p.add_argument("--mask_mvs_dist", action="store_true")
        p.add_argument("--mask_mvs_geo", action="store_true")
        p.add_argument("--mask_mvs_auto", action="store_true")
        p.add_argument("--mvs_smooth_loss", action="store_true")
        p.add_argument("--dcn", action="store_true")
        p.add_argument("--disable_automasking", action="store_true")
        p.add_argument("--no_ssim", action="store_true")
        # optimisation
        p.add_argument("--batch_size", type=int, default=12)
        p.add_argument("--learning_rate", type=float, default=1e-4)
        p.add_argument("--lr_fac", type=float, default=1)
        p.add_argument("--num_epochs", type=int, default=20)
        p.add_argument("--scheduler_step_size", type=int, default=15)
        p.add_argument("--pytorch_random_seed", default=None, type=int)
        p.add_argument("--weights_init", type=str, default="pretrained", choices=["pretrained", "scratch"])
        # system
        p.add_argument("--no_cuda", action="store_true")
        p.add_argument("--num_workers", type=int, default=12)
        p.add_argument("--local_rank", default=int(os.environ.get("LOCAL_RANK", 0)), type=int)
        p.add_argument("--ddp", action="store_true")
        # loading / logging
        p.add_argument("--load_weights_folder", type=str)
        p.add_argument("--mono_weights_folder", type=str)
        p.add_argument("--models_to_load", nargs="+", type=str,
                       default=["mono_encoder", "mono_depth", "pose_encoder", "pose", "mask_cnn", "mvs_encoder", "reg3d",
                                "up"])
        p.add_argument("--log_frequency", type=int, default=250)
        p.add_argument("--save_frequency", type=int, default=1)
        p.add_argument("--save_intermediate_models", action="store_true")
        # MI355X build additions (not in the reference)
        p.add_argument("--steps_per_epoch", type=int, default=100, help="synthetic data: steps per epoch")
        p.add_argument("--automask_noise", type=str, default="device", choices=["device", "host"],
                       help="where the 1e-5 auto-mask tie-break noise is drawn: 'host' reproduces the reference's "
                            "torch.randn(CPU).to(device) draws, 'device' avoids the per-step host round trip")
        p.add_argument("--miopen_find", type=int, default=1, help="MIOpen solver search: 0 off, 1 the 3-D regulariser's convolutions, 2 every convolution")
        p.add_argument("--reg3d_channels_last", type=int, default=1,
                       help="run the 3-D regulariser in channels_last_3d and write the cost volume as (B,D,h,w,G)")
        p.add_argument("--hip_prob_conv", type=int, default=1,
                       help="the 3-D regulariser's last (C->1) convolution on the hand-written kernels (0: library)")
        p.add_argument("--hip_conv2", type=int, default=1,
                       help="reg3d.conv2 (32 -> 32 at half resolution) as 16 x 16 channel blocks on the hand-written bf16 x 3 kernels "
                            "(fp32 steps; 0: the library convolution)")
        p.add_argument("--hip_conv0", default="all", choices=["all", "wgrad", "none"],
                       help="the 3-D regulariser's first convolution on the MFMA kernels: all three directions, the "
                            "weight gradient only, or none (library)")
        p.add_argument("--vol_layout", default="auto", choices=["auto", "bgd", "ndhwc"],
                       help="storage of the grouped cost volume: planar (B,G,D,h,w) or channels-last (B,D,h,w,G); auto = "
                            "channels-last when the 3-D regulariser runs channels-last (measured fastest), else planar")
        p.add_argument("--nets2d_channels_last", type=int, default=1,
                       help="run the 2-D networks (encoders, decoders, FPN, pose) in channels_last: the library then picks "
                            "NHWC kernels without transposes around them; 52.5 vs 54.6 ms per step at config 2")
        p.add_argument("--nets2d_channels_last_skip", default="mono_depth",
                       help="comma-separated model names kept in NCHW.  The depth decoder pads by reflection before every "
                            "convolution, and the padded tensor comes back in NCHW, so each of its convolutions paid a layout "
                            "copy in channels_last: 50.06 vs 51.60 ms per step with it left in NCHW")
        p.add_argument("--bn_counter_on_host", type=int, default=1,
                       help="keep BatchNorm's num_batches_tracked counters in host memory (no GPU kernel per BatchNorm call)")
        p.add_argument("--hip_bn_relu", type=int, default=1,
                       help="the 3-D regulariser's two full-resolution BatchNorm+ReLU (+skip add) on the fused kernels "
                            "(47.8 vs 49.1 ms per step; 0: library BatchNorm + torch ops)")
        p.add_argument("--fused_photometric", type=int, default=1,
                       help="generate_images_pred + compute_losses' photometric part as one kernel launch each way per group of "
                            "losses (csrc/photo.hip); 0: one kernel per warp / loss / reduction, as in round 2")
        p.add_argument("--lazy_sample_grids", type=int, default=1,
                       help="with --fused_photometric: outputs[(\"sample\", f, s)] -- which nothing on the training path reads "
                            "(reference trainer.py:524-529 only hands it to grid_sample) -- is computed when first accessed "
                            "instead of being stored by every step (8 maps, 47 MB per step at config 2); the same bits either way")
        p.add_argument("--amp", default="none", choices=["none", "bf16", "fp16"],
                       help="mixed precision (BASELINE configs 4 / 5): networks under autocast, 2-byte cost volume; the "
                            "headline bench is fp32")
        p.add_argument("--fused_adam", type=int, default=1, help="torch's fused Adam kernel on the GPU (0: default implementation)")
        p.add_argument("--sync_bn", type=int, default=1, help="with --ddp: convert BatchNorm to SyncBatchNorm (reference)")
        p.add_argument("--sync_bn_impl", default="hip", choices=["hip", "torch"],
                       help="synchronised BatchNorm on the hand-written kernels (networks.HipSyncBatchNorm: one all-reduce of 2C "
                            "sums per layer and direction) or torch.nn.SyncBatchNorm (torch's native batch-norm kernels)")
        p.add_argument("--force_sync_bn", type=int, default=0,
                       help="measurement: run the synchronised-BatchNorm path of --sync_bn_impl without --ddp (a group of one)")
        p.add_argument("--grad_bucket_mb", type=float, default=32.0, help="with --ddp: all-reduce bucket size")
        p.add_argument("--direct_rccl", type=int, default=int(os.environ.get("MD_DIRECT_RCCL", "0")),
                       help="with --ddp on the nccl backend: every collective of the step (BatchNorm statistics and gradient buckets) as "
                            "ncclAllReduce on the compute stream through one communicator (rccl_direct.py) instead of torch's group and its "
                            "stream; default: the MD_DIRECT_RCCL environment variable of the LAUNCHER, else 0")
        self.parser = p

    def parse(self, args=None):
        # flags of out-of-scope reference subsystems are tolerated
        self.options, _unknown = self.parser.parse_known_args(args)
        return self.options


MovedepthOptions = MonodepthOptions  # the name the reference's train.py imports (train.py:5)
