"""Trainer with the reference's step API (movedepth/trainer.py:33-911): `Trainer(options)`, `train()`,
`process_batch(inputs, is_train=False) -> (outputs, losses)`, `generate_images_pred`, `compute_losses`,
`compute_reprojection_loss`, `compute_fuse_losses`, `compute_loss_masks` -- same argument meaning, same output /
loss dictionary keys, same quirks -- with the geometry / cost-volume / photometric work done by
the HIP kernels behind movedepth_amd.ops and the dense convolutions left to MIOpen.

Deliberate differences from the reference's host code (values unchanged):
  * no per-sample Python loop, no (B,D,C,h,w) tensor: one kernel writes the grouped volume in reg3d's layout;
  * the identity reprojection loss is evaluated once per step, not once per scale (App. B-4: identical values);
  * the masked-consistency loss is a masked mean instead of boolean indexing: no device->host sync in the step;
  * one gradient reducer over all parameters instead of 8 DDP wrappers (movedepth_amd/dp.py).
"""
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F
import torch.optim as optim

from . import networks, ops
from .dp import GradSync, broadcast_parameters
from .layers import (SSIM, BackprojectDepth, Project3D, disp_to_depth, random_image_mask,
                     transformation_from_parameters)
from .synthetic import SyntheticLoader


def build_models(opt, num_pose_frames=2):
    """The sub-models of the reference's Trainer.__init__ (trainer.py:67-131), on the CPU, in its construction order.
    Returns (models, names trained at `learning_rate`, names trained at `learning_rate * lr_fac`)."""
    models = {}
    pretrained = opt.weights_init == "pretrained"
    models["mono_encoder"] = networks.ResnetEncoder(opt.res_arch, pretrained)
    models["mono_depth"] = networks.DepthDecoder(models["mono_encoder"].num_ch_enc, opt.scales)
    main = ["mono_encoder", "mono_depth"]
    if not opt.load_pose:
        models["pose_encoder"] = networks.ResnetEncoder(opt.res_arch, pretrained, num_input_images=num_pose_frames)
        models["pose"] = networks.PoseDecoder(models["pose_encoder"].num_ch_enc, num_input_features=1,
                                              num_frames_to_predict_for=2)
        main += ["pose_encoder", "pose"]
    models["mask_cnn"] = networks.UncertNet()
    models["mvs_encoder"] = networks.FPN4(base_channels=8, scale=opt.prior_scale, dcn=opt.dcn)
    if opt.num_depth_bins >= 8:
        models["reg3d"] = networks.reg3d(opt.reg3d_c, opt.reg3d_c, down_size=3, fused_bn=bool(getattr(opt, "hip_bn_relu", 0)))
    else:
        models["reg3d"] = networks.reg2d(opt.reg3d_c, opt.reg3d_c)
    mvs = ["mask_cnn", "mvs_encoder", "reg3d"]
    if opt.convex_up:
        models["up"] = networks.convex_upsample_layer(feature_dim=8 * 2 ** opt.prior_scale, scale=opt.prior_scale)
        main.append("up")
    return models, main, mvs


class StepOutputs(dict):
    """The `outputs` dictionary of process_batch.  An entry registered with lazy() is produced when first accessed: `key in
    outputs` and `outputs[key]` behave as if it were stored; keys() / items() list it only once it has been."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._lazy = {}

    def lazy(self, key, fn):
        self._lazy[key] = fn

    def __missing__(self, key):
        fn = self._lazy.pop(key, None)
        if fn is None:
            raise KeyError(key)
        value = self[key] = fn()
        return value

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    def get(self, key, default=None):
        return self[key] if key in self else default


class Trainer:
    def __init__(self, options):
        self.opt = options
        self.log_path = os.path.join(self.opt.log_dir, self.opt.model_name)
        assert self.opt.height % 32 == 0, "'height' must be a multiple of 32"
        assert self.opt.width % 32 == 0, "'width' must be a multiple of 32"
        assert self.opt.frame_ids[0] == 0, "frame_ids must start with 0"
        assert len(self.opt.frame_ids) > 1, "frame_ids must have more than 1 frame specified"
        if self.opt.no_cuda or not torch.cuda.is_available():
            raise RuntimeError("movedepth_amd runs its hot path on HIP kernels: a GPU is required (no CPU fallback)")

        self.local_rank = self.opt.local_rank
        # MD_SHARE_GPU=1 (testing on a 1-GPU box only): every rank uses device 0 and the gloo backend
        share_gpu = os.environ.get("MD_SHARE_GPU", "0") == "1"
        dev_index = 0 if share_gpu else self.local_rank
        torch.cuda.set_device(dev_index)
        self.device = torch.device("cuda", dev_index)
        self.rank, self.world_size = 0, 1
        if self.opt.ddp:
            if not dist.is_initialized():
                # "nccl" = RCCL over xGMI.  An explicit time limit (MD_DIST_TIMEOUT_S, default 600 s): with torch's asynchronous
                # error handling (on by default) a collective that cannot complete -- a rank died, a mismatch -- ends the
                # process with an error instead of blocking for ever
                import datetime
                dist.init_process_group(backend="gloo" if share_gpu else "nccl",
                                        timeout=datetime.timedelta(seconds=float(os.environ.get("MD_DIST_TIMEOUT_S", "600"))))
            self.rank, self.world_size = dist.get_rank(), dist.get_world_size()

        self.num_scales = len(self.opt.scales)
        self.num_pose_frames = 2
        self.matching_ids = self.opt.matching_ids
        opt = self.opt

        # ---- models (names = checkpoint file names of the reference, trainer.py:67-131)
        self.parameters_to_train, self.mvs_parameters_to_train = [], []
        self.models, main, mvs = build_models(opt, self.num_pose_frames)
        # library convolutions: the 3-D regulariser runs channels-last (its input volume is written in that layout by
        # the HIP kernel) with MIOpen's solver search enabled for its convs only (see networks.reg3d)
        self.models["reg3d"].find_convs = bool(opt.miopen_find)
        if opt.miopen_find >= 2:  # every convolution of the model (minutes of solver search at the first step of a process)
            torch.backends.cudnn.benchmark = True
        self.models["reg3d"].hip_prob = bool(opt.hip_prob_conv)
        self.models["reg3d"].hip_conv0_wgrad = opt.hip_conv0 != "none"
        if hasattr(self.models["reg3d"], "conv2") and isinstance(self.models["reg3d"].conv2, networks.ConvBnReLU3D):
            # 32 -> 32 only: at 64 and 128 channels the 16 / 64 block launches lose to the library (profiles/r05_conv3d_channel_blocks.txt)
            self.models["reg3d"].conv2.hip_cb = bool(getattr(opt, "hip_conv2", 1)) and self.models["reg3d"].conv2.conv.in_channels == 32
        self.models["reg3d"].lib_conv0_fwd_dgrad = opt.hip_conv0 == "wgrad"
        self.vol_layout = "bgd"
        if opt.reg3d_channels_last and opt.num_depth_bins >= 8:
            self.models["reg3d"] = self.models["reg3d"].to(memory_format=torch.channels_last_3d)
            # channels-last volume (B,D,h,w,G).  The hand-written first layer can also read the planar volume in place
            # (--vol_layout bgd), but that measured 0.6 ms per step slower: its planar-input weight gradient and
            # forward cost +245 / +104 us, the plane-sweep backward gains 52 us, and the planar plane-sweep forward is
            # no faster inside the step (76-78 us) than the channels-last one (75 us) -- DESIGN.md 4.3
            self.vol_layout = "ndhwc"
        if opt.vol_layout != "auto":
            self.vol_layout = opt.vol_layout
        # the statistics all-reduces of the synchronised BatchNorm layers: straight to RCCL on the compute stream through a
        # communicator of their own when the backend is nccl (rccl_direct: no cross-stream hops, 2 ms per step), else torch's group
        # (opt-in, MD_DIRECT_RCCL=1; then the gradient buckets go the same way -- one communicator, one stream: rccl_direct's header)
        bn_group = None
        self.direct_all_reduce = None
        if opt.ddp and opt.sync_bn:
            from . import rccl_direct
            rccl_direct.ENABLED = bool(getattr(opt, "direct_rccl", 0))
            self.direct_all_reduce = rccl_direct.make(None) if opt.sync_bn_impl == "hip" else None
            bn_group = self.direct_all_reduce or dist.group.WORLD
        self.bn_group = bn_group
        for k, m in self.models.items():
            if opt.sync_bn and (opt.ddp or opt.force_sync_bn):   # also with MD_SHARE_GPU=1: over gloo on CUDA tensors
                if opt.sync_bn_impl == "hip":
                    # every BatchNorm on the kernels of csrc/syncbn.hip: statistics over the global batch, one all-reduce of 2C
                    # sums per layer and direction (torch's SyncBatchNorm is built from its native batch-norm kernels, which
                    # cost a rank 14 % of a step before any collective: DESIGN 6)
                    m = networks.convert_hip_sync_batchnorm(m, bn_group if opt.ddp else None,
                                                            fuse_relu=os.environ.get("MD_HIPBN_FUSE_RELU", "1") == "1")
                elif opt.ddp:
                    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m)
            self.models[k] = m.to(self.device)
            if opt.nets2d_channels_last and k != "reg3d" and k not in opt.nets2d_channels_last_skip.split(","):
                # 2-D networks in channels_last: same results (outputs equal to 1e-7, tests/test_trainer_parity.py), the
                # library's NHWC kernels without the NCHW<->NHWC transposes around them; -2.1 ms per step at config 2
                self.models[k] = self.models[k].to(memory_format=torch.channels_last)
        for m in self.models.values():
            for mod in m.modules():
                if isinstance(mod, networks.FusedBNReLU3d):
                    if opt.ddp and opt.sync_bn:   # what convert_sync_batchnorm does for the others
                        mod.sync_group = bn_group
        if opt.bn_counter_on_host:
            # BatchNorm's num_batches_tracked += 1 is a GPU kernel per BatchNorm call (115 per step, ~0.5 ms) for a counter
            # nothing on the device reads (momentum is fixed): keep the counters in host memory.  Same state_dict.
            for m in self.models.values():
                for mod in m.modules():
                    if isinstance(mod, (torch.nn.modules.batchnorm._BatchNorm, networks.FusedBNReLU3d, networks.HipSyncBatchNorm)) and \
                            mod.num_batches_tracked is not None:
                        mod.num_batches_tracked = mod.num_batches_tracked.cpu()
        for k in main:
            self.parameters_to_train += list(self.models[k].parameters())
        for k in mvs:
            self.mvs_parameters_to_train += list(self.models[k].parameters())

        # geometry modules kept for API compatibility (the kernels fuse them)
        fh, fw = opt.height // 2 ** opt.prior_scale, opt.width // 2 ** opt.prior_scale
        self.backprojector = BackprojectDepth(opt.num_depth_bins, fh, fw).to(self.device)
        self.projector = Project3D(opt.num_depth_bins, fh, fw).to(self.device)
        self.backproject_depth, self.project_3d = {}, {}
        for scale in opt.scales:
            h, w = opt.height // 2 ** scale, opt.width // 2 ** scale
            self.backproject_depth[scale] = BackprojectDepth(opt.batch_size, h, w).to(self.device)
            self.project_3d[scale] = Project3D(opt.batch_size, h, w).to(self.device)
        if not opt.no_ssim:
            self.ssim = SSIM().to(self.device)

        # same Adam as the reference (trainer.py:137-141); on the GPU as torch's single-kernel ("fused") variant
        # instead of ~10 multi-tensor passes over the 28 M parameters per step
        adam_kw = {"fused": True} if (opt.fused_adam and self.device.type == "cuda") else {}
        self.model_optimizer = optim.Adam([
            {"params": self.parameters_to_train, "lr": opt.learning_rate},
            {"params": self.mvs_parameters_to_train, "lr": opt.learning_rate * opt.lr_fac}], **adam_kw)
        self.model_lr_scheduler = optim.lr_scheduler.StepLR(self.model_optimizer, opt.scheduler_step_size, 0.1)

        if opt.load_weights_folder is not None:
            self.load_model()
        if opt.mono_weights_folder is not None:
            self.load_mono_model()

        self.grad_sync = None
        if opt.ddp:
            broadcast_parameters(self.models.values())
            self.grad_sync = GradSync(self.parameters_to_train + self.mvs_parameters_to_train, opt.grad_bucket_mb,
                                      direct=self.direct_all_reduce)
            opt.log_frequency = max(1, opt.log_frequency // self.world_size)

        self.train_sampler = None
        if opt.data_path == "synthetic":
            self.train_loader = SyntheticLoader(opt.batch_size, opt.height, opt.width, opt.frame_ids, opt.steps_per_epoch,
                                                self.rank, self.world_size, device="cpu")
        else:
            # KITTI raw layout, the reference's sharding (trainer.py:166-179): one rank-strided shard per process, drop_last
            from . import datasets
            split = opt.train_files or os.path.join(opt.data_path, "splits", opt.split, "train_files.txt")
            train = datasets.KITTIRAWDataset(opt.data_path, datasets.read_split(split), opt.height, opt.width, opt.frame_ids,
                                             4, is_train=True, img_ext=".png" if opt.png else ".jpg", seed=self.rank)
            self.train_loader, self.train_sampler = datasets.make_loader(train, opt.batch_size, self.rank, self.world_size,
                                                                         shuffle=True, num_workers=opt.num_workers)
        self.num_total_steps = len(self.train_loader) * opt.num_epochs
        self.depth_metric_names = ["de/abs_rel", "de/sq_rel", "de/rms", "de/log_rms", "da/a1", "da/a2", "da/a3"]
        self.epoch, self.step = 0, 0
        if self.rank == 0:
            print("Training model named:\n  ", opt.model_name, "\nTraining is using:\n  ", self.device)

    # ------------------------------------------------------------------ mode / loop
    def set_train(self):
        for m in self.models.values():
            m.train()

    def set_eval(self):
        for m in self.models.values():
            m.eval()

    def train(self):
        self.epoch, self.step = 0, 0
        self.start_time = time.time()
        for self.epoch in range(self.opt.num_epochs):
            if self.train_sampler is not None:
                self.train_sampler.set_epoch(self.epoch)
            elif hasattr(self.train_loader, "set_epoch"):
                self.train_loader.set_epoch(self.epoch)
            self.run_epoch()
            if (self.epoch + 1) % self.opt.save_frequency == 0 and self.epoch > 15:
                self.save_model()

    def train_step(self, inputs):
        """process_batch -> backward -> (gradient all-reduce) -> optimizer step (reference trainer.py:269-272)."""
        amp = getattr(self.opt, "amp", "none")
        if amp == "none":
            outputs, losses = self.process_batch(inputs, is_train=True)
        else:
            # BASELINE configs 4 / 5: the networks under autocast (library convolutions on the bf16 / fp16 MFMA paths), the
            # plane-sweep kernels on their 2-byte builds (they dispatch on the feature dtype); every other hand-written
            # kernel widens its inputs to fp32, as autocast does for the losses in the reference's ecosystem
            with torch.autocast("cuda", dtype=torch.bfloat16 if amp == "bf16" else torch.float16):
                outputs, losses = self.process_batch(inputs, is_train=True)
        if self.grad_sync is not None:
            self.grad_sync.zero_grad()
        else:
            self.model_optimizer.zero_grad(set_to_none=True)
        if amp == "fp16":
            if getattr(self, "_scaler", None) is None:
                self._scaler = torch.amp.GradScaler("cuda")
            self._scaler.scale(losses["loss"]).backward()
            if self.grad_sync is not None:
                self.grad_sync.finish()
            self._scaler.step(self.model_optimizer)
            self._scaler.update()
        else:
            losses["loss"].backward()
            if self.grad_sync is not None:
                self.grad_sync.finish()
            self.model_optimizer.step()
        return outputs, losses

    def run_epoch(self):
        if self.rank == 0:
            print("Training")
        self.set_train()
        for batch_idx, inputs in enumerate(self.train_loader):
            before_op_time = time.time()
            outputs, losses = self.train_step(inputs)
            early_phase = batch_idx % self.opt.log_frequency == 0 and self.step < 2000
            late_phase = self.step % 2000 == 0
            if (early_phase or late_phase) and self.rank == 0:
                torch.cuda.synchronize()
                self.log_time(batch_idx, time.time() - before_op_time, float(losses["loss"]))
            self.step += 1
        self.model_lr_scheduler.step()

    def log_time(self, batch_idx, duration, loss):
        samples_per_sec = self.opt.batch_size * self.world_size / duration
        print("epoch {:>3} | batch {:>6} | examples/s: {:5.1f} | loss: {:.5f}".format(self.epoch, batch_idx, samples_per_sec,
                                                                                     loss))

    # ------------------------------------------------------------------ the step
    def process_batch(self, inputs, is_train=False):
        """Pass a minibatch through the networks and generate images and losses (reference trainer.py:297-442)."""
        opt = self.opt
        for key, ipt in inputs.items():
            if torch.is_tensor(ipt):     # (private entries of an earlier call on the same dictionary are not tensors)
                inputs[key] = ipt.to(self.device, non_blocking=True)
        outputs = StepOutputs()
        if not opt.load_pose:
            outputs.update(self.predict_poses(inputs, None))
        else:
            for f_i in opt.frame_ids[1:]:
                outputs[("cam_T_cam", 0, f_i)] = inputs["relative_pose", f_i]
        relative_poses = torch.stack([inputs[("relative_pose", idx)] for idx in self.matching_ids[1:]], 1)  # B N 4 4

        # mvs feature extraction
        ref_match_feat, ref_context_feat = self.models["mvs_encoder"](inputs["color_aug", 0, 0])
        src_match_feats = [self.models["mvs_encoder"](inputs["color_aug", f_i, 0])[0] for f_i in self.matching_ids[1:]]

        # single frame path + mono reprojection loss
        feats = self.models["mono_encoder"](inputs["color_aug", 0, 0])
        outputs.update(self.models["mono_depth"](feats, no_match=False))
        self.generate_images_pred(inputs, outputs)
        mono_losses = self.compute_losses(inputs, outputs)

        # mono depth prior -> hypotheses around it (velocity-guided after ztrans_start_epc)
        disp_prior = outputs[("disp", opt.prior_scale)].detach()
        depth_prior = 1 / (1 / opt.max_depth + disp_prior * (1 / opt.min_depth - 1 / opt.max_depth))
        z_trans = None
        if self.epoch > opt.ztrans_start_epc:
            z_trans = opt.z_scale * relative_poses[:, :, 2:3, -1:]  # B N 1 1
            if z_trans.shape[1] != 1:
                # the reference's broadcast fails for N > 1; defined here as: first lookup frame
                z_trans = z_trans[:, :1]
            z_trans = z_trans.reshape(-1).contiguous()
        sched = dict(prior=depth_prior, ndepth=opt.num_depth_bins, scale_fac=opt.depth_bin_fac, z_trans=z_trans,
                     type=opt.schedule_type)
        # Only the first / last hypothesis planes are needed outside the kernel (localmax endpoints).  For the inverse and linear
        # spacings the interval position of bin k is k / (D - 1): exactly 0 for the first and 1 for the last bin whatever D is, so
        # a two-bin schedule IS those two planes, bit for bit -- 0.2 MB instead of the (B,D,h,w) tensor (17.7 MB, 14 us per step).
        # The 'log' spacing evaluates exp(log .1 + (log 10 * k) / (D - 1)) left to right (layers.py:277): (a * 95) / 95 need not
        # round to a, so there the end planes are taken from the full schedule.
        if opt.schedule_type == "log":
            full = ops.schedule_depth_range(depth_prior, opt.num_depth_bins, opt.depth_bin_fac, z_trans, opt.schedule_type)
            end_planes = torch.stack([full[:, 0], full[:, -1]], 1)
        else:
            end_planes = ops.schedule_depth_range(depth_prior, 2, opt.depth_bin_fac, z_trans, opt.schedule_type)
        min_inv, max_inv = 1 / end_planes[:, 1], 1 / end_planes[:, 0]  # swapped on purpose (App. B-6)

        def mvs_branch(ref_feat, want_prob=False):
            vols = [ops.costvol_grouped(ref_feat, src_match_feats[f_idx], inputs[("K", 2)], inputs[("inv_K", 2)],
                                        relative_poses[:, f_idx], opt.reg3d_c, layout=self.vol_layout, **sched)
                    for f_idx in range(len(self.matching_ids) - 1)]
            cor_feats, _ = ops.fuse_volumes(vols, layout=self.vol_layout)
            logits = self.models["reg3d"](cor_feats)  # B D h w
            return ops.softmax_entropy_localmax(logits, min_inv, max_inv, opt.norm_radius, want_prob=want_prob)

        depth_mvs, cost_prob_entropy, _ = mvs_branch(ref_match_feat)
        trust_mono_mask = self.models["mask_cnn"](cost_prob_entropy)  # B 1 h w

        # mask-augmented depth prediction
        ori_H, ori_W = inputs["color_aug", 0, 0].shape[2:]
        masked_img, this_aug_mask = random_image_mask(inputs["color_aug", 0, 0], [ori_H // 3, ori_W // 3])
        ref_aug_feat, _ = self.models["mvs_encoder"](masked_img)
        # the probability volume behind --mask_mvs_conf is the MASK-AUGMENTED pass's: upstream overwrites `cost_prob` at
        # trainer.py:394-395 before it is upsampled and thresholded at :414-416
        depth_mvs_aug, _, cost_prob = mvs_branch(ref_aug_feat, want_prob=opt.mask_mvs_conf)

        this_mask = F.interpolate(this_aug_mask, [depth_mvs_aug.shape[1], depth_mvs_aug.shape[2]], mode="bilinear",
                                  align_corners=True).sum(1).to(torch.bool).float()
        # == smooth_l1_loss(aug[mask], mvs[mask]) without the boolean-index host sync
        sl1 = F.smooth_l1_loss(depth_mvs_aug, depth_mvs, reduction="none")
        this_masked_loss = (sl1 * this_mask).sum() / this_mask.sum() * opt.mask_lw
        mono_losses["masked_loss"] = this_masked_loss * opt.mask_lw  # mask_lw twice, as upstream (App. B-3)
        mono_losses["loss"] = mono_losses["loss"] + mono_losses["masked_loss"]
        outputs["masked_depth"] = depth_mvs_aug
        outputs["masked_aug"] = this_aug_mask

        # upsample mvs depth
        if not opt.convex_up:
            depth_mvs = F.interpolate(depth_mvs.unsqueeze(1), [opt.height, opt.width], mode="bilinear", align_corners=True)[:, 0]
        else:
            depth_mvs = self.models["up"](depth_mvs, ref_context_feat)
        outputs["depth_mvs"] = depth_mvs
        _, mono_depth = disp_to_depth(outputs[("disp", 0)], opt.min_depth, opt.max_depth)
        trust_mono_mask = F.interpolate(trust_mono_mask, [opt.height, opt.width], mode="bilinear", align_corners=True)
        fused_depth = (1 - trust_mono_mask) * depth_mvs[:, None].detach() + trust_mono_mask * mono_depth.detach()
        outputs["fused_depth"] = fused_depth
        outputs["trust_mono_mask"] = trust_mono_mask
        fuse_losses = self.compute_fuse_losses(inputs, outputs)

        if opt.mask_mvs_conf:
            D = opt.num_depth_bins
            cp = F.interpolate(cost_prob.unsqueeze(1), [D, opt.height, opt.width], mode="trilinear", align_corners=True)
            outputs["photo_conf_map"] = cp.max(2)[0] > opt.photo_conf
        if opt.mask_mvs_dist:
            outputs["dist_mask"] = outputs[("disp", 0)] > opt.dist_thres

        self.generate_images_pred(inputs, outputs, is_mvs=True)
        mvs_losses = self.compute_losses(inputs, outputs, is_mvs=True)
        for extra in (mono_losses, fuse_losses):
            for key, val in extra.items():
                mvs_losses[key] = mvs_losses[key] + val if key in mvs_losses else val
        return outputs, mvs_losses

    def predict_poses(self, inputs, features=None):
        """reference trainer.py:445-468"""
        outputs = {}
        pose_feats = {f_i: inputs["color_aug", f_i, 0] for f_i in self.opt.frame_ids}
        for f_i in self.opt.frame_ids[1:]:
            pair = [pose_feats[f_i], pose_feats[0]] if f_i < 0 else [pose_feats[0], pose_feats[f_i]]
            pose_inputs = [self.models["pose_encoder"](torch.cat(pair, 1))]
            axisangle, translation = self.models["pose"](pose_inputs)
            outputs[("axisangle", 0, f_i)] = axisangle
            outputs[("translation", 0, f_i)] = translation
            outputs[("cam_T_cam", 0, f_i)] = transformation_from_parameters(axisangle[:, 0], translation[:, 0],
                                                                            invert=(f_i < 0))
        for fi in self.matching_ids[1:]:
            inputs[("relative_pose", fi)] = outputs[("cam_T_cam", 0, fi)].clone().detach()
        return outputs

    def generate_images_pred(self, inputs, outputs, is_mvs=False):
        """Warp the neighbouring frames into the reference view (reference trainer.py:491-532)."""
        opt = self.opt
        K, inv_K = inputs[("K", 0)], inputs[("inv_K", 0)]
        if opt.fused_photometric:
            return self._generate_images_pred_fused(inputs, outputs, is_mvs)
        if is_mvs:
            depth_mvs = outputs["depth_mvs"]
            for frame_id in opt.frame_ids[1:]:
                T = outputs[("cam_T_cam", 0, frame_id)].detach()
                warped, _, oob = ops.warp_border(inputs[("color", frame_id, 0)], depth_mvs, K, inv_K, T, want_mask=True)
                outputs[("mvs_mask", frame_id)] = oob.bool()
                outputs[("mvs_color", frame_id)] = warped
            return
        for scale in opt.scales:
            depth = ops.disp_to_depth_up(outputs[("disp", scale)], opt.height, opt.width, opt.min_depth, opt.max_depth)
            outputs[("depth", 0, scale)] = depth
            for frame_id in opt.frame_ids[1:]:
                T = outputs[("cam_T_cam", 0, frame_id)]
                warped, pix, _ = ops.warp_border(inputs[("color", frame_id, 0)], depth, K, inv_K, T, want_pix=True)
                outputs[("sample", frame_id, scale)] = pix
                outputs[("color", frame_id, scale)] = warped
                outputs[("color_identity", frame_id, scale)] = inputs[("color", frame_id, 0)]

    def _generate_images_pred_fused(self, inputs, outputs, is_mvs):
        """generate_images_pred with the losses of the following compute_losses call formed in the same launch
        (ops.photometric_loss: warps of every frame and scale + SSIM/L1 + min over frames + auto-mask + masked mean).  Fills
        the same `outputs` entries; the loss scalars wait under a private key for compute_losses.  The warped images it
        stores are outputs, not graph nodes: gradients flow through the stashed losses."""
        opt = self.opt
        K, inv_K = inputs[("K", 0)], inputs[("inv_K", 0)]
        frames = opt.frame_ids[1:]
        target, srcs = self._packed_frames(inputs)
        B, _, H, W = inputs[("color", 0, 0)].shape
        common = dict(min_depth=opt.min_depth, max_depth=opt.max_depth, ssim_w=opt.ssim_lw, no_ssim=opt.no_ssim)
        if is_mvs:
            ext = None
            if opt.mask_mvs_conf:
                ext = outputs["photo_conf_map"].float()
            if opt.mask_mvs_dist:
                ext = outputs["dist_mask"].float() if ext is None else ext * outputs["dist_mask"].float()
            if opt.mask_mvs_geo:
                for f_id in frames:
                    ext = outputs[("geo_mask", f_id)] if ext is None else ext * outputs[("geo_mask", f_id)]  # KeyError upstream too
            res = ops.photometric_loss(target, srcs, [outputs[("cam_T_cam", 0, f)].detach() for f in frames], K, inv_K,
                                       [outputs["depth_mvs"]], mvs_mode=True, ext_mask=ext, want_oob=True, want_mask=True, **common)
            for i, f in enumerate(frames):
                outputs[("mvs_mask", f)] = res["oob"][i].bool()
                outputs[("mvs_color", f)] = res["warped"][0][i]
            outputs[("_photo", "mvs")] = res
            return
        ident = noise = None
        if not opt.disable_automasking:
            ident = ops.identity_loss(target, srcs, opt.ssim_lw, opt.no_ssim)
            noise = self._automask_noise((B, 1, H, W), len(opt.scales))   # one draw per scale, in the reference's order
        lazy = bool(getattr(opt, "lazy_sample_grids", 0)) and isinstance(outputs, StepOutputs)
        res = ops.photometric_loss(target, srcs, [outputs[("cam_T_cam", 0, f)] for f in frames], K, inv_K,
                                   [outputs[("disp", s)] for s in opt.scales], is_disp=True, ident_min=ident, noise=noise,
                                   want_pix=not lazy, **common)
        for si, scale in enumerate(opt.scales):
            outputs[("depth", 0, scale)] = res["depth"][si]
            for i, f in enumerate(frames):
                if lazy:   # the per-operation warp evaluates the same arithmetic: the same bits (tests/test_photo_fused.py)
                    def grid(f=f, scale=scale):
                        with torch.no_grad():
                            return ops.warp_border(inputs[("color", f, 0)], outputs[("depth", 0, scale)].detach(), K, inv_K,
                                                   outputs[("cam_T_cam", 0, f)].detach(), want_pix=True)[1]
                    outputs.lazy(("sample", f, scale), grid)
                else:
                    outputs[("sample", f, scale)] = res["pix"][si][i]
                outputs[("color", f, scale)] = res["warped"][si][i]
                outputs[("color_identity", f, scale)] = inputs[("color", f, 0)]
        outputs[("_photo", "mono")] = res

    def _packed_frames(self, inputs):
        """The step's frames in the fused kernels' image layout ((B,H,W,4) RGBx), packed once per step in one launch and kept in
        `inputs` under a private key.  Returns (target, [source frames])."""
        key = ("_rgbx", 0)
        if key not in inputs or inputs[key][0] is not inputs[("color", 0, 0)]:
            packed = ops.pack_rgbx([inputs[("color", f, 0)] for f in self.opt.frame_ids])
            inputs[key] = (inputs[("color", 0, 0)], packed)
        packed = inputs[key][1]
        return packed[0], packed[1:]

    def compute_reprojection_loss(self, pred, target, ssim_lw=None):
        """SSIM + L1 photometric loss (reference trainer.py:535-550) -> (B,1,H,W)."""
        w = self.opt.ssim_lw if ssim_lw is None else ssim_lw
        no_ssim = self.opt.no_ssim or w == 0  # weight 0: the reference still evaluates SSIM, the value is the L1 term
        return ops.reprojection_loss(pred, target, ssim_w=w, no_ssim=no_ssim)

    @staticmethod
    def compute_loss_masks(reprojection_loss, identity_reprojection_loss):
        """reference trainer.py:552-567"""
        if identity_reprojection_loss is None:
            return torch.ones_like(reprojection_loss)
        all_losses = torch.cat([reprojection_loss, identity_reprojection_loss], dim=1)
        return (torch.argmin(all_losses, dim=1, keepdim=True) == 0).float()

    def _automask_noise(self, shape, count=None):
        """The reference's tie-break: identity += randn(shape) * 1e-5 (trainer.py:698).  count: that many consecutive draws
        stacked (host mode: the same draws, in the same order, as `count` calls)."""
        if self.opt.automask_noise == "host":
            if count is None:
                return (torch.randn(shape) * 0.00001).to(self.device)
            return torch.stack([torch.randn(shape) * 0.00001 for _ in range(count)]).to(self.device)
        full = shape if count is None else (count,) + tuple(shape)
        return torch.randn(full, device=self.device) * 0.00001

    def _identity_losses(self, inputs, ssim_lw=None):
        target = inputs[("color", 0, 0)]
        return torch.cat([self.compute_reprojection_loss(inputs[("color", f, 0)], target, ssim_lw)
                          for f in self.opt.frame_ids[1:]], 1)

    def compute_fuse_losses(self, inputs, outputs):
        """Photometric (L1-only) loss of the fused depth (reference trainer.py:569-612)."""
        opt = self.opt
        depth_fuse = outputs["fused_depth"]
        K, inv_K = inputs[("K", 0)], inputs[("inv_K", 0)]
        target = inputs[("color", 0, 0)]
        if opt.fused_photometric:
            frames = opt.frame_ids[1:]
            ptarget, srcs = self._packed_frames(inputs)
            ident = noise = None
            if opt.mask_mvs_auto:
                ident = ops.identity_loss(ptarget, srcs, 0.0, True)
                noise = self._automask_noise((target.shape[0], 1) + tuple(target.shape[2:]), 1)
            res = ops.photometric_loss(ptarget, srcs, [outputs[("cam_T_cam", 0, f)].detach() for f in frames], K, inv_K,
                                       [depth_fuse], min_depth=opt.min_depth, max_depth=opt.max_depth, ssim_w=0.0, no_ssim=True,
                                       ident_min=ident, noise=noise, want_mask=True)
            for i, f in enumerate(frames):
                outputs[("mvs_color_fuse", f)] = res["warped"][0][i]
            outputs["reprojection_loss_mask"] = res["mask"][0]
            return {"fuse_reproj_loss": res["loss"][0], "loss": res["loss"][0]}
        reprojection_losses = []
        for frame_id in opt.frame_ids[1:]:
            T = outputs[("cam_T_cam", 0, frame_id)].detach()
            warped, _, _ = ops.warp_border(inputs[("color", frame_id, 0)], depth_fuse, K, inv_K, T)
            outputs[("mvs_color_fuse", frame_id)] = warped
            reprojection_losses.append(self.compute_reprojection_loss(warped, target, ssim_lw=0))
        reprojection_losses = torch.cat(reprojection_losses, 1)
        if opt.mask_mvs_auto:
            ident = self._identity_losses(inputs, ssim_lw=0)
            noise = self._automask_noise((target.shape[0], 1) + tuple(target.shape[2:]))
            loss, _, mask = ops.masked_min_loss(reprojection_losses, ident, noise)
        else:
            loss, _, mask = ops.masked_min_loss(reprojection_losses)
        outputs["reprojection_loss_mask"] = mask
        return {"fuse_reproj_loss": loss, "loss": loss}

    def compute_losses(self, inputs, outputs, is_mvs=False):
        """Reprojection + smoothness losses (reference trainer.py:614-724)."""
        opt = self.opt
        losses = {}
        target = inputs[("color", 0, 0)]
        B, _, H, W = target.shape

        fused = outputs.get(("_photo", "mvs" if is_mvs else "mono")) if opt.fused_photometric else None
        if is_mvs and fused is not None:
            if opt.mask_mvs_auto:
                self._automask_noise((B, 1, H, W))  # drawn and discarded upstream: the mask is overwritten (App. B-4)
            loss = fused["loss"][0]
            outputs["mvs_reprojection_loss"] = fused["min"][0]
            outputs["reprojection_loss_mask"] = fused["mask"][0]
            outputs["mvs_reproj_loss"] = loss
            if opt.mvs_smooth_loss:
                smooth_loss = ops.smooth_loss(outputs["depth_mvs"].unsqueeze(1), inputs[("color", 0, 0)], normalize=True)
                losses["mvs_smooth_loss/0"] = smooth_loss
                loss = loss + opt.disparity_smoothness * smooth_loss
            losses["loss"] = loss
            return losses
        if is_mvs:
            reprojection_losses = torch.cat([self.compute_reprojection_loss(outputs[("mvs_color", f)], target)
                                             for f in opt.frame_ids[1:]], 1)
            if opt.mask_mvs_auto:
                self._automask_noise((B, 1, H, W))  # drawn and discarded upstream: the mask is overwritten (App. B-4)
            ext = None
            if opt.mask_mvs_conf:
                ext = outputs["photo_conf_map"].float()
            if opt.mask_mvs_dist:
                ext = outputs["dist_mask"].float() if ext is None else ext * outputs["dist_mask"].float()
            if opt.mask_mvs_geo:
                for f_id in opt.frame_ids[1:]:
                    ext = outputs[("geo_mask", f_id)] if ext is None else ext * outputs[("geo_mask", f_id)]  # KeyError upstream too
            loss, min_reproj, mask = ops.masked_min_loss(reprojection_losses, ext_mask=ext, mvs_mode=True)
            outputs["mvs_reprojection_loss"] = min_reproj
            outputs["reprojection_loss_mask"] = mask
            outputs["mvs_reproj_loss"] = loss
            if opt.mvs_smooth_loss:
                smooth_loss = ops.smooth_loss(outputs["depth_mvs"].unsqueeze(1), inputs[("color", 0, 0)], normalize=True)
                losses["mvs_smooth_loss/0"] = smooth_loss
                loss = loss + opt.disparity_smoothness * smooth_loss
            losses["loss"] = loss
            return losses

        ident = None if (opt.disable_automasking or fused is not None) else self._identity_losses(inputs)  # same values at every scale
        total_loss = 0
        smooth = None
        if opt.fused_photometric:   # every level's smoothness term in one launch per pass
            smooth = ops.smooth_losses([outputs[("disp", sc)] for sc in opt.scales], [inputs[("color", 0, sc)] for sc in opt.scales])
        for si, scale in enumerate(opt.scales):
            if fused is not None:   # formed by generate_images_pred's launch
                loss, min_reproj = fused["loss"][si], fused["min"][si]
            else:
                reprojection_losses = torch.cat([self.compute_reprojection_loss(outputs[("color", f, scale)], target)
                                                 for f in opt.frame_ids[1:]], 1)
                if ident is not None:
                    loss, min_reproj, _ = ops.masked_min_loss(reprojection_losses, ident, self._automask_noise((B, 1, H, W)))
                else:
                    loss, min_reproj, _ = ops.masked_min_loss(reprojection_losses)
            if scale == 0:
                outputs["mono_reproj_loss"] = min_reproj
            smooth_loss = smooth[si] if smooth is not None else \
                ops.smooth_loss(outputs[("disp", scale)], inputs[("color", 0, scale)], normalize=True)
            losses["mono_smooth_loss/{}".format(scale)] = smooth_loss
            loss = loss + opt.disparity_smoothness * smooth_loss / (2 ** scale)
            total_loss = total_loss + loss
            losses["loss/{}".format(scale)] = loss
        losses["loss"] = total_loss / self.num_scales
        return losses

    # ------------------------------------------------------------------ checkpoints (reference trainer.py:796-880)
    def save_opts(self):
        models_dir = os.path.join(self.log_path, "models")
        os.makedirs(models_dir, exist_ok=True)
        with open(os.path.join(models_dir, "opt.json"), "w") as f:
            json.dump(self.opt.__dict__.copy(), f, indent=2)

    def save_model(self, save_step=False):
        """One {model}.pth plain state_dict per sub-model + adam.pth, rank 0 only (reference trainer.py:807-831): the
        reference evaluator loads each file with strict=True (evaluate_depth.py:118-174), so nothing but the module's own
        keys may be in it.  Folder: weights_{epoch} (weights_{epoch}_{step} with save_step), 'last' on the final epoch --
        'last' wins over both, as upstream (:816-817)."""
        if self.rank != 0:
            return
        name = "weights_{}_{}".format(self.epoch, self.step) if save_step else "weights_{}".format(self.epoch)
        if self.epoch == self.opt.num_epochs - 1:
            name = "last"
        folder = os.path.join(self.log_path, "models", name)
        os.makedirs(folder, exist_ok=True)
        for model_name, model in self.models.items():
            torch.save(model.state_dict(), os.path.join(folder, "{}.pth".format(model_name)))
        torch.save(self.model_optimizer.state_dict(), os.path.join(folder, "adam.pth"))

    def load_mono_model(self):
        for n in ("pose_encoder", "pose", "mono_encoder", "mono_depth"):
            self._load_one(self.opt.mono_weights_folder, n)

    def load_model(self):
        folder = os.path.expanduser(self.opt.load_weights_folder)
        assert os.path.isdir(folder), "Cannot find folder {}".format(folder)
        for n in self.opt.models_to_load:
            self._load_one(folder, n)
        adam = os.path.join(folder, "adam.pth")
        if os.path.isfile(adam):
            try:
                self.model_optimizer.load_state_dict(torch.load(adam, map_location="cpu"))
            except ValueError:
                print("Can't load Adam - using random")

    def _load_one(self, folder, n):
        # a misspelt model name or a missing file raises, as upstream (KeyError / FileNotFoundError at trainer.py:861-866):
        # silently training from random weights is the worse failure
        if n not in self.models:
            # names the reference knows but this configuration does not build ('up' without --convex_up, the pose networks under
            # --load_pose; 'up' is in the reference's default --models_to_load): skipped with a note.  Anything else is a typo
            # and raises, as upstream (KeyError at trainer.py:861)
            if n in ("up", "pose_encoder", "pose"):
                print("load_model: '%s' is not part of this configuration, skipped" % n)
                return
            raise KeyError("--models_to_load: unknown model %r (have %s)" % (n, ", ".join(self.models)))
        path = os.path.join(folder, "{}.pth".format(n))
        if not os.path.isfile(path):
            raise FileNotFoundError("checkpoint file not found: %s" % path)
        model_dict = self.models[n].state_dict()
        pretrained = torch.load(path, map_location="cpu")
        model_dict.update({k: v for k, v in pretrained.items() if k in model_dict})  # key intersection, as upstream
        self.models[n].load_state_dict(model_dict)
