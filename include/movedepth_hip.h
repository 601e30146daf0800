/*
 * movedepth_hip.h -- C ABI of libmovedepth_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for MOVEDepth's cost-volume + photometric-loss training hot path
 * (SURVEY.md section 8b).  The reference has no FFI layer: its "operator API" is the Python
 * signatures in movedepth/layers.py and movedepth/trainer.py.  Each entry point below names the
 * reference code it replaces (paths relative to /root/reference/movedepth/); the Python host
 * (movedepth_amd/layers.py, ops.py) binds these with ctypes and keeps the reference signatures.
 *
 * Conventions
 *   - plain C symbols; every pointer is a DEVICE pointer owned by the caller, never retained or freed;
 *   - all tensors are contiguous row-major float32 unless a stride argument says otherwise;
 *   - no hidden allocation: scratch is passed in (`ws`, size from the matching *_ws_bytes());
 *   - asynchronous: work is enqueued on `stream` (a hipStream_t; NULL = default stream); re-entrant
 *     across streams;
 *   - return 0 on success, a negative MD_E* code otherwise; md_last_error() gives the message of the
 *     last failure on the calling thread.
 */
#ifndef MOVEDEPTH_HIP_H
#define MOVEDEPTH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *md_stream_t; /* hipStream_t */

#define MD_OK 0
#define MD_EINVAL (-1)   /* bad argument / unsupported shape */
#define MD_ELAUNCH (-2)  /* HIP launch or runtime error      */

const char *md_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int md_abi_version(void);

/* depth-range schedule types (layers.py:264-279 `type`) */
#define MD_SCHED_INVERSE 0
#define MD_SCHED_LINEAR 1
#define MD_SCHED_LOG 2

/* ---- depth-range sampling ---------------------------------------------------------------
 * schedule_depth_rangev2 (layers.py:256-284) when ztrans == NULL,
 * schedule_depth_range_zv2 (layers.py:370-398) otherwise, with ztrans[b] = z_scale*T[b,2,3]
 * (trainer.py:340; one lookup frame).  prior [B,1,h,w] -> out [B,D,h,w].  no_grad in the reference. */
int md_schedule_depth_range(const float *prior, const float *ztrans, int B, int h, int w, int D, float scale_fac,
                            int type, float *out, md_stream_t stream);

/* ---- plane-sweep cost volume ------------------------------------------------------------
 * generate_costvol (layers.py:778-794: BackprojectDepth 581-586, Project3D 608-620, grid_sample zeros /
 * bilinear / align_corners=True, x ref) fused with the group mean of trainer.py:359
 * (group g = mean of channels {g, g+G, ...}).  G == C gives the reference's ungrouped (B,D,C,h,w) volume.
 *
 *   ref, src [B,C,h,w]; K, invK, pose [B,4,4] (scale-`prior_scale` intrinsics, pose = pose[:,0]);
 *   hypotheses: either hyp [B,D,h,w], or hyp == NULL and the schedule fused in from
 *   prior [B,1,h,w] (+ ztrans/scale_fac/sched_type as in md_schedule_depth_range; same arithmetic).
 *   out element (b,d,g,y,x) is written at out[b*out_sb + d*out_sd + g*out_sg + (y*w + x)*out_sp]
 *   (strides in floats), so the volume can be laid out (B,D,G,h,w) like the reference (sp = 1), (B,G,D,h,w)
 *   as the 3-D regulariser permutes it (resnet_encoder.py:257), or channels-last (B,D,h,w,G) (sg = 1, sp = G),
 *   the layout MIOpen's fast 3-D convolutions take -- without a permute copy in any case.
 *   feat_cl != 0: ref and src (and d_ref, d_src of the backward) are channels-last [B,h,w,C] -- what the 2-D encoder
 *   (resnet_encoder.py:360,387 FPN4) produces when it runs in torch.channels_last -- instead of [B,C,h,w]; taken by the
 *   channels-last-volume kernels only (sg = 1, sp = G, G = 8 or 16, C/G = 1, 2 or 4), MD_EINVAL otherwise.
 *   flags (ABI 17): MD_CV_FINE_SLICES = twice the hypothesis slices per work item (channels-last kernels): evens out launches
 *   whose tiles differ in cost (parallax: moderate poses 66 -> 62 us, driving scene 64.8 -> 61.5 at B=6, 48x160, D=96) at the
 *   price of a second window staging per item when they do not (58 -> 59.1 us); same results.  The caller's choice
 *   (movedepth_amd/ops.py BackwardPolicy: from the previous backward's census); unknown bits are MD_EINVAL.
 */
#define MD_CV_FINE_SLICES 2u
int md_costvol_fwd(const float *ref, const float *src, const float *K, const float *invK, const float *pose,
                   const float *hyp, const float *prior, const float *ztrans, float scale_fac, int sched_type,
                   int B, int C, int G, int h, int w, int D, int feat_cl, float *out, long long out_sb, long long out_sd,
                   long long out_sg, long long out_sp, unsigned flags, md_stream_t stream);

/* Autograd of md_costvol_fwd w.r.t. ref and src (the sampling grid is under no_grad, layers.py:784).
 * gout addressed with the same four strides; d_ref, d_src [B,C,h,w] ([B,h,w,C] with feat_cl) are overwritten (d_src is
 * zeroed, then accumulated with atomics; d_ref is stored directly when every pixel's hypotheses belong to one workgroup,
 * otherwise it takes the same route: then one fill instead of two when d_src == d_ref + B*C*h*w).  Poses that scatter a
 * tile's taps over more source cells than the kernel's window holds (an untrained pose network, a camera driving into the
 * scene): with feat_cl the kernel itself switches such hypothesis sub-slices to 16-byte gathers from L2 and queues their
 * d_src terms in LDS; with planar features they take the kernel's per-tap path (slower for such poses: the trainer hands over
 * channels-last features).
 *   flags (ABI 17; the library reads nothing from the process environment): MD_CV_GATHER_TABLE selects the build of the
 *   16 x 4-tile kernel that also merges those terms per source cell in LDS before they leave the CU: for phases with wild
 *   poses (5.1x instead of 8.7x the sane time), ~7 % slower when poses are sane; same results to float-atomic ordering.
 *   census (ABI 17; may be NULL): one device word, 8-byte aligned, overwritten by this launch with four counts --
 *   [63:46] hypothesis steps walked in gather mode / 4, [45:28] all hypothesis steps walked / 4, [27:14] windows staged,
 *   [13:0] segments (meaningful below 2^20 steps and 2^14 segments per launch) -- what a caller decides the flags of its NEXT launches from without
 *   a host synchronisation (an asynchronous copy to pinned memory read one call later; movedepth_amd/ops.py BackwardPolicy).
 *   shares, n_shares, cost (ABI 17; NULL / 0 / NULL: the library's own equal partition, nothing recorded; taken by the
 *   channels-last-volume kernels only): every workgroup of the backward is resident at once, so no hardware scheduler evens out
 *   what parallax makes uneven -- the launch lasts as long as its slowest tile (workgroup lifetimes max / mean 1.4-1.5 on driving
 *   scenes).  `cost`: device array of md_costvol_bwd_plan's `items` counters, overwritten with the shader cycles the launch spent
 *   per item (tile of 16 x 4 pixels of one sample, all D hypotheses).  `shares`: device array of n_shares pairs [lo, hi) in units
 *   of hypothesis steps (item * D + d), one pair per workgroup, together covering [0, items * D) exactly once -- a partition the
 *   caller computes from the `cost` of an earlier launch on similar poses (movedepth_amd/ops.py BackwardPolicy: equal COST per
 *   workgroup instead of equal steps; asynchronous copies, no host synchronisation).  A pair may span items and may be empty.
 * One launch either way, no state kept between calls (ABI 16: the pose pre-pass of ABI <= 15, with its ring of device-global
 * flag slots, is gone): re-entrant across streams like every other entry point. */
#define MD_CV_GATHER_TABLE 1u
/* Launch geometry of md_costvol_bwd* for a channels-last volume (B,D,h,w,G): the number of work items (what `cost` counts and
 * `shares` partitions, in units of D steps each) and the number of workgroups of the library's own partition (a sensible n_shares).
 * items = 0: this problem runs on the planar-era kernels, which take neither. */
int md_costvol_bwd_plan(int B, int C, int G, int h, int w, int D, int feat_cl, int *items, int *workgroups);
int md_costvol_bwd(const float *gout, long long g_sb, long long g_sd, long long g_sg, long long g_sp, const float *ref,
                   const float *src, const float *K, const float *invK, const float *pose, const float *hyp,
                   const float *prior, const float *ztrans, float scale_fac, int sched_type, int B, int C, int G,
                   int h, int w, int D, int feat_cl, float *d_ref, float *d_src, unsigned flags, unsigned long long *census,
                   const long long *shares, int n_shares, unsigned *cost, md_stream_t stream);

/* The same two entry points with 2-byte feature maps and volume (BASELINE configs 4 and 5: bf16 / fp16 mixed precision;
 * SURVEY 8d table rows 4-5): ref, src, out and gout are bf16 (`_bf16`) or IEEE half (`_f16`) bit patterns (uint16_t
 * here, no vendor types in the ABI); arithmetic, hypotheses, poses and the gradients d_ref / d_src stay fp32; strides
 * are in elements.  Half the volume bytes (147.5 MB algorithmic per launch at config 2's shape instead of 295.1). */
int md_costvol_fwd_bf16(const uint16_t *ref, const uint16_t *src, const float *K, const float *invK, const float *pose,
                        const float *hyp, const float *prior, const float *ztrans, float scale_fac, int sched_type, int B,
                        int C, int G, int h, int w, int D, int feat_cl, uint16_t *out, long long out_sb, long long out_sd,
                        long long out_sg, long long out_sp, unsigned flags, md_stream_t stream);
int md_costvol_bwd_bf16(const uint16_t *gout, long long g_sb, long long g_sd, long long g_sg, long long g_sp,
                        const uint16_t *ref, const uint16_t *src, const float *K, const float *invK, const float *pose,
                        const float *hyp, const float *prior, const float *ztrans, float scale_fac, int sched_type, int B,
                        int C, int G, int h, int w, int D, int feat_cl, float *d_ref, float *d_src, unsigned flags, unsigned long long *census,
                   const long long *shares, int n_shares, unsigned *cost, md_stream_t stream);
int md_costvol_fwd_f16(const uint16_t *ref, const uint16_t *src, const float *K, const float *invK, const float *pose,
                       const float *hyp, const float *prior, const float *ztrans, float scale_fac, int sched_type, int B,
                       int C, int G, int h, int w, int D, int feat_cl, uint16_t *out, long long out_sb, long long out_sd,
                       long long out_sg, long long out_sp, unsigned flags, md_stream_t stream);
int md_costvol_bwd_f16(const uint16_t *gout, long long g_sb, long long g_sd, long long g_sg, long long g_sp,
                       const uint16_t *ref, const uint16_t *src, const float *K, const float *invK, const float *pose,
                       const float *hyp, const float *prior, const float *ztrans, float scale_fac, int sched_type, int B,
                       int C, int G, int h, int w, int D, int feat_cl, float *d_ref, float *d_src, unsigned flags, unsigned long long *census,
                   const long long *shares, int n_shares, unsigned *cost, md_stream_t stream);

/* ---- frame-confidence fusion --------------------------------------------------------------
 * trainer.py:349-363: w_f = max_G softmax_G(mean_D vol_f); out = sum_f w_f vol_f / (1e-8 + sum_f w_f).
 * vols: N device pointers (host array) to grouped volumes addressed with (sb, sd, sg, sp) strides, as is out.
 * weights [N,B,h,w] may be NULL.  eval_mode != 0: the evaluation script's weight instead (evaluate_depth.py:236:
 * soft-max over D of the mean over G, then max over D); forward only. */
int md_fuse_fwd(const float *const *vols, int N, int B, int D, int G, int hw, long long sb, long long sd,
                long long sg, long long sp, int eval_mode, float *out, float *weights, md_stream_t stream);
/* Autograd of md_fuse_fwd (the weights are not detached in the reference). d_vols: N pointers. */
int md_fuse_bwd(const float *gout, const float *const *vols, int N, int B, int D, int G, int hw, long long sb,
                long long sd, long long sg, long long sp, float *const *d_vols, md_stream_t stream);

/* ---- reprojection warp ----------------------------------------------------------------------
 * generate_images_pred / compute_fuse_losses warp (trainer.py:501-507, 519-529, 575-580):
 * BackprojectDepth[0] -> Project3D[0] -> grid_sample(padding_mode='border', align_corners=True).
 * img [B,Ci,H,W]; depth [B,1,H,W]; K, invK, T [B,4,4].  pix [B,H,W,2] (normalised grid, may be NULL),
 * out [B,Ci,H,W], oob_mask [B,H,W] uint8 (any coordinate outside [-1,1], trainer.py:503; may be NULL). */
int md_warp_fwd(const float *img, const float *depth, const float *K, const float *invK, const float *T, int B,
                int Ci, int H, int W, float *pix, float *out, unsigned char *oob_mask, md_stream_t stream);
/* Gradients to depth [B,1,H,W] and T [B,4,4].  ws: md_warp_bwd_ws_bytes(B,H,W) bytes. */
size_t md_warp_bwd_ws_bytes(int B, int H, int W);
int md_warp_bwd(const float *gout, const float *img, const float *depth, const float *K, const float *invK,
                const float *T, int B, int Ci, int H, int W, float *d_depth, float *d_T, void *ws,
                md_stream_t stream);

/* Disparity pyramid level -> full-resolution depth: F.interpolate(bilinear, align_corners=False)
 * (trainer.py:512) followed by disp_to_depth (layers.py:400-409).  disp [B,1,h,w] -> depth [B,1,H,W]. */
int md_disp_to_depth_up_fwd(const float *disp, int B, int h, int w, int H, int W, float min_depth, float max_depth,
                            float *depth, md_stream_t stream);
int md_disp_to_depth_up_bwd(const float *g_depth, const float *disp, int B, int h, int w, int H, int W,
                            float min_depth, float max_depth, float *d_disp, md_stream_t stream);

/* ---- SSIM + L1 reprojection loss ----------------------------------------------------------
 * compute_reprojection_loss (trainer.py:535-550) with SSIM.forward (layers.py:663-677):
 * out = ssim_w * mean_c SSIM(pred,target) + (1-ssim_w) * mean_c |target - pred|; no_ssim -> L1 only.
 * pred, target [B,C,H,W] -> out [B,1,H,W].  md_ssim: the bare SSIM map [B,C,H,W]. */
int md_ssim(const float *x, const float *y, int B, int C, int H, int W, float *out, md_stream_t stream);
int md_reproj_loss_fwd(const float *pred, const float *target, int B, int C, int H, int W, float ssim_w, int no_ssim,
                       float *out, md_stream_t stream);
int md_reproj_loss_bwd(const float *gout, const float *pred, const float *target, int B, int C, int H, int W,
                       float ssim_w, int no_ssim, float *d_pred, md_stream_t stream);

/* ---- min over frames / auto-mask / masked mean ------------------------------------------
 * trainer.py:687-709 (mono) and 630-662 (MVS).  reproj, ident [B,N,H,W] (ident NULL = no automask);
 * noise [B,1,H,W] already scaled by 1e-5 (trainer.py:698), may be NULL; ext_mask [B,1,H,W] may be NULL;
 * mvs_mode != 0 replaces the automask by ones (trainer.py:647).
 * Outputs: min_reproj, mask [B,1,H,W]; loss[0] = sum(min*mask) / (sum(mask) + 1e-7); loss[1] = sum(mask).
 * ws: md_masked_min_ws_bytes(B,H,W) bytes. */
size_t md_masked_min_ws_bytes(int B, int H, int W);
int md_masked_min_fwd(const float *reproj, const float *ident, const float *noise, const float *ext_mask, int B,
                      int N, int H, int W, int mvs_mode, float *min_reproj, float *mask, float *loss, void *ws,
                      md_stream_t stream);
/* d loss / d reproj [B,N,H,W]; gloss: device scalar; loss: the [2] array written by the forward. */
int md_masked_min_bwd(const float *gloss, const float *reproj, const float *mask, const float *loss, int B, int N,
                      int H, int W, float *d_reproj, md_stream_t stream);

/* ---- the photometric chain fused: warp(s) + SSIM/L1 + min over frames + auto-mask + masked mean -------------------
 * One forward and one backward launch for everything generate_images_pred + compute_losses do for one group of losses
 * (trainer.py:491-532 + 675-724 mono, all scales; 498-509 + 621-673 MVS; 569-612 fused depth): per scale s and source
 * frame f  pred = grid_sample(src[f], Project3D(BackprojectDepth(depth_s), K, T[f]), border)  (layers.py:556-621),
 * loss_f = compute_reprojection_loss(pred, target) (trainer.py:535-550, layers.py:663-677), min over f, mask =
 * [min <= ident_min + noise_s] (trainer.py:698-705; ones with mvs_mode, trainer.py:647) x ext_mask, and
 * loss[s] = sum(min * mask) / (sum(mask) + 1e-7).  Same arithmetic per step as md_disp_to_depth_up / md_warp /
 * md_reproj_loss / md_masked_min above, which remain for callers that want the pieces.
 *
 * All images PACKED, [B,H,W,4] float (r, g, b, 0 per pixel: md_pack_rgbx converts [B,3,H,W] frames): target, src[f] and the
 * warped[s][f] outputs; maps [B,H,W] (= [B,1,H,W]); K, invK, T[f] [B,4,4].  NULL output pointers are skipped.
 * Domain of the bit-equality with those pieces: finite pixel values of magnitude below 2^20 (images in [0,1] or [0,255]) and a
 * depth range inside [2^-40, 2^40] -- the fused kernels form their quotients as the steps of the IEEE division without its range
 * scaling (csrc/md_photo.hpp), which is the identity there.  H * W < 2^27 (error otherwise). */
#define MD_PHOTO_MAX_FRAMES 4
#define MD_PHOTO_MAX_SCALES 4
typedef struct md_photo_desc {
    int B, H, W;       /* batch, full resolution */
    int F, S;          /* source frames (1..4), scales (1..4) */
    int is_disp;       /* dz[s] is a disparity pyramid level [B,1,dh[s],dw[s]]: F.interpolate(bilinear, align_corners=False) to
                          HxW + disp_to_depth inside (trainer.py:512-514); else dz[s] is a depth map [B,H,W] */
    int identity;      /* 1: no warp, pred_f = src[f]: the identity reprojection loss, min over frames -> mn[0] (trainer.py:690-696) */
    int mvs_mode;      /* 1: the auto-mask is replaced by ones (trainer.py:647) */
    int no_ssim;       /* 1: L1 only (--no_ssim, or ssim_lw = 0 at trainer.py:588) */
    float ssim_w, min_depth, max_depth;
    int dh[MD_PHOTO_MAX_SCALES], dw[MD_PHOTO_MAX_SCALES];
    const float *target;
    const float *src[MD_PHOTO_MAX_FRAMES];
    const float *T[MD_PHOTO_MAX_FRAMES];
    const float *K, *invK;
    const float *dz[MD_PHOTO_MAX_SCALES];
    const float *ident_min;  /* [B,H,W] min over frames of the identity loss, or NULL (no auto-mask) */
    const float *noise;      /* [S,B,H,W] tie-break noise already scaled by 1e-5 (trainer.py:698), or NULL */
    const float *ext_mask;   /* [B,H,W] or NULL (photo_conf_map / dist_mask, trainer.py:650-657) */
    /* forward outputs */
    float *warped[MD_PHOTO_MAX_SCALES][MD_PHOTO_MAX_FRAMES];  /* ("color",f,s) / ("mvs_color",f); REQUIRED by the backward */
    float *pix[MD_PHOTO_MAX_SCALES][MD_PHOTO_MAX_FRAMES];     /* ("sample",f,s) [B,H,W,2] */
    unsigned char *oob[MD_PHOTO_MAX_FRAMES];                  /* ("mvs_mask",f): any coordinate outside [-1,1] (trainer.py:503), scale 0 */
    float *depth_out[MD_PHOTO_MAX_SCALES];                    /* ("depth",0,s) */
    float *mn[MD_PHOTO_MAX_SCALES];                           /* min over frames of the reprojection loss */
    float *mask[MD_PHOTO_MAX_SCALES];                         /* the mask as floats */
    unsigned char *sel[MD_PHOTO_MAX_SCALES];                  /* arg-min frame | 0x80 where mask != 0; REQUIRED by the backward */
    float *loss;                                              /* [S][2]: loss, sum(mask) */
    /* backward only */
    const float *gloss[MD_PHOTO_MAX_SCALES];  /* d L / d loss[s]: device scalars, NULL = 0 */
    float *d_dz[MD_PHOTO_MAX_SCALES];         /* gradient w.r.t. dz[s], same shape */
    float *d_T[MD_PHOTO_MAX_FRAMES];          /* [B,4,4] summed over scales, or NULL (T detached, trainer.py:499) */
} md_photo_desc;
/* [B,3,H,W] -> [B,H,W,4] for n <= 5 images in one launch (once per step and frame) */
int md_pack_rgbx(const float *const *imgs, int n, int B, int H, int W, float *const *out, md_stream_t stream);
size_t md_photo_fwd_ws_bytes(int B, int S, int H, int W);
int md_photo_fwd(const md_photo_desc *desc, void *ws, md_stream_t stream);
size_t md_photo_bwd_ws_bytes(int B, int S, int F, int H, int W, int is_disp);
int md_photo_bwd(const md_photo_desc *desc, void *ws, md_stream_t stream);

/* ---- edge-aware smoothness ---------------------------------------------------------------
 * get_smooth_loss (layers.py:630-643) on disp / (mean_hw(disp) + 1e-7) (trainer.py:712-714) when
 * normalize != 0.  disp [B,1,h,w]; img [B,Ci,h,w]; loss: device scalar.
 * ws: md_smooth_ws_bytes(B,h,w) bytes, shared by forward and backward of one call pair. */
size_t md_smooth_ws_bytes(int B, int h, int w);
int md_smooth_fwd(const float *disp, const float *img, int B, int Ci, int h, int w, int normalize, float *loss,
                  void *ws, md_stream_t stream);
int md_smooth_bwd(const float *gloss, const float *disp, const float *img, int B, int Ci, int h, int w,
                  int normalize, float *d_disp, void *ws, md_stream_t stream);

/* The same for every disparity pyramid level of one compute_losses call (trainer.py:712-714 inside the scale loop) in one
 * launch per pass: disp[s] [B,1,h[s],w[s]], img[s] [B,Ci,h[s],w[s]], loss[s]; gloss[s] device scalars (NULL = 0).
 * ws: md_smooth_multi_ws_bytes(B, S). */
size_t md_smooth_multi_ws_bytes(int B, int S);
int md_smooth_multi_fwd(const float *const *disp, const float *const *img, const int *h, const int *w, int S, int B, int Ci,
                        int normalize, float *loss, void *ws, md_stream_t stream);
int md_smooth_multi_bwd(const float *const *gloss, const float *const *disp, const float *const *img, const int *h, const int *w,
                        int S, int B, int Ci, int normalize, float *const *d_disp, void *ws, md_stream_t stream);

/* sizeof(md_photo_desc): lets a binding check its struct layout against the library's */
size_t md_photo_desc_bytes(void);

/* ---- post-volume ops ("next" rows) --------------------------------------------
 * Fused softmax over D (trainer.py:367) + entropy (layers.py:862-863) + localmax (layers.py:796-812).
 * logits [B,D,h,w]; min_inv, max_inv [B,h,w] (the caller passes 1/hyp[:,-1], 1/hyp[:,0], trainer.py:371).
 * Outputs: prob [B,D,h,w] (may be NULL), entropy [B,1,h,w] (may be NULL), depth [B,h,w]. */
int md_softmax_entropy_localmax_fwd(const float *logits, int B, int D, int h, int w, int radius,
                                    const float *min_inv, const float *max_inv, float *prob, float *entropy,
                                    float *depth, md_stream_t stream);
int md_softmax_entropy_localmax_bwd(const float *g_depth, const float *g_entropy, const float *logits, int B, int D,
                                    int h, int w, int radius, const float *min_inv, const float *max_inv,
                                    float *d_logits, md_stream_t stream);

/* Convex upsampling (layers.py:200-214): depth [B,h,w], mask [B,9*s*s,h,w] (s = 2**scale) -> out [B,s*h,s*w]
 * (softmax over the 9 taps of a zero-padded 3x3 unfold).  Backward: d_depth [B,h,w], d_mask like mask;
 * ws: md_convex_upsample_bwd_ws_bytes(B,h,w) bytes (per-cell sums of the terms of the depth gradient). */
int md_convex_upsample_fwd(const float *depth, const float *mask, int B, int h, int w, int scale, float *out,
                           md_stream_t stream);
size_t md_convex_upsample_bwd_ws_bytes(int B, int h, int w);
int md_convex_upsample_bwd(const float *gout, const float *depth, const float *mask, int B, int h, int w, int scale,
                           float *d_depth, float *d_mask, void *ws, md_stream_t stream);

/* ---- reg3d's last layer, the producer of the logits above (the 3-D conv hand-off) ------------
 * `prob = nn.Conv3d(base_channels, 1, 3, stride=1, padding=1, bias=False)` (networks/resnet_encoder.py:254,
 * applied :277): 3x3x3, zero padding, ONE output channel.
 * x, dx: channels-last volume [B,D,H,W,C] (torch channels_last_3d storage of a [B,C,D,H,W] tensor), C in {8,16};
 * y, gy: [B,D,H,W] (= [B,1,D,H,W]); weight element (tap k = (kd*3+kh)*3+kw, channel c) at
 * wt[k*w_stride_k + c*w_stride_c]  (contiguous [1,C,3,3,3]: 1, 27; channels_last_3d: C, 1); dwt likewise.
 * ws: md_conv3d_c1_bwd_weight_ws_bytes() bytes (one partial per workgroup, summed in a fixed order). */
int md_conv3d_c1_fwd(const float *x, const float *wt, long long w_stride_k, long long w_stride_c, float *y, int B,
                     int C, int D, int H, int W, md_stream_t stream);
int md_conv3d_c1_bwd_data(const float *gy, const float *wt, long long w_stride_k, long long w_stride_c, float *dx,
                          int B, int C, int D, int H, int W, md_stream_t stream);
size_t md_conv3d_c1_bwd_weight_ws_bytes(int B, int C, int D, int H, int W);
int md_conv3d_c1_bwd_weight(const float *x, const float *gy, float *dwt, long long dw_stride_k, long long dw_stride_c,
                            void *ws, size_t ws_bytes, int B, int C, int D, int H, int W, md_stream_t stream);

/* ---- reg3d's first layer, the consumer of the grouped cost volume ------------------------------------------
 * `conv0.conv = nn.Conv3d(16, 16, 3, stride=1, padding=1, bias=False)` (networks/resnet_encoder.py:231 through
 * ConvBnReLU3D, applied :258).  x, dx, y, gy: channels-last volumes [B,D,H,W,16]; weight element (co, ci, tap k =
 * (kd*3+kh)*3+kw) at wt[co*w_stride_co + ci*w_stride_ci + k*w_stride_k] (contiguous [16,16,3,3,3]: 432, 27, 1;
 * channels_last_3d: 432, 1, 16), dwt likewise.  Ci == Co == 16 only (MD_EINVAL otherwise).  fp32 MFMA
 * (v_mfma_f32_16x16x4_f32, exact fp32 products and sums); the weight gradient is reduced in a fixed order.
 * x_planar / dx_planar != 0: that volume is planar [B,16,D,H,W] instead -- the cost volume's `bgd` layout as
 * md_costvol_fwd writes it and md_costvol_bwd reads its gradient (sb = 16*D*h*w, sg = D*h*w, sd = h*w, sp = 1), so
 * the plane-sweep kernels run in their fastest layout and no layout copy sits between them and this layer. */
int md_conv3d_c16_fwd(const float *x, int x_planar, const float *wt, long long w_stride_co, long long w_stride_ci,
                      long long w_stride_k, float *y, int B, int Ci, int Co, int D, int H, int W, md_stream_t stream);
int md_conv3d_c16_bwd_data(const float *gy, const float *wt, long long w_stride_co, long long w_stride_ci,
                           long long w_stride_k, float *dx, int dx_planar, int B, int Ci, int Co, int D, int H, int W,
                           md_stream_t stream);
size_t md_conv3d_c16_bwd_weight_ws_bytes(int B, int D, int H, int W);
int md_conv3d_c16_bwd_weight(const float *x, int x_planar, const float *gy, float *dwt, long long dw_stride_co,
                             long long dw_stride_ci, long long dw_stride_k, void *ws, size_t ws_bytes, int B, int Ci, int Co,
                             int D, int H, int W, md_stream_t stream);

/* ---- reg3d's interior 3x3x3 layers with Ci, Co multiples of 16 (`conv2`, `conv4`, `conv6` of networks/resnet_encoder.py:233-239, applied :260-262:
 * ConvBnReLU3D(32, 32), (64, 64), (128, 128), stride 1, padding 1, bias=False), as sums over 16 x 16 channel blocks on the
 * bf16 x 3 kernels of the 16 -> 16 layer: forward / data gradient one launch per input block covering every output block, the later
 * input blocks added in the kernel's epilogue; weight gradient one launch for all block pairs.  Channels-last volumes [B,D,H,W,C] only, 16-byte aligned; weights addressed as above (any strides);
 * Ci, Co <= 256.  ws: md_conv3d_cb_bwd_weight_ws_bytes() bytes (the partials of every block pair: one launch, then a fixed-order finish per pair). */
int md_conv3d_cb_fwd(const float *x, const float *wt, long long w_stride_co, long long w_stride_ci, long long w_stride_k,
                     float *y, int B, int Ci, int Co, int D, int H, int W, md_stream_t stream);
int md_conv3d_cb_bwd_data(const float *gy, const float *wt, long long w_stride_co, long long w_stride_ci,
                          long long w_stride_k, float *dx, int B, int Ci, int Co, int D, int H, int W, md_stream_t stream);
size_t md_conv3d_cb_bwd_weight_ws_bytes(int B, int Ci, int Co, int D, int H, int W);
int md_conv3d_cb_bwd_weight(const float *x, const float *gy, float *dwt, long long dw_stride_co, long long dw_stride_ci,
                            long long dw_stride_k, void *ws, size_t ws_bytes, int B, int Ci, int Co, int D, int H, int W,
                            md_stream_t stream);

/* ---- BatchNorm with synchronised statistics, every normalisation layer of the model ------------------------------------
 * The reference's data-parallel path (trainer.py:69-135, train_movedepth.sh:15 --ddp) converts every BatchNorm2d / BatchNorm3d
 * to torch.nn.SyncBatchNorm: batch statistics over the GLOBAL batch.  These five entry points are that layer for a
 * channels-last tensor x[row * C + c] (torch.channels_last, channels_last_3d, or an (N, C) matrix; C a multiple of 4 up to
 * 4096; rows = N * spatial size), one launch each, deterministic reductions:
 *   md_bn_stats       sums[0:C] = sum x, sums[C:2C] = sum x^2, in DOUBLE  -> the caller all-reduces the 2C sums over the ranks
 *   md_bn_apply       mean / biased variance from the (global) sums and n_total rows; y = (x - mean) * invstd * gamma + beta,
 *                     relu != 0: followed by max(0, .); writes stat[0:C] = mean, stat[C:2C] = invstd for the backward and
 *                     updates running_mean / running_var (unbiased variance, `momentum`; either may be NULL)
 *   md_bn_eval        evaluation mode: the running statistics instead
 *   md_bn_bwd_reduce  sums[0:C] = sum dz (= dbeta), sums[C:2C] = sum dz * xhat (= dgamma), dz = dy * [z > 0] when relu
 *                     -> the caller all-reduces a copy of the 2C sums
 *   md_bn_bwd_dx      dx = gamma * invstd * (dz - sums[0:C] / n_total - xhat * sums[C:2C] / n_total)
 * ws: md_bn_ws_bytes() bytes, ZEROED ONCE by the caller and then reused (the reduction kernels leave it zeroed); one
 * workspace per stream. */
size_t md_bn_ws_bytes(void);
/* dtype = element type of the activations and their gradients (x, y, dy, dx): 0 float, 1 bf16, 2 fp16 -- what torch.autocast hands
 * a BatchNorm layer and expects back (ABI 14).  Sums, statistics, gamma / beta and their gradients are float / double in every case. */
int md_bn_stats(const void *x, int dtype, long long nrows, int C, double *sums, void *ws, md_stream_t stream);
int md_bn_apply(const void *x, int dtype, const double *sums, long long n_total, float eps, float momentum, const float *gamma,
                const float *beta, int relu, float *running_mean, float *running_var, float *stat, void *y, long long nrows,
                int C, md_stream_t stream);
int md_bn_eval(const void *x, int dtype, const float *running_mean, const float *running_var, float eps, const float *gamma,
               const float *beta, int relu, void *y, long long nrows, int C, md_stream_t stream);
/* sums_copy (may be NULL): a second copy of the 2C sums, for the caller to all-reduce in place while `sums` stays this rank's
 * d_beta / d_gamma (ABI 15; saves a copy kernel per layer and step) */
int md_bn_bwd_reduce(const void *dy, const void *x, int dtype, const float *stat, const float *gamma, const float *beta, int relu,
                     long long nrows, int C, float *sums, float *sums_copy, void *ws, md_stream_t stream);
int md_bn_bwd_dx(const void *dy, const void *x, int dtype, const float *stat, const float *gamma, const float *beta, int relu,
                 const float *sums, long long n_total, long long nrows, int C, void *dx, md_stream_t stream);

/* ---- training-mode BatchNorm + ReLU (+ residual) of the regulariser's two full-resolution layers -------------
 * conv0's BatchNorm3d + ReLU (networks/resnet_encoder.py:231 through ConvBnReLU3D) and conv11's BatchNorm3d + ReLU
 * followed by `x = conv0 + self.conv11(x)` (:249-252, :264).  x, y, res, dy, dx: channels-last volumes flattened to
 * [nvox,16].  Two-phase so that a data-parallel caller can all-reduce the 32 sums in between (SyncBatchNorm):
 *   md_bn_relu_stats      -> sums[0:16] = sum x, sums[16:32] = sum x^2, in DOUBLE (E[x^2] - E[x]^2 from float sums loses
 *                            (mean/std)^2 * 6e-8 of the variance); the caller all-reduces them, md_bn_relu_finalize turns them
 *                            into mean, biased variance, invstd
 *   md_bn_relu_apply      y = max(0, (x - mean) * invstd * gamma + beta) [+ res]   (res may be NULL; y may alias x)
 *   md_bn_relu_bwd_reduce -> sums[0:16] = sum dz (= dbeta), sums[16:32] = sum dz * xhat (= dgamma), dz = dy * [z > 0]
 *   md_bn_relu_bwd_dx     dx = gamma * invstd * (dz - sums[0]/n_total - xhat * sums[1]/n_total)
 * ws: md_bn_relu_ws_bytes() bytes.  Reductions are two-stage in a fixed order (fp64 final sum). */
size_t md_bn_relu_ws_bytes(void);
int md_bn_relu_stats(const float *x, long long nvox, int C, double *sums, void *ws, md_stream_t stream);
/* mean / invstd from the (all-reduced) sums over n_total voxels; running_mean / running_var (may be NULL) updated in place
 * with `momentum` (unbiased variance), as F.batch_norm does in training */
int md_bn_relu_finalize(const double *sums, long long n_total, int C, float eps, float momentum, float *mean, float *invstd,
                        float *running_mean, float *running_var, md_stream_t stream);
int md_bn_relu_apply(const float *x, const float *mean, const float *invstd, const float *gamma, const float *beta,
                     const float *res, long long nvox, int C, float *y, md_stream_t stream);
int md_bn_relu_bwd_reduce(const float *dy, const float *x, const float *mean, const float *invstd, const float *gamma,
                          const float *beta, long long nvox, int C, float *sums, void *ws, md_stream_t stream);
int md_bn_relu_bwd_dx(const float *dy, const float *x, const float *mean, const float *invstd, const float *gamma,
                      const float *beta, const float *sums, long long n_total, long long nvox, int C, float *dx,
                      md_stream_t stream);

/* ---- pose parameters -> 4x4 -------------------------------------------------------------------
 * transformation_from_parameters (layers.py:412-429; rot_from_axisangle :479-518, get_translation_matrix :464-477):
 * axisangle, translation [B,3] -> T [B,4,4]; invert != 0: R^T . T(-t), else T(t) . R.  The reference's
 * axis = v / (|v| + 1e-7) is kept.  Backward: gT [B,4,4] -> d_axisangle, d_translation [B,3] (d|v|/dv = 0 at v = 0). */
int md_pose_matrix_fwd(const float *axisangle, const float *translation, int B, int invert, float *T, md_stream_t stream);
int md_pose_matrix_bwd(const float *gT, const float *axisangle, const float *translation, int B, int invert,
                       float *d_axisangle, float *d_translation, md_stream_t stream);

/* ---- standalone geometry, for call compatibility (the hot kernels fuse these; forward only) -------
 * BackprojectDepth.forward (layers.py:581-586): depth [Bs,h*w], invK [nk,4,4] (nk = 1 or Bs) -> cam_points [Bs,4,h*w].
 * Project3D.forward (layers.py:601-621): points [Bs,4,h*w], K, T [nk,4,4] -> pix [Bs,h,w,2] in [-1,1]. */
int md_backproject(const float *depth, const float *invK, int Bs, int nk, int h, int w, float *cam_points,
                   md_stream_t stream);
int md_project3d(const float *points, const float *K, const float *T, int Bs, int nk, int h, int w, float eps,
                 float *pix, md_stream_t stream);

/* ---- per-kernel timing (measurement only; not part of the reference's interface) ---------------------------------
 * md_kernel_timing_enable(mask): classes of entry points to time -- 2 = plane sweep + md_conv3d_*, 4 = photometric / smoothness / packing,
 * 8 = md_bn_*; 1 = all of them; 0 = off (drops the records).  A timed dispatch costs its stream about 5 us (completion signal with timestamps, two
 * events): 444 timed BatchNorm launches per training step read as +2.6 ms of step time, so bench.py times class 8 only on request.
 * A timed entry point attaches a HIP start / stop event pair to its (main) KERNEL dispatch
 * (hipExtLaunchKernelGGL: the dispatch's own begin / end timestamps, on the launch stream; not the argument checks or memsets); md_kernel_timing_enable(0) stops and drops the records.
 * md_kernel_timing_read(name, ...) synchronises the recorded events of entry point `name` ("md_costvol_fwd",
 * "md_costvol_bwd", with _bf16 / _f16 suffixes) and returns their average / minimum duration in microseconds and their
 * count (0 launches: avg = min = 0).  bench.py's roofline figure is read from here: events recorded from Python around
 * the ctypes call also time the host's launch latency whenever the GPU has run dry (77 vs 57 us inside the training step). */
int md_kernel_timing_enable(int on);
int md_kernel_timing_read(const char *name, double *avg_us, double *min_us, int *launches);
/* Work counters of the channels-last plane-sweep kernels (diagnostics).  md_costvol_stats(1, NULL) switches them on and clears
 * them; md_costvol_stats(on, out8) first synchronises the device and copies the 8 counters accumulated so far: [0] segments,
 * [1] windows staged (sub-slices), [2] window-fit attempts, [3] (lane, sub-slice) pairs that had to redo steps from global memory, [4] wave-level
 * cell-change blocks executed (forward: coefficient rebuilds, backward: flushes), [5] lanes active in them, [6] wave-steps. */
int md_costvol_stats(int enable, unsigned long long *out8);
/* ... and, while they are on, the lifetime in shader cycles of every workgroup of the launches since the last clear (entry
 * blockIdx.x, summed over launches; the first 8192 workgroups): copies min(cap, 8192) entries, returns that count (0: never enabled).
 * Read it BEFORE md_costvol_stats(., out8), which clears.  max / mean over the non-zero entries = how evenly a launch's work is spread. */
int md_costvol_stats_wg(unsigned long long *out, int cap);
/* the individual durations, in launch order: fills us[0 .. min(n, cap)) and returns n (>= 0) */
int md_kernel_timing_list(const char *name, double *us, int cap);

#ifdef __cplusplus
}
#endif
#endif /* MOVEDEPTH_HIP_H */
