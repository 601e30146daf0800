// The photometric chain as ONE forward and ONE backward kernel per group of losses:
//
//   disparity pyramid level -> full-resolution depth   (trainer.py:512-514, layers.py:400-409)
//   -> BackprojectDepth / Project3D / grid_sample(border)  for every source frame   (trainer.py:519-529, 501-507, 575-580)
//   -> SSIM + L1 reprojection loss per frame            (layers.py:663-677, trainer.py:535-550)
//   -> min over frames, auto-mask against the identity loss (+ tie-break noise), external masks, masked mean
//                                                       (trainer.py:687-709 mono, 630-662 MVS, 583-612 fused depth)
//
// Round 2 ran this as warp_fwd -> ssim_fwd -> (cat) -> masked_min -> finish per (scale, frame): each warped image was written,
// re-read by the loss kernel, re-read by its backward, its gradient written and re-read by the warp's backward -- 12 times per
// training step, 1.67 ms of 13-31 us kernels at 10-15 % of the HBM roofline.  Here the warped pixel goes from the gather
// straight into LDS, the loss, the minimum and the mask are formed from there, and all scales of one compute_losses call are
// one launch (grid z = scale x sample).  The warped images are still written once (they are outputs of the reference's
// generate_images_pred, and the backward reads them instead of re-sampling a two-pixel halo).
//
//   forward : tile 32 x 16 pixels + halo 1 (reflection: ReflectionPad2d(1) of the SSIM windows), 256 threads.  Phase 1: every
//             halo position is back-projected, projected and sampled for each frame -> LDS as float4 (3 channels).  Phase 2: a
//             thread owns two vertically adjacent pixels and reads 4 x 3 taps per image with ds_read_b128.
//   backward: tile 32 x 8 + halo 2.  d loss / d pred collapses to three coefficient maps per channel (mu_x, E[x^2], E[xy]; see
//             ssim.hip), built at halo 1 only where the frame is the selected minimum and the mask is set; the gradient at a
//             pixel is their reflection-adjoint 3x3 sum, handed in registers to the warp's backward (gradients to depth -- or to
//             the up-sampled disparity -- and, per workgroup, 12 partial sums of dL/dP per frame).
//   The disparity gradient is then gathered through the adjoint of the bilinear up-sampling by up_adjoint_kernel (all scales
//   in one launch); the dL/dP partials are summed in double and turned into dL/dT = K^T dP by a one-block-per-(frame, sample)
//   finish.  Every reduction is two-stage in a fixed order: results are bit-reproducible.
//
// Image layout.  Every image these kernels touch -- target, source frames, the warped frames they write -- is PACKED: (B,H,W,4)
// float, one 16-byte RGBx record per pixel (md_pack_rgbx converts the (B,3,H,W) input frames once per step).  With planar images
// the forward issued 24 dword gathers + 3 target loads + 6 dword stores per halo position and the kernels were bound by the
// vector-memory instruction rate (GRBM_TA_BUSY 90 % of the kernel, ~45 cycles per wave-level access); packed it is 8 + 1 + 2
// 16-byte accesses.  A (B,3,H,W) view of a packed image (strides 4HW, 1, 4W, 4) is an ordinary torch tensor: that is what
// outputs[("color", f, s)] holds.
#include "md_photo.hpp"

namespace {
using namespace mdp;

constexpr int MAXF = MD_PHOTO_MAX_FRAMES, MAXS = MD_PHOTO_MAX_SCALES;
constexpr int FT_W = 32, FT_H = 16, FP_W = FT_W + 2, FP_H = FT_H + 2, FP_N = FP_W * FP_H;  // forward tile, + halo 1
constexpr int BT_W = 32, BT_H = 8;                                                         // backward tile
#ifndef MD_PHOTO_BWD_WAVES
#define MD_PHOTO_BWD_WAVES 4     // waves per SIMD the backward is compiled for (up to two source frames): 128 registers, 9 spilled at F = 2; 3: 135, none (A/B: 201 -> 177 us)
#endif
#ifndef MD_PHOTO_CAM_SGPR
#define MD_PHOTO_CAM_SGPR 1   // camera matrices in scalar registers (A/B)
#endif
#ifndef MD_PHOTO_BWD_EARLY
#define MD_PHOTO_BWD_EARLY 3   // bit 0: frame 0's tap loads before the coefficient phase; bit 1: frame f+1's before frame f's reductions (A/B)
#endif
#ifndef MD_PHOTO_FWD_WAVES
#define MD_PHOTO_FWD_WAVES 3     // waves per SIMD the forward is compiled for (up to two source frames); A/B: tools/ab_build.sh
#endif
#ifndef MD_PHOTO_BWD_ONEPASS
#define MD_PHOTO_BWD_ONEPASS 1   // coefficient maps of all frames in one pass (photo_bwd_kernel); 0: one masked pass per frame (A/B)
#endif
constexpr int B2_W = BT_W + 4, B2_H = BT_H + 4, B2_N = B2_W * B2_H;                        // + halo 2 (images)
constexpr int B1_W = BT_W + 2, B1_H = BT_H + 2, B1_N = B1_W * B1_H;                        // + halo 1 (coefficients)

// Quotients of launch constants every thread would otherwise form with a 12-instruction IEEE division each (7 per thread: 4 % of the
// backward's vector instructions): formed once on the host, in float -- the same correctly rounded values.
struct PhotoConsts {
    float min_disp, max_disp;     // 1 / max_depth, 1 / min_depth
    float rw, rh;                 // 1 / (W - 1), 1 / (H - 1)
    float upx[MAXS], upy[MAXS];   // dw[s] / W, dh[s] / H: F.interpolate's scales
};
inline PhotoConsts photo_consts(const md_photo_desc *d) {
    PhotoConsts k{};
    k.min_disp = 1.f / d->max_depth; k.max_disp = 1.f / d->min_depth;
    k.rw = 1.f / (float)(d->W - 1); k.rh = 1.f / (float)(d->H - 1);
    for (int s = 0; s < MAXS; ++s) {
        k.upx[s] = d->is_disp && s < d->S ? (float)d->dw[s] / (float)d->W : 0.f;
        k.upy[s] = d->is_disp && s < d->S ? (float)d->dh[s] / (float)d->H : 0.f;
    }
    return k;
}

__device__ __forceinline__ float f4c(const float4 &v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : v.z); }

// The four records of a bilinear footprint, from clamped addresses: one 32-bit byte offset from the image's (wave-uniform) base
// and three additions -- as four 64-bit element addresses the loads cost 15 vector instructions per footprint.
__device__ __forceinline__ void load_taps(const v4f *__restrict__ img, int W, const Tap &t, bool vx1, bool vy1, v4f &v00, v4f &v01, v4f &v10,
                                          v4f &v11) {
    const char *base = reinterpret_cast<const char *>(img);
    const unsigned o00 = (unsigned)(t.y0 * W + t.x0) * 16u, dx = vx1 ? 16u : 0u, dy = vy1 ? (unsigned)W * 16u : 0u;
    v00 = *reinterpret_cast<const v4f *>(base + o00);
    v01 = *reinterpret_cast<const v4f *>(base + (o00 + dx));
    v10 = *reinterpret_cast<const v4f *>(base + (o00 + dy));
    v11 = *reinterpret_cast<const v4f *>(base + (o00 + dy + dx));
}

// grid_sample(border, align_corners=True) of a packed image at the clipped position (the forward of warp.hip).
// The four loads are unconditional, from clamped addresses: a load under a (divergent) branch sits in its own basic block behind
// an s_waitcnt, which made the taps SERIAL memory round trips (the 4-scale forward spent 72 % of its wave cycles parked; so does
// warp_fwd_kernel, 13 us for 0.7 M pixels).  A tap beyond the last column / row needs no select either: border clipping puts
// such a position ON the integer W-1 / H-1, so its weight wx1 / wy1 is exactly 0, and the value read from the clamped address
// instead is a finite pixel: the product is the exact zero md_warp_fwd adds there.  The interpolation is grid_sample's in the
// reference's order (as warp_fwd_kernel and the oracle's tap_sample: bit-equal results), on the (R, G) and (B, pad) halves of
// the 16-byte records: 8 packed instructions instead of 12 + 9 selects.  The pad lane of the result is not defined.
__device__ __forceinline__ v4f sample_border(const v4f *__restrict__ img, int W, int H, const Clip &c) {
    const Tap t = md_make_tap(c.ix, c.iy, W, H);
    const bool vx1 = t.x0 + 1 < W, vy1 = t.y0 + 1 < H;  // x0, y0 are in range after clipping
    const float wx0 = 1.f - t.wx1, wy0 = 1.f - t.wy1;
    v4f v00, v01, v10, v11;
    load_taps(img, W, t, vx1, vy1, v00, v01, v10, v11);
    v4f o;
    {
#pragma clang fp contract(off)
        const float w00 = wy0 * wx0, w01 = wy0 * t.wx1, w10 = t.wy1 * wx0, w11 = t.wy1 * t.wx1;
        o.xy = md_fma(v11.xy, (v2f)w11, md_fma(v10.xy, (v2f)w10, md_fma(v01.xy, (v2f)w01, v00.xy * w00)));
        o.zw = md_fma(v11.zw, (v2f)w11, md_fma(v10.zw, (v2f)w10, md_fma(v01.zw, (v2f)w01, v00.zw * w00)));
    }
    return o;
}

// Work item of this workgroup.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8: observed, relied on for
// speed only), each with its own 4 MB L2; the images of one call are 26 MB at config 2.  Every XCD therefore gets a
// contiguous eighth of the (sample, tile) items -- 3.3 MB of images: L2-resident -- and runs the S scales of an item back to
// back, so a tile's target / source lines miss to the Infinity Cache once instead of once per scale (PMC before: 314 MB
// fetched by the 4-scale forward for 30 MB of inputs, waves parked 72 % of their cycles on those misses).
__device__ __forceinline__ bool photo_item(int nitems, int S, int &item, int &s) {
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int per = (nitems + 7) >> 3;
    item = xcd * per + j / S;
    s = j % S;
    return j / S < per && item < nitems;
}
__host__ inline unsigned photo_grid(int nitems, int S) { return 8u * (unsigned)((nitems + 7) / 8) * (unsigned)S; }

// P = (K T)[:3] of every frame and inv_K[:3,:3], computed once per workgroup (12 F + 9 threads), read back by everyone
template <int F>
__device__ __forceinline__ void load_cams_lds(const md_photo_desc &a, int b, float *camS /*[F*12+9]*/) {
    const int tid = threadIdx.x;
    if (tid < 12 * F) {
        const int f = tid / 12, i = (tid % 12) / 4, j = tid % 4;
        const float *K = a.K + b * 16, *T = a.T[f] + b * 16;
        float sum = 0.f;
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int k = 0; k < 4; ++k) sum = sum + K[i * 4 + k] * T[k * 4 + j];   // as md_load_cam_plain (= the oracle's kt_rows)
        }
        camS[tid] = sum;
    } else if (tid < 12 * F + 9) {
        const int k = tid - 12 * F;
        camS[tid] = a.invK[b * 16 + (k / 3) * 4 + k % 3];
    }
    __syncthreads();
}
template <int F>
__device__ __forceinline__ void load_cams(const md_photo_desc &a, int b, float *camS, CamMats (&cam)[F]) {
    load_cams_lds<F>(a, b, camS);
#pragma unroll
    for (int f = 0; f < F; ++f) {
#pragma unroll
        for (int k = 0; k < 12; ++k) cam[f].P[k] = camS[f * 12 + k];
#pragma unroll
        for (int k = 0; k < 9; ++k) cam[f].iK[k] = camS[F * 12 + k];
    }
}
// the same with the x and y rows of P as register pairs (md_project_pk)
template <int F, bool SG>
__device__ __forceinline__ void load_cams_pk(const md_photo_desc &a, int b, float *camS, CamPk (&cam)[F], float (&iK)[9]) {
    load_cams_lds<F>(a, b, camS);
    // the matrices are the same in every lane: v_readfirstlane_b32 moves them to scalar registers, which the vector instructions
    // read directly (12 F + 9 vector registers less: what lets the backward keep a frame's taps in flight, see issue_taps); where
    // the scalar file holds them without spilling (SG: forward up to two frames, backward up to three)
    auto uni = [](float v) { return (MD_PHOTO_CAM_SGPR && SG) ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))) : v; };
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cam[f].Pxy[j] = (v2f){uni(camS[f * 12 + j]), uni(camS[f * 12 + 4 + j])};
            cam[f].Pz[j] = uni(camS[f * 12 + 8 + j]);
        }
#pragma unroll
    for (int k = 0; k < 9; ++k) iK[k] = uni(camS[F * 12 + k]);
}

// ------------------------------------------------------------------------------------------------ forward
template <int F, bool IDENT>
__global__ __launch_bounds__(256, (F <= 2 ? MD_PHOTO_FWD_WAVES : 2)) void photo_fwd_kernel(const md_photo_desc a, const PhotoConsts kc, float *__restrict__ ws) {
    extern __shared__ float4 lds[];
    float4 *tg = lds;          // target, halo 1
    float4 *wp = lds + FP_N;   // wp[f * FP_N + i]: prediction of frame f
    __shared__ float red[4][2];
    __shared__ float camS[MAXF * 12 + 9];
    const int tid = threadIdx.x;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + FT_W - 1) / FT_W, tiles = tiles_x * ((H + FT_H - 1) / FT_H);
    int item, s;
    if (!photo_item(a.B * tiles, IDENT ? 1 : a.S, item, s)) return;
    const int b = item / tiles, tile = item % tiles;
    const int x0 = (tile % tiles_x) * FT_W, y0 = (tile / tiles_x) * FT_H;
    CamPk cam[F];
    float iK[9];
    if (!IDENT) load_cams_pk<F, (F <= 2)>(a, b, camS, cam, iK);
    const size_t HW = (size_t)H * W;
    const float4 *tgt = reinterpret_cast<const float4 *>(a.target) + (size_t)b * HW;
    const float min_disp = kc.min_disp, max_disp = kc.max_disp;
    const v2f wh1 = {(float)(W - 1), (float)(H - 1)}, rwh = {kc.rw, kc.rh};

    // ---- phase 1: every halo position -> target + F predictions in LDS.  Unrolled: the loads of a thread's (up to) three
    // positions are independent and should all be in flight together -- the kernel is bound by memory latency, not bandwidth
#pragma unroll
    for (int i = tid; i < FP_N; i += 256) {
        const int cy = i / FP_W, cx = i % FP_W;
        const int gy = y0 - 1 + cy, gx = x0 - 1 + cx;
        const int py = clampi(reflect1(gy, H), 0, H - 1), px = clampi(reflect1(gx, W), 0, W - 1);
        const size_t p = (size_t)py * W + px;
        tg[i] = tgt[p];
        if (IDENT) {
#pragma unroll
            for (int f = 0; f < F; ++f) wp[f * FP_N + i] = (reinterpret_cast<const float4 *>(a.src[f]) + (size_t)b * HW)[p];
        } else {
            // the pixel this thread also writes the per-pixel outputs of (the interior of the tile, inside the image)
            const bool own = cy >= 1 && cy <= FT_H && cx >= 1 && cx <= FT_W && gy < H && gx < W;
            float depth;
            if (a.is_disp) {
                const float sd = disp_up_sd_s(a.dz[s] + (size_t)b * a.dh[s] * a.dw[s], a.dh[s], a.dw[s], kc.upy[s], kc.upx[s], py, px, min_disp, max_disp);
#if MD_PHOTO_FAST_DIV
                depth = md_div_fixup(md_div_core(1.f, sd, md_rcp_newton(sd)), sd, 1.f);   // sd in [1 / max_depth, 1 / min_depth]
#else
                depth = 1.f / sd;
#endif
            } else {
                depth = a.dz[s][(size_t)b * HW + p];
            }
            if (own && a.depth_out[s]) a.depth_out[s][(size_t)b * HW + p] = depth;
            float r0, r1, r2;   // inv_K is the same for every frame (md_ray's operations)
            {
#pragma clang fp contract(off)
                r0 = fmaf(iK[1], (float)py, iK[0] * (float)px) + iK[2];
                r1 = fmaf(iK[4], (float)py, iK[3] * (float)px) + iK[5];
                r2 = fmaf(iK[7], (float)py, iK[6] * (float)px) + iK[8];
            }
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const ProjPk pr = md_project_pk(cam[f], r0, r1, r2, depth, wh1, rwh);
                const Clip c = clip_border(pr.i.x, pr.i.y, W, H);
                v4f o4 = sample_border(reinterpret_cast<const v4f *>(a.src[f]) + (size_t)b * HW, W, H, c);
                o4.w = 0.f;
                reinterpret_cast<v4f *>(wp)[f * FP_N + i] = o4;
                if (own) {
                    if (a.warped[s][f]) (reinterpret_cast<v4f *>(a.warped[s][f]) + (size_t)b * HW)[p] = o4;
                    if (a.pix[s][f]) *reinterpret_cast<v2f *>(a.pix[s][f] + ((size_t)b * HW + p) * 2) = pr.g;
                    if (s == 0 && a.oob[f])
                        a.oob[f][(size_t)b * HW + p] = (pr.g.x < -1.f || pr.g.x > 1.f || pr.g.y < -1.f || pr.g.y > 1.f) ? 1 : 0;
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 2: two vertically adjacent pixels per thread
    const int tx = tid & 31, rp = tid >> 5;
    const bool use_ssim = !a.no_ssim;
    const float ssim_w = a.ssim_w;
    float lossf[2][F];
    {
        // No contraction in this block: the window sums follow the reference's order exactly -- AvgPool2d accumulates the nine
        // taps row-major in float, over tensors of already-rounded products x*x, x*y (layers.py:663-672) -- because
        // sigma = E[x^2] - mu^2 cancels to 1e-3 of its terms on smooth images and any other association moves the loss by ~1e-4.
#pragma clang fp contract(off)
        // pixel k = 0 uses LDS rows 2rp .. 2rp+2, pixel k = 1 rows 2rp+1 .. 2rp+3; both walk rows and columns in ascending order.
        // Every sum lives in three register pairs: (R, G) of pixel 0, (R, G) of pixel 1 -- the low half of the 16-byte LDS record --
        // and (B of pixel 0, B of pixel 1): the two middle rows feed both pixels, so their B tap is added to both halves of that
        // pair in one instruction.  30 packed / single instructions per sum of 54 additions (md_photo.hpp: same bits).
        const v4f *tgv = reinterpret_cast<const v4f *>(tg);
        v2f muy[3], ey2[3];   // [0], [1]: (R, G) of pixel k; [2]: B of both
        v4f yc[2];
#pragma unroll
        for (int g = 0; g < 3; ++g) muy[g] = ey2[g] = (v2f)0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = (2 * rp + r) * FP_W + tx;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const v4f t = tgv[o + j];
                const v2f t0 = t.xy, q0 = t0 * t0;
                const float qz = t.z * t.z;
                if (r <= 2) { muy[0] += t0; ey2[0] += q0; }
                if (r >= 1) { muy[1] += t0; ey2[1] += q0; }
                if (r == 0) { muy[2].x += t.z; ey2[2].x += qz; }
                else if (r == 3) { muy[2].y += t.z; ey2[2].y += qz; }
                else { muy[2] += (v2f)t.z; ey2[2] += (v2f)qz; }
                if (j == 1 && r == 1) yc[0] = t;
                if (j == 1 && r == 2) yc[1] = t;
            }
        }
        v2f my[3], ey[3];   // window means of the target
        if (use_ssim) {
#pragma unroll
            for (int g = 0; g < 3; ++g) { my[g] = div9(muy[g]); ey[g] = div9(ey2[g]); }
        }
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const v4f *wf = reinterpret_cast<const v4f *>(wp + f * FP_N);
            v2f mux[3], ex2[3], exy[3];
            v4f xc[2];
#pragma unroll
            for (int g = 0; g < 3; ++g) mux[g] = ex2[g] = exy[g] = (v2f)0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = (2 * rp + r) * FP_W + tx;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const v4f x = wf[o + j], t = tgv[o + j];
                    const v2f x0 = x.xy, q0 = x0 * x0, p0 = x0 * t.xy;
                    const float qz = x.z * x.z, pz = x.z * t.z;
                    if (r <= 2) { mux[0] += x0; ex2[0] += q0; exy[0] += p0; }
                    if (r >= 1) { mux[1] += x0; ex2[1] += q0; exy[1] += p0; }
                    if (r == 0) { mux[2].x += x.z; ex2[2].x += qz; exy[2].x += pz; }
                    else if (r == 3) { mux[2].y += x.z; ex2[2].y += qz; exy[2].y += pz; }
                    else { mux[2] += (v2f)x.z; ex2[2] += (v2f)qz; exy[2] += (v2f)pz; }
                    if (j == 1 && r == 1) xc[0] = x;
                    if (j == 1 && r == 2) xc[1] = x;
                }
            }
            // SSIM of three pairs: (R, G) of pixel 0, (R, G) of pixel 1, B of both
            v2f sv[3] = {(v2f)0.f, (v2f)0.f, (v2f)0.f};
            if (use_ssim) {
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    MomentsT<v2f> m;
                    m.mux = div9(mux[g]); m.ex2 = div9(ex2[g]); m.exy = div9(exy[g]);
                    m.muy = my[g]; m.ey2 = ey[g];
                    const v2f raw = ssim_from_t<v2f>(m);
                    sv[g] = (v2f){fminf(fmaxf(raw.x, 0.f), 1.f), fminf(fmaxf(raw.y, 0.f), 1.f)};  // torch.clamp(., 0, 1)
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                // channel sums in the reference's order: ((R + G) + B), from zero
                float l1 = 0.f, ss = 0.f;
                l1 += fabsf(yc[k].x - xc[k].x); l1 += fabsf(yc[k].y - xc[k].y); l1 += fabsf(yc[k].z - xc[k].z);
                ss += sv[k].x; ss += sv[k].y; ss += (k == 0 ? sv[2].x : sv[2].y);
                // mean over the three channels: an IEEE quotient by the constant 3 (div_by: 3 instructions, same bits)
                l1 = div_by(l1, 3.f, 1.f / 3.f);
                ss = div_by(ss, 3.f, 1.f / 3.f);
                lossf[k][f] = use_ssim ? ssim_w * ss + (1.f - ssim_w) * l1 : l1;
            }
        }
    }
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int qy = y0 + 2 * rp + k, qx = x0 + tx;
        if (qy < H && qx < W) {
            const size_t q = (size_t)b * HW + (size_t)qy * W + qx;
            float r = lossf[k][0];
            int am = 0;
#pragma unroll
            for (int f = 1; f < F; ++f)
                if (lossf[k][f] < r) { r = lossf[k][f]; am = f; }  // torch.min: first index wins ties
            float m = 1.f;
            if (!IDENT) {
                if (a.ident_min && !a.mvs_mode) {
                    float id = a.ident_min[q];
                    if (a.noise) id += a.noise[(size_t)s * a.B * HW + q];
                    m = (r <= id) ? 1.f : 0.f;  // argmin([reproj, identity]) == 0, first index wins ties
                }
                if (a.ext_mask) m *= a.ext_mask[q];
                if (a.mask[s]) a.mask[s][q] = m;
                if (a.sel[s]) a.sel[s][q] = (unsigned char)(am | (m != 0.f ? 0x80 : 0));
                num += r * m;
                den += m;
            }
            if (a.mn[s]) a.mn[s][q] = r;
        }
    }
    if (!IDENT) {
        num = md_wave_sum(num);
        den = md_wave_sum(den);
        if ((tid & 63) == 0) { red[tid >> 6][0] = num; red[tid >> 6][1] = den; }
        __syncthreads();
        if (tid == 0) {
            const size_t blk = ((size_t)s * a.B + b) * tiles + tile;
            ws[blk * 2] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
            ws[blk * 2 + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        }
    }
}

// loss[s] = sum(min * mask) / (sum(mask) + 1e-7), loss[S + ...]: one block per scale, fixed order, double
__global__ __launch_bounds__(256) void photo_fwd_finish_kernel(const float *__restrict__ ws, int per_scale, float *__restrict__ loss) {
    __shared__ double red[4][2];
    const int s = blockIdx.x;
    double num = 0.0, den = 0.0;
    for (int k = threadIdx.x; k < per_scale; k += 256) {
        num += (double)ws[((size_t)s * per_scale + k) * 2];
        den += (double)ws[((size_t)s * per_scale + k) * 2 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { num += __shfl_xor(num, o, 64); den += __shfl_xor(den, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = num; red[threadIdx.x >> 6][1] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        num = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        den = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        loss[s * 2] = (float)num / ((float)den + 1e-7f);
        loss[s * 2 + 1] = (float)den;
    }
}

// ------------------------------------------------------------------------------------------------ backward
// three waves per SIMD (<= 168 registers) for up to two source frames, two beyond
template <int F>
__global__ __launch_bounds__(256, (F <= 2 ? MD_PHOTO_BWD_WAVES : 2)) void photo_bwd_kernel(const md_photo_desc a, const PhotoConsts kc, float *__restrict__ gup, float *__restrict__ wsP) {
    extern __shared__ float4 lds[];
    float4 *tg = lds;                         // target, halo 2
    float4 *wp = lds + B2_N;                  // wp[f * B2_N + i]: warped frame f, halo 2
    float4 *cf = lds + (1 + F) * B2_N;        // cf[k * B1_N + i], k = 0..2: coefficient maps A, B, C of the current frame
    __shared__ float redf[16 * 12];
    __shared__ float camS[MAXF * 12 + 9];
    __shared__ int anyon[MAXF];               // does any halo-1 position of this tile hold coefficients of frame f?
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + BT_W - 1) / BT_W, nblk = tiles_x * ((H + BT_H - 1) / BT_H);
    int item, s;
    if (!photo_item(a.B * nblk, a.S, item, s)) return;
    const int b = item / nblk, blk = item % nblk;
    const int x0 = (blk % tiles_x) * BT_W, y0 = (blk / tiles_x) * BT_H;
    const size_t HW = (size_t)H * W;
    const float min_disp = kc.min_disp, max_disp = kc.max_disp;
    const v2f wh1 = {(float)(W - 1), (float)(H - 1)}, rwh = {kc.rw, kc.rh};
    const bool use_ssim = !a.no_ssim && a.ssim_w != 0.f;
    const float wl1 = a.no_ssim ? 1.f : (1.f - a.ssim_w);
    // d loss_s / d (min * mask)[p] = gloss_s / (sum(mask) + 1e-7)
    const float gscale = a.gloss[s] ? a.gloss[s][0] / (a.loss[s * 2 + 1] + 1e-7f) : 0.f;
    const unsigned char *selb = a.sel[s] + (size_t)b * HW;
    // a fractional external mask scales the gradient; the selection byte only carries mask != 0
    const float *maskb = a.mask[s] ? a.mask[s] + (size_t)b * HW : nullptr;
    const int tx = tid % BT_W, ty = tid / BT_W;
    const int qx = x0 + tx, qy = y0 + ty;
    const bool qvalid = qx < W && qy < H;
    const size_t q = (size_t)qy * W + qx;

    // ---- Every global load whose address does not depend on another load is issued HERE, before the first barrier: the images
    // with halo 2, the selection bytes of the thread's coefficient positions, the pixel's own byte, its disparity taps.  The
    // kernel is bound by memory latency (its inputs were written a whole forward + 0.5 GB of cost-volume traffic ago: HBM, not
    // the Infinity Cache); as first written it chained nine dependent round trips per workgroup (camera matrices -> two staging
    // iterations -> per frame: selection bytes, then the twelve taps) and ran 316 us inside the training step.
    constexpr int NST = (B2_N + 255) / 256, NCF = (B1_N + 255) / 256;
    v4f st[NST][1 + F];   // a first-class vector: an array of HIP's float4 structs stayed in scratch
    {
        const v4f *tgt = reinterpret_cast<const v4f *>(a.target) + (size_t)b * HW;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = tid + 256 * k;   // clamped: the last round's surplus threads re-read position B2_N - 1
            const int ii = i < B2_N ? i : B2_N - 1;
            const int py = clampi(reflect1(y0 - 2 + ii / B2_W, H), 0, H - 1), px = clampi(reflect1(x0 - 2 + ii % B2_W, W), 0, W - 1);
            const size_t p = (size_t)py * W + px;
            st[k][0] = tgt[p];
#pragma unroll
            for (int f = 0; f < F; ++f) st[k][1 + f] = (reinterpret_cast<const v4f *>(a.warped[s][f]) + (size_t)b * HW)[p];
        }
    }
    unsigned char selc[NCF];
    float mskc[NCF];
#pragma unroll
    for (int k = 0; k < NCF; ++k) {
        const int i = tid + 256 * k;
        const int py = y0 - 1 + i / B1_W, px = x0 - 1 + i % B1_W;
        const bool in = i < B1_N && py >= 0 && py < H && px >= 0 && px < W;
        const size_t p = in ? (size_t)py * W + px : 0;
        selc[k] = selb[p];
        mskc[k] = maskb ? maskb[p] : 1.f;
        if (!in) selc[k] = 0;
    }
    const unsigned char selq = selb[qvalid ? q : 0];
    const float maskq = maskb ? maskb[qvalid ? q : 0] : 1.f;
    // geometry of the pixel, shared by the frames
    float depth = 1.f, sd = 1.f;
    {
        const int cqy = qvalid ? qy : 0, cqx = qvalid ? qx : 0;
        if (a.is_disp) {
            sd = disp_up_sd_s(a.dz[s] + (size_t)b * a.dh[s] * a.dw[s], a.dh[s], a.dw[s], kc.upy[s], kc.upx[s], cqy, cqx, min_disp, max_disp);
#if MD_PHOTO_FAST_DIV
            depth = md_div_fixup(md_div_core(1.f, sd, md_rcp_newton(sd)), sd, 1.f);   // the forward's bits: the same taps
#else
            depth = 1.f / sd;
#endif
        } else {
            depth = a.dz[s][(size_t)b * HW + (size_t)cqy * W + cqx];
        }
    }
    CamPk cam[F];
    float iK[9];
    load_cams_pk<F, (F <= 3)>(a, b, camS, cam, iK);   // (barrier inside)
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = tid + 256 * k;
        if (i < B2_N) {
            reinterpret_cast<v4f *>(tg)[i] = st[k][0];
#pragma unroll
            for (int f = 0; f < F; ++f) reinterpret_cast<v4f *>(wp)[f * B2_N + i] = st[k][1 + f];
        }
    }
    // Coefficient maps A, B, C at halo 1 (d SSIM term / d mu_x, E[x^2], E[xy] of the window centred there, times the upstream
    // factor), non-zero only where a frame is the selected minimum and the mask is set.  ONE pass for all frames
    // (MD_PHOTO_BWD_ONEPASS): every position belongs to at most one frame -- the arg-min -- so a lane takes ITS frame's window
    // from LDS (wp + frame * B2_N) and the moments are computed once per position.  As first written the phase ran once per
    // frame with the positions of the other frames masked off: with a selection that varies from pixel to pixel (every step of a
    // run from random weights; profiles/r04_bench_line.json) both passes executed in nearly every wave, twice the phase's 27 taps
    // x 5 moments x 3 channels per position for the same results.  only_f >= 0: the per-frame form (A/B builds).
    auto coeff_phase = [&](int only_f) {
#pragma unroll
        for (int k = 0; k < NCF; ++k) {
            const int i = tid + 256 * k;
            if (i >= B1_N) break;
            const int cy = i / B1_W, cx = i % B1_W;
            float4 A = make_float4(0.f, 0.f, 0.f, -1.f), Bc = make_float4(0.f, 0.f, 0.f, 0.f), Cc = Bc;
            const int fs = selc[k] & 0x7f;
            const bool on = (selc[k] & 0x80) && fs < F && (only_f < 0 || fs == only_f);
            if (on) {
                const v4f *wf = reinterpret_cast<const v4f *>(wp + fs * B2_N), *tgv = reinterpret_cast<const v4f *>(tg);
                const float gs = gscale * mskc[k] * a.ssim_w * (1.f / 3.f);
                // each tap's 16-byte record is read from LDS once for its three channels (read per channel, the kernel issued 400
                // DS instructions per wave and was LDS-bound: SQ_LDS_IDX_ACTIVE 60 % of its duration); (R, G) as a register pair,
                // B on its own (md_photo.hpp); sums in the reference's order, no contraction (see the forward)
                MomentsT<v2f> m2 = {(v2f)0.f, (v2f)0.f, (v2f)0.f, (v2f)0.f, (v2f)0.f};
                MomentsT<float> mz = {0.f, 0.f, 0.f, 0.f, 0.f};
                {
#pragma clang fp contract(off)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const v4f x4 = wf[(cy + dy) * B2_W + cx + dx], y4 = tgv[(cy + dy) * B2_W + cx + dx];
                            const v2f x = x4.xy, y = y4.xy;
                            m2.mux += x; m2.muy += y; m2.ex2 += x * x; m2.ey2 += y * y; m2.exy += x * y;
                            mz.mux += x4.z; mz.muy += y4.z; mz.ex2 += x4.z * x4.z; mz.ey2 += y4.z * y4.z; mz.exy += x4.z * y4.z;
                            // one window row of LDS records in flight at a time: all 18 (72 registers) beside frame 0's taps spilled
                            if ((MD_PHOTO_BWD_EARLY & 1) && dx == 2) __builtin_amdgcn_sched_barrier(0);
                        }
                    m2.mux = div9(m2.mux); m2.muy = div9(m2.muy); m2.ex2 = div9(m2.ex2); m2.ey2 = div9(m2.ey2); m2.exy = div9(m2.exy);
                    mz.mux = div9(mz.mux); mz.muy = div9(mz.muy); mz.ex2 = div9(mz.ex2); mz.ey2 = div9(mz.ey2); mz.exy = div9(mz.exy);
                }
                v2f A2, B2, C2, raw2;
                float Az, Bz, Cz, rawz;
                ssim_coeffs_t<v2f>(m2, gs, A2, B2, C2, raw2);
                ssim_coeffs_t<float>(mz, gs, Az, Bz, Cz, rawz);
                const bool k0 = md_in01(raw2.x), k1 = md_in01(raw2.y), k2 = md_in01(rawz);
                A = make_float4(k0 ? A2.x : 0.f, k1 ? A2.y : 0.f, k2 ? Az : 0.f, (float)fs);   // pad lane: the frame these belong to
                Bc = make_float4(k0 ? B2.x : 0.f, k1 ? B2.y : 0.f, k2 ? Bz : 0.f, 0.f);
                Cc = make_float4(k0 ? C2.x : 0.f, k1 ? C2.y : 0.f, k2 ? Cz : 0.f, 0.f);
            }
            cf[i] = A; cf[B1_N + i] = Bc; cf[2 * B1_N + i] = Cc;
            if (on) anyon[fs] = 1;
        }
    };
    float r0, r1, r2;   // md_ray's operations
    {
#pragma clang fp contract(off)
        r0 = fmaf(iK[1], (float)qy, iK[0] * (float)qx) + iK[2];
        r1 = fmaf(iK[4], (float)qy, iK[3] * (float)qx) + iK[5];
        r2 = fmaf(iK[7], (float)qy, iK[6] * (float)qx) + iK[8];
    }
    float d_depth = 0.f;
    // ---- a frame's four bilinear taps and what the warp's backward needs of the projection.  The loads are unconditional (see
    // sample_border), for every pixel whether or not a gradient will reach it, and REQUESTED EARLY: frame 0's before the coefficient
    // phase, frame f+1's before frame f's reductions and barrier -- a request issued at the top of its own frame's iteration was
    // waited for with nothing left to overlap (the kernel spent 84 of its 203 us per step on those two round trips).
    struct Taps {
        v4f t0, t1, t2, t3;   // north-west, north-east, south-west, south-east
        float pzz, wx1, wy1;
        v2f puv, pg;
    };
    auto issue_taps = [&](int f) {
        Taps o;
        const ProjPk pr = md_project_pk(cam[f], r0, r1, r2, depth, wh1, rwh);
        const Clip c = clip_border(pr.i.x, pr.i.y, W, H);
        const Tap t = md_make_tap(c.ix, c.iy, W, H);
        const bool vx1 = t.x0 + 1 < W, vy1 = t.y0 + 1 < H;
        o.pzz = pr.zz; o.puv = pr.uv; o.pg = (v2f){c.gmx, c.gmy}; o.wx1 = t.wx1; o.wy1 = t.wy1;
        load_taps(reinterpret_cast<const v4f *>(a.src[f]) + (size_t)b * HW, W, t, vx1, vy1, o.t0, o.t1, o.t2, o.t3);
        __builtin_amdgcn_sched_barrier(0);   // keep the loads up here: the scheduler otherwise sinks them to their first use
        return o;
    };
    Taps cur;
    if (MD_PHOTO_BWD_EARLY & 1) cur = issue_taps(0);
    // weights of the 3 x 3 coefficient gather, the same for every frame.  ReflectionPad2d adjoint: row 1 is also pad row -1 (seen by
    // window row 0), row H-2 also pad row H; a neighbour outside the image contributes nothing.
    float gwy[3], gwx[3];
#pragma unroll
    for (int d = -1; d <= 1; ++d) {
        const int py = qy + d, px = qx + d;
        gwy[d + 1] = (py < 0 || py >= H) ? 0.f : 1.f + ((qy == 1 && py == 0) ? 1.f : 0.f) + ((qy == H - 2 && py == H - 1) ? 1.f : 0.f);
        gwx[d + 1] = (px < 0 || px >= W) ? 0.f : 1.f + ((qx == 1 && px == 0) ? 1.f : 0.f) + ((qx == W - 2 && px == W - 1) ? 1.f : 0.f);
    }
    if (tid < F) anyon[tid] = 0;
    __syncthreads();
    if (use_ssim && MD_PHOTO_BWD_ONEPASS) {
        coeff_phase(-1);
        __syncthreads();
    }

#pragma unroll
    for (int f = 0; f < F; ++f) {
        const float4 *wf = wp + f * B2_N;
        if (!(MD_PHOTO_BWD_EARLY & 1) && f == 0) cur = issue_taps(0);
        if (!(MD_PHOTO_BWD_EARLY & 2) && f > 0) cur = issue_taps(f);
        Taps nxt;
        // ---- coefficient maps at halo 1, only where frame f is the selected minimum and the mask is set
        if (use_ssim && !MD_PHOTO_BWD_ONEPASS) {
            coeff_phase(f);
            __syncthreads();
        }
        float dP[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) dP[k] = 0.f;
        bool has = false;   // does a gradient reach this pixel through frame f?
        if (qvalid) {
            // ---- d loss / d pred_f[q][c]: (R, G) and (B, pad) halves of the coefficient records, two channels per instruction
            v2f gA[2] = {(v2f)0.f, (v2f)0.f}, gB[2] = {(v2f)0.f, (v2f)0.f}, gC[2] = {(v2f)0.f, (v2f)0.f};
            if (use_ssim && anyon[f]) {   // (a tile without a coefficient of this frame: masked out, or the other frame's everywhere)
                const v4f *cfv = reinterpret_cast<const v4f *>(cf);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int o = (ty + dy) * B1_W + tx + dx;
                        const v4f A = cfv[o], Bc = cfv[B1_N + o], Cc = cfv[2 * B1_N + o];
                        // one-pass maps hold every frame's coefficients, the frame in A's pad lane: those of the other frames
                        // contribute exact zeros, as the zero-filled per-frame maps did
                        const v2f wgt = (v2f)((!MD_PHOTO_BWD_ONEPASS || A.w == (float)f) ? gwy[dy] * gwx[dx] : 0.f);
                        gA[0] = md_fma(wgt, A.xy, gA[0]); gA[1] = md_fma(wgt, A.zw, gA[1]);
                        gB[0] = md_fma(wgt, Bc.xy, gB[0]); gB[1] = md_fma(wgt, Bc.zw, gB[1]);
                        gC[0] = md_fma(wgt, Cc.xy, gC[0]); gC[1] = md_fma(wgt, Cc.zw, gC[1]);
                    }
            }
            const v4f xq4 = reinterpret_cast<const v4f *>(wf)[(ty + 2) * B2_W + tx + 2], yq4 = reinterpret_cast<const v4f *>(tg)[(ty + 2) * B2_W + tx + 2];
            const float gl1 = (selq == (unsigned char)(0x80 | f)) ? gscale * maskq * wl1 * (1.f / 3.f) : 0.f;
            v2f dpred[2];   // (R, G), (B, pad)
            bool any = false;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const v2f xq = h ? xq4.zw : xq4.xy, yq = h ? yq4.zw : yq4.xy;
                const v2f diff = yq - xq;
                const v2f sg = {diff.x > 0.f ? 1.f : (diff.x < 0.f ? -1.f : 0.f), diff.y > 0.f ? 1.f : (diff.y < 0.f ? -1.f : 0.f)};
                v2f g = -sg * gl1;
                if (use_ssim) g += div9(gA[h] + 2.f * gB[h] * xq + gC[h] * yq);
                dpred[h] = g;
                any |= g.x != 0.f || (h == 0 && g.y != 0.f);
            }
            // ---- the warp's backward at q (warp.hip): gradients to the depth and to P = (K T)[:3]
            has = any;
            if (any) {
                // A tap beyond the last column / row needs no select (see sample_border): there wx1 / wy1 is exactly 0 and the clip
                // has zeroed the grid gradient of that axis, so the finite value read instead meets a zero factor either way.
                const v4f tv0 = cur.t0, tv1 = cur.t1, tv2 = cur.t2, tv3 = cur.t3;
                const float pzz = cur.pzz, wx1 = cur.wx1, wy1 = cur.wy1, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
                const v2f puv = cur.puv, pg = cur.pg;
                v2f gxy = (v2f)0.f;   // (d / d ix, d / d iy)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const v2f nw = h ? tv0.zw : tv0.xy, ne = h ? tv1.zw : tv1.xy, sw = h ? tv2.zw : tv2.xy, se = h ? tv3.zw : tv3.xy;
                    const v2f tx2 = (ne - nw) * wy0 + (se - sw) * wy1, ty2 = (sw - nw) * wx0 + (se - ne) * wx1;
                    gxy = md_fma((v2f)dpred[h].x, (v2f){tx2.x, ty2.x}, gxy);
                    if (h == 0) gxy = md_fma((v2f)dpred[h].y, (v2f){tx2.y, ty2.y}, gxy);
                }
                const v2f duv = gxy * pg * (2.f / wh1);
                const float rzz = md_rcp_newton(pzz);   // a gradient: the reciprocal within an ulp serves three quotients
                const v2f dc01 = duv * rzz;
                const float dc0 = dc01.x, dc1 = dc01.y, dc2 = -(duv.x * puv.x + duv.y * puv.y) * rzz;
                const v2f a01 = md_fma(cam[f].Pxy[2], (v2f)r2, md_fma(cam[f].Pxy[1], (v2f)r1, cam[f].Pxy[0] * r0));
                const float a2 = cam[f].Pz[0] * r0 + cam[f].Pz[1] * r1 + cam[f].Pz[2] * r2;
                d_depth += dc0 * a01.x + dc1 * a01.y + dc2 * a2;
                if (a.d_T[f]) {
                    const float Xh[4] = {depth * r0, depth * r1, depth * r2, 1.f}, dc[3] = {dc0, dc1, dc2};
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) dP[i * 4 + j] = dc[i] * Xh[j];
                }
            }
        }
        if (a.d_T[f]) {
            // partial sums of dL/dP: float butterflies inside each row of 16 lanes (neighbouring pixels, DPP adds), the 16 row sums
            // of the workgroup through LDS, summed in double by twelve threads (the terms cancel across image regions; as __shfl_xor
            // on doubles this was 288 LDS-crossbar instructions per wave and the kernel LDS-bound, as all-DPP double sums a
            // quarter of its VALU instructions)
            const bool wave_has = __builtin_amdgcn_ballot_w64(has) != 0;   // a wave without a gradient sums twelve zeros: skip
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                float v = dP[k];
#define MD_DPP_ADDF(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false))
                if (wave_has) { MD_DPP_ADDF(0xB1); MD_DPP_ADDF(0x4E); MD_DPP_ADDF(0x141); MD_DPP_ADDF(0x140); }
#undef MD_DPP_ADDF
                if ((lane & 15) == 0) redf[(wave * 4 + (lane >> 4)) * 12 + k] = v;
            }
            if ((MD_PHOTO_BWD_EARLY & 2) && f + 1 < F) nxt = issue_taps(f + 1);   // behind the barriers and the next frame's LDS phase
            __syncthreads();
            if (tid < 12) {
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc += (double)redf[r * 12 + tid];
                wsP[((((size_t)s * F + f) * a.B + b) * nblk + blk) * 12 + tid] = (float)acc;
            }
        }
        else if ((MD_PHOTO_BWD_EARLY & 2) && f + 1 < F) nxt = issue_taps(f + 1);
        __syncthreads();  // cf / red are reused by the next frame
        if ((MD_PHOTO_BWD_EARLY & 2) && f + 1 < F) cur = nxt;
    }
    if (qvalid) {
        if (a.is_disp) {
            const float gv = -d_depth * (max_disp - min_disp) * (depth * depth);   // d depth / d v = -(max-min)/sd^2, depth = 1/sd
            // a level of the image's own size is up-sampled by the identity: its gradient is final here (up_adjoint_kernel skips it)
            if (a.dh[s] == H && a.dw[s] == W) a.d_dz[s][(size_t)b * HW + q] = gv;
            else gup[((size_t)s * a.B + b) * HW + q] = gv;
        }
        else a.d_dz[s][(size_t)b * HW + q] = d_depth;
    }
}

// dL/dT[f][b] = K[b][:3,:]^T  sum_{scales, workgroups} dP   (fixed order, double)
__global__ __launch_bounds__(256) void photo_bwd_finish_kernel(const md_photo_desc a, const float *__restrict__ wsP, int nblk) {
    __shared__ double red[4][12];
    __shared__ double dP[12];
    const int b = blockIdx.x, f = blockIdx.y;
    if (!a.d_T[f]) return;
    double acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.0;
    for (int s = 0; s < a.S; ++s) {
        const float *base = wsP + ((((size_t)s * a.F + f) * a.B + b) * nblk) * 12;
        for (int k = threadIdx.x; k < nblk; k += 256)
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] += (double)base[(size_t)k * 12 + i];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        double v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) dP[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 16) {
        const int k = threadIdx.x / 4, j = threadIdx.x % 4;
        double sum = 0.0;
        for (int i = 0; i < 3; ++i) sum += (double)a.K[b * 16 + i * 4 + k] * dP[i * 4 + j];
        a.d_T[f][b * 16 + threadIdx.x] = (float)sum;
    }
}

// Adjoint of F.interpolate(bilinear, align_corners=False) for every scale in one launch (gather form: deterministic).
// gup [S][B][H][W]: gradient w.r.t. the up-sampled disparity; d_dz[s] [B,1,dh,dw].  lpp lanes per low-resolution pixel scan the
// full-resolution pixels whose two source taps can include it: for an r-fold up-sampling, rows r*i - r/2 - 1 ... r*i + 3r/2
// (the window of (3r+3)^2 used by md_disp_to_depth_up_bwd covers non-integer ratios; here 2r+2 suffices when H % h == 0).
__global__ __launch_bounds__(256) void up_adjoint_kernel(const md_photo_desc a, const float *__restrict__ gup) {
    const int s = blockIdx.y;
    const int h = a.dh[s], w = a.dw[s], H = a.H, W = a.W;
    if (h == H && w == W) return;   // written by photo_bwd_kernel itself
    const int ry = (H + h - 1) / h, rx = (W + w - 1) / w;
    // grid: (lanes of one sample, level, sample); 32-bit index arithmetic, the lanes per pixel a power of two (as 64-bit divisions
    // the three quotients below were several hundred instructions per lane)
    const int ll = ry * rx >= 64 ? 6 : (ry * rx >= 16 ? 4 : (ry * rx >= 4 ? 2 : 0)), lpp = 1 << ll;
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    const unsigned pixl = gid >> ll;
    const int sub = (int)(gid & (unsigned)(lpp - 1));
    const bool live = pixl < (unsigned)(h * w);
    float acc = 0.f;
    const int b = blockIdx.z;
    int iy = 0, ix = 0;
    if (live) {
        iy = (int)(pixl / (unsigned)w);
        ix = (int)(pixl - (unsigned)iy * (unsigned)w);
        const bool exact = H % h == 0 && W % w == 0;
        int oy_lo, oy_hi, ox_lo, ox_hi;
        const bool pow2 = exact && (ry & (ry - 1)) == 0 && (rx & (rx - 1)) == 0;
        if (pow2) {
            // a power-of-two ratio r: the source coordinate (o + 0.5) / r - 0.5 is exact in float, and the outputs with a tap on
            // cell i are exactly o in [r i - r/2, r i + 3r/2): 2r of them per axis, clamped taps at the borders included --
            // 4 window elements per lane at r = 2, 4, 8 where the margin of one on either side made it 9 to 16
            oy_lo = max(0, ry * iy - ry / 2); oy_hi = min(H - 1, ry * iy + (3 * ry) / 2 - 1);
            ox_lo = max(0, rx * ix - rx / 2); ox_hi = min(W - 1, rx * ix + (3 * rx) / 2 - 1);
        } else if (exact) {
            oy_lo = max(0, ry * iy - ry / 2 - 1); oy_hi = min(H - 1, ry * iy + (3 * ry) / 2 + 1);
            ox_lo = max(0, rx * ix - rx / 2 - 1); ox_hi = min(W - 1, rx * ix + (3 * rx) / 2 + 1);
        } else {
            const float fy = (float)H / (float)h, fx = (float)W / (float)w;
            oy_lo = max(0, (int)floorf(((float)iy - 1.f) * fy) - 1); oy_hi = min(H - 1, (int)ceilf(((float)iy + 2.f) * fy) + 1);
            ox_lo = max(0, (int)floorf(((float)ix - 1.f) * fx) - 1); ox_hi = min(W - 1, (int)ceilf(((float)ix + 2.f) * fx) + 1);
        }
        const int fw = ox_hi - ox_lo + 1;
        const float *g = gup + ((size_t)s * a.B + b) * H * W;
        const float sy = (float)h / (float)H, sx = (float)w / (float)W;
        // same size: F.interpolate is the identity (source index = o exactly, weight 1) -- the window loop cost scale 0 nine
        // iterations per pixel for one non-zero term
        const int n = (h == H && w == W) ? 0 : fw * (oy_hi - oy_lo + 1);
        if (n == 0) acc = g[(size_t)iy * W + ix];
        const int side = lpp >= 64 ? 8 : (lpp >= 16 ? 4 : (lpp >= 4 ? 2 : 1));   // the pixel's lanes as a side x side grid
        constexpr int NC = 5;   // window columns per lane the fast path holds weights for
        if (n > 0 && pow2 && side * side == lpp && fw <= 2 * side && oy_hi - oy_lo + 1 <= 2 * side) {
            // r = 2, 4, 8: a lane's share of the 2r x 2r support is 2 x 2 elements -- four unconditional loads from clamped
            // positions, in flight together (as a row loop with a skip per empty row they were two dependent round trips, and the
            // kernel ran at the latency of its 46,000 short waves)
            const int sub_y = sub / side, sub_x = sub % side;
            float wx2[2], wy2[2];
            int ox2[2], oy2[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ox = ox_lo + sub_x + side * k, oy = oy_lo + sub_y + side * k;
                const bool inx = ox <= ox_hi, iny = oy <= oy_hi;
                ox2[k] = inx ? ox : ox_hi; oy2[k] = iny ? oy : oy_hi;
                int i0, i1; float l;
                interp_idx_s(ox2[k], w, sx, i0, i1, l);
                wx2[k] = inx ? (i0 == ix ? 1.f - l : 0.f) + (i1 == ix ? l : 0.f) : 0.f;
                interp_idx_s(oy2[k], h, sy, i0, i1, l);
                wy2[k] = iny ? (i0 == iy ? 1.f - l : 0.f) + (i1 == iy ? l : 0.f) : 0.f;
            }
            const float g00 = g[(size_t)oy2[0] * W + ox2[0]], g01 = g[(size_t)oy2[0] * W + ox2[1]];
            const float g10 = g[(size_t)oy2[1] * W + ox2[0]], g11 = g[(size_t)oy2[1] * W + ox2[1]];
            acc += g00 * wy2[0] * wx2[0]; acc += g01 * wy2[0] * wx2[1];
            acc += g10 * wy2[1] * wx2[0]; acc += g11 * wy2[1] * wx2[1];
        } else if (n > 0 && exact && side * side == lpp && fw <= side * NC) {
            // lane (sub / side, sub % side) takes window rows sub_y + side * j and columns sub_x + side * i: no division by the
            // window width per element (the index arithmetic of the flat loop below was 70 instructions per element: 28 us of the
            // kernel's 42), and the weights are separable: a column's is formed once per lane, a row's once per row -- formed per
            // element they were 15 of an element's 25 instructions
            const int sub_y = sub / side, sub_x = sub % side;
            const int fh = oy_hi - oy_lo + 1;
            float wxk[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int ox = ox_lo + sub_x + side * k;
                int x0, x1; float lx;
                interp_idx_s(ox < W ? ox : W - 1, w, sx, x0, x1, lx);
                wxk[k] = ox < ox_lo + fw ? (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f) : 0.f;
            }
            for (int oy = oy_lo + sub_y; oy < oy_lo + fh; oy += side) {
                int y0, y1; float ly;
                interp_idx_s(oy, h, sy, y0, y1, ly);
                const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
                if (wy == 0.f) continue;   // (uniform over most of the wave: whole window rows fall outside the pixel's support)
                const float *grow = g + (size_t)oy * W + ox_lo + sub_x;
#pragma unroll
                for (int k = 0; k < NC; ++k)
                    if (wxk[k] != 0.f) acc += grow[side * k] * wy * wxk[k];
            }
        } else {
        // (non-integer ratios) no early `continue`: the loads stay unconditional
#pragma unroll 2
        for (int k = sub; k < n; k += lpp) {
            const int oy = oy_lo + k / fw, ox = ox_lo + k % fw;
            const float gv = g[(size_t)oy * W + ox];
            int y0, y1; float ly;
            interp_idx_s(oy, h, sy, y0, y1, ly);
            const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            int x0, x1; float lx;
            interp_idx_s(ox, w, sx, x0, x1, lx);
            const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
            const float wgt = wy * wx;
            if (wgt != 0.f) acc += gv * wy * wx;
        }
        }
    }
    for (int o = 1; o < lpp; o <<= 1) acc += __shfl_xor(acc, o, 64);  // fixed tree: deterministic
    if (live && sub == 0) a.d_dz[s][((size_t)b * h + iy) * w + ix] = acc;
}

// (B,3,H,W) -> (B,H,W,4), up to MAXF + 1 images per launch
struct PackArgs {
    const float *in[MAXF + 1];
    float *out[MAXF + 1];
    int n;
    long long BHW, HW;
};
__global__ __launch_bounds__(256) void pack_rgbx_kernel(const PackArgs a) {
    const int k = blockIdx.y;
    const float *in = a.in[k];
    float4 *out = reinterpret_cast<float4 *>(a.out[k]);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.BHW; i += (long long)gridDim.x * 256) {
        const long long b = i / a.HW, p = i - b * a.HW;
        const float *px = in + b * 3 * a.HW + p;
        out[i] = make_float4(px[0], px[a.HW], px[2 * a.HW], 0.f);
    }
}

int check_desc(const char *fn, const md_photo_desc *d) {
    MD_REQUIRE(d, "%s: null descriptor", fn);
    MD_REQUIRE(d->B > 0 && d->B <= 4096 && d->H >= 3 && d->W >= 3, "%s: bad dims B=%d H=%d W=%d (H, W >= 3)", fn, d->B, d->H, d->W);
    MD_REQUIRE(d->F >= 1 && d->F <= MAXF && d->S >= 1 && d->S <= MAXS, "%s: F=%d (1..%d), S=%d (1..%d)", fn, d->F, MAXF, d->S, MAXS);
    MD_REQUIRE((long long)d->H * d->W < (1ll << 27), "%s: %d x %d: an image of 2^27 pixels or more (32-bit byte offsets into a packed frame)", fn, d->H, d->W);
        MD_REQUIRE(d->target, "%s: null target", fn);
    for (int f = 0; f < d->F; ++f) MD_REQUIRE(d->src[f], "%s: null src[%d]", fn, f);
    if (!d->identity) {
        MD_REQUIRE(d->K && d->invK, "%s: null K / invK", fn);
        for (int f = 0; f < d->F; ++f) MD_REQUIRE(d->T[f], "%s: null T[%d]", fn, f);
        for (int s = 0; s < d->S; ++s) {
            MD_REQUIRE(d->dz[s], "%s: null dz[%d]", fn, s);
            if (d->is_disp) MD_REQUIRE(d->dh[s] > 0 && d->dw[s] > 0, "%s: bad disparity size at scale %d", fn, s);
        }
        MD_REQUIRE(d->min_depth > 0.f && d->max_depth > d->min_depth, "%s: bad depth range", fn);
    }
    return MD_OK;
}

}  // namespace

extern "C" size_t md_photo_desc_bytes(void) { return sizeof(md_photo_desc); }

extern "C" int md_pack_rgbx(const float *const *imgs, int n, int B, int H, int W, float *const *out, md_stream_t stream) {
    MD_REQUIRE(imgs && out && n >= 1 && n <= MAXF + 1, "md_pack_rgbx: 1..%d images", MAXF + 1);
    MD_REQUIRE(B > 0 && H > 0 && W > 0, "md_pack_rgbx: bad dims");
    PackArgs a{};
    for (int k = 0; k < n; ++k) {
        MD_REQUIRE(imgs[k] && out[k], "md_pack_rgbx: null image %d", k);
        a.in[k] = imgs[k];
        a.out[k] = out[k];
    }
    a.n = n; a.HW = (long long)H * W; a.BHW = a.HW * B;
    MD_LAUNCH_TIMED("md_pack_rgbx", pack_rgbx_kernel, dim3((unsigned)md_cdiv(a.BHW, 256 * 2), n), dim3(256), 0, (hipStream_t)stream, a);
    MD_CHECK_LAUNCH("md_pack_rgbx");
    return MD_OK;
}

extern "C" size_t md_photo_fwd_ws_bytes(int B, int S, int H, int W) {
    return sizeof(float) * 2 * (size_t)B * S * md_cdiv(W, FT_W) * md_cdiv(H, FT_H);
}

extern "C" int md_photo_fwd(const md_photo_desc *d, void *ws, md_stream_t stream) {
    int rc = check_desc("md_photo_fwd", d);
    if (rc) return rc;
    const int S = d->identity ? 1 : d->S, F = d->F;
    if (!d->identity) {
        MD_REQUIRE(ws && d->loss, "md_photo_fwd: null workspace / loss");
    } else {
        MD_REQUIRE(d->mn[0], "md_photo_fwd: identity mode writes mn[0]");
    }
    const int tiles = md_cdiv(d->W, FT_W) * md_cdiv(d->H, FT_H);
    dim3 grid(photo_grid(d->B * tiles, S));
    const size_t lds = sizeof(float4) * (size_t)(1 + F) * FP_N;
    hipStream_t st = (hipStream_t)stream;
    const PhotoConsts kc = photo_consts(d);
#define MD_PH_FWD(F_)                                                                                                       \
    do {                                                                                                                    \
        if (d->identity) MD_LAUNCH_TIMED("md_photo_fwd", (photo_fwd_kernel<F_, true>), grid, dim3(256), lds, st, *d, kc, (float *)ws); \
        else MD_LAUNCH_TIMED("md_photo_fwd", (photo_fwd_kernel<F_, false>), grid, dim3(256), lds, st, *d, kc, (float *)ws); \
    } while (0)
    if (F == 1) MD_PH_FWD(1); else if (F == 2) MD_PH_FWD(2); else if (F == 3) MD_PH_FWD(3); else MD_PH_FWD(4);
#undef MD_PH_FWD
    MD_CHECK_LAUNCH("md_photo_fwd");
    if (!d->identity) {
        MD_LAUNCH_TIMED("md_photo_fwd", photo_fwd_finish_kernel, dim3(S), dim3(256), 0, st, (const float *)ws, tiles * d->B, d->loss);
        MD_CHECK_LAUNCH("md_photo_fwd(finish)");
    }
    return MD_OK;
}

extern "C" size_t md_photo_bwd_ws_bytes(int B, int S, int F, int H, int W, int is_disp) {
    const size_t parts = (size_t)12 * S * F * B * md_cdiv(W, BT_W) * md_cdiv(H, BT_H);
    return sizeof(float) * (parts + (is_disp ? (size_t)S * B * H * W : 0));
}

extern "C" int md_photo_bwd(const md_photo_desc *d, void *ws, md_stream_t stream) {
    int rc = check_desc("md_photo_bwd", d);
    if (rc) return rc;
    MD_REQUIRE(!d->identity, "md_photo_bwd: the identity loss has no gradient path (its inputs are the input frames)");
    MD_REQUIRE(ws && d->loss, "md_photo_bwd: null workspace / loss");
    for (int s = 0; s < d->S; ++s) {
        MD_REQUIRE(d->sel[s] && d->d_dz[s], "md_photo_bwd: null sel / d_dz at scale %d", s);
        for (int f = 0; f < d->F; ++f) MD_REQUIRE(d->warped[s][f], "md_photo_bwd: the forward's warped[%d][%d] is required", s, f);
    }
    const int F = d->F;
    const int nblk = md_cdiv(d->W, BT_W) * md_cdiv(d->H, BT_H);
    dim3 grid(photo_grid(d->B * nblk, d->S));
    float *wsP = (float *)ws;
    float *gup = wsP + (size_t)12 * d->S * F * d->B * nblk;
    const size_t lds = sizeof(float4) * ((size_t)(1 + F) * B2_N + 3 * B1_N);
    hipStream_t st = (hipStream_t)stream;
    const PhotoConsts kc = photo_consts(d);
    if (F == 1) MD_LAUNCH_TIMED("md_photo_bwd", (photo_bwd_kernel<1>), grid, dim3(256), lds, st, *d, kc, gup, wsP);
    else if (F == 2) MD_LAUNCH_TIMED("md_photo_bwd", (photo_bwd_kernel<2>), grid, dim3(256), lds, st, *d, kc, gup, wsP);
    else if (F == 3) MD_LAUNCH_TIMED("md_photo_bwd", (photo_bwd_kernel<3>), grid, dim3(256), lds, st, *d, kc, gup, wsP);
    else MD_LAUNCH_TIMED("md_photo_bwd", (photo_bwd_kernel<4>), grid, dim3(256), lds, st, *d, kc, gup, wsP);
    MD_CHECK_LAUNCH("md_photo_bwd");
    bool any_T = false;
    for (int f = 0; f < F; ++f) any_T |= d->d_T[f] != nullptr;
    if (any_T) {
        MD_LAUNCH_TIMED("md_photo_bwd", photo_bwd_finish_kernel, dim3(d->B, F), dim3(256), 0, st, *d, (const float *)wsP, nblk);
        MD_CHECK_LAUNCH("md_photo_bwd(finish)");
    }
    if (d->is_disp) {
        long long mx = 0;
        for (int s = 0; s < d->S; ++s) {
            if (d->dh[s] == d->H && d->dw[s] == d->W) continue;   // photo_bwd_kernel wrote d_dz[s] itself
            int lpp = md_cdiv(d->H, d->dh[s]) * md_cdiv(d->W, d->dw[s]);
            lpp = lpp >= 64 ? 64 : (lpp >= 16 ? 16 : (lpp >= 4 ? 4 : 1));
            const long long n = (long long)d->dh[s] * d->dw[s] * lpp;   // lanes of one sample
            mx = n > mx ? n : mx;
        }
        if (mx > 0) {
            MD_REQUIRE(mx < (1ll << 31), "md_photo_bwd: a disparity level of %lld lanes per sample", mx);
            MD_LAUNCH_TIMED("md_photo_bwd", up_adjoint_kernel, dim3((unsigned)md_cdiv(mx, 256), d->S, d->B), dim3(256), 0, st, *d, (const float *)gup);
            MD_CHECK_LAUNCH("md_photo_bwd(up-sampling adjoint)");
        }
    }
    return MD_OK;
}
