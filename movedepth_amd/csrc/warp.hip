// Photometric reprojection warp (reference trainer.py:501-507, 519-529, 575-580):
//   BackprojectDepth[0] (layers.py:581-586) -> Project3D[0] (layers.py:608-620)
//   -> F.grid_sample(img, pix, padding_mode='border', align_corners=True)
// fused into one kernel (the reference runs ~12 launches and materialises (B,4,HW) points), and its autograd
// w.r.t. depth and the 4x4 pose T.  Also the disparity-pyramid -> full-res depth step
// (F.interpolate bilinear align_corners=False, trainer.py:512 + disp_to_depth, layers.py:400-409).
//
// Images are 4.4 MB at 192x640xB6: L2-resident, launch-latency-bound.  One thread per target pixel, lanes
// along x so the source taps of a wave are neighbouring columns (coalesced gathers).
#include "md_photo.hpp"

namespace {

using namespace mdp;

__global__ __launch_bounds__(256) void warp_fwd_kernel(const float *__restrict__ img, const float *__restrict__ depth,
                                                       const float *__restrict__ K, const float *__restrict__ invK,
                                                       const float *__restrict__ T, int Ci, int H, int W,
                                                       float *__restrict__ pix, float *__restrict__ out,
                                                       unsigned char *__restrict__ oob) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const CamMats cam = md_load_cam_plain(K + b * 16, invK + b * 16, T + b * 16);
    const size_t HW = (size_t)H * W, p = (size_t)y * W + x;
    float r0, r1, r2;
    md_ray(cam, (float)x, (float)y, r0, r1, r2);
    const Proj pr = md_project(cam, r0, r1, r2, depth[b * HW + p], W, H);
    if (pix) { pix[(b * HW + p) * 2] = pr.gx; pix[(b * HW + p) * 2 + 1] = pr.gy; }
    if (oob) oob[b * HW + p] = (pr.gx < -1.f || pr.gx > 1.f || pr.gy < -1.f || pr.gy > 1.f) ? 1 : 0;
    const Clip c = clip_border(pr.ix, pr.iy, W, H);
    const Tap t = md_make_tap(c.ix, c.iy, W, H);
    const int x1 = t.x0 + 1, y1 = t.y0 + 1;
    const bool vx1 = x1 < W, vy1 = y1 < H;  // x0,y0 are in range after clipping
    const float wx0 = 1.f - t.wx1, wy0 = 1.f - t.wy1;
    for (int ch = 0; ch < Ci; ++ch) {
        // grid_sample's interpolation in the reference's order (nw product, then fused multiply-adds of ne, sw, se: the order
        // the fixtures pin, md_common.hpp / oracle tap_sample): the warped frame is bit-equal to the oracle's and the reference's
#pragma clang fp contract(off)
        const float *im = img + ((size_t)b * Ci + ch) * HW;
        float o = im[t.y0 * W + t.x0] * (wy0 * wx0);
        if (vx1) o = fmaf(im[t.y0 * W + x1], wy0 * t.wx1, o);
        if (vy1) o = fmaf(im[y1 * W + t.x0], t.wy1 * wx0, o);
        if (vx1 && vy1) o = fmaf(im[y1 * W + x1], t.wy1 * t.wx1, o);
        out[((size_t)b * Ci + ch) * HW + p] = o;
    }
}

// Backward: per pixel d_depth, plus 12 partial sums of dL/dP per block (wave shuffles -> LDS -> ws).
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float *__restrict__ gout, const float *__restrict__ img,
                                                       const float *__restrict__ depth, const float *__restrict__ K,
                                                       const float *__restrict__ invK, const float *__restrict__ T,
                                                       int Ci, int H, int W, float *__restrict__ d_depth,
                                                       double *__restrict__ ws) {
    __shared__ double red[4][12];
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const bool valid = x < W && y < H;
    const CamMats cam = md_load_cam_plain(K + b * 16, invK + b * 16, T + b * 16);
    const size_t HW = (size_t)H * W, p = (size_t)y * W + x;
    float dP[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) dP[i] = 0.f;
    if (valid) {
        float r0, r1, r2;
        md_ray(cam, (float)x, (float)y, r0, r1, r2);
        const Proj pr = md_project(cam, r0, r1, r2, depth[b * HW + p], W, H);
        const Clip c = clip_border(pr.ix, pr.iy, W, H);
        const Tap t = md_make_tap(c.ix, c.iy, W, H);
        const int x1 = t.x0 + 1, y1 = t.y0 + 1;
        const bool vx1 = x1 < W, vy1 = y1 < H;
        const float wx0 = 1.f - t.wx1, wy0 = 1.f - t.wy1;
        float gix = 0.f, giy = 0.f;
        for (int ch = 0; ch < Ci; ++ch) {
            const float *im = img + ((size_t)b * Ci + ch) * HW;
            const float g = gout[((size_t)b * Ci + ch) * HW + p];
            const float nw = im[t.y0 * W + t.x0];
            const float ne = vx1 ? im[t.y0 * W + x1] : 0.f;
            const float sw = vy1 ? im[y1 * W + t.x0] : 0.f;
            const float se = (vx1 && vy1) ? im[y1 * W + x1] : 0.f;
            gix += g * ((ne - nw) * wy0 + (se - sw) * t.wy1);
            giy += g * ((sw - nw) * wx0 + (se - ne) * t.wx1);
        }
        const float du = gix * c.gmx * (2.f / (float)(W - 1));
        const float dv = giy * c.gmy * (2.f / (float)(H - 1));
        const float dc0 = du / pr.zz, dc1 = dv / pr.zz, dc2 = -(du * pr.u + dv * pr.v) / pr.zz;
        const float a0 = cam.P[0] * r0 + cam.P[1] * r1 + cam.P[2] * r2;
        const float a1 = cam.P[4] * r0 + cam.P[5] * r1 + cam.P[6] * r2;
        const float a2 = cam.P[8] * r0 + cam.P[9] * r1 + cam.P[10] * r2;
        d_depth[b * HW + p] = dc0 * a0 + dc1 * a1 + dc2 * a2;
        const float Xh[4] = {pr.X, pr.Y, pr.Z, 1.f}, dc[3] = {dc0, dc1, dc2};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) dP[i * 4 + j] = dc[i] * Xh[j];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // float inside 16-lane rows (neighbouring pixels), double above: the terms cancel across image regions (see photo.hip)
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const double s = md_wave_sum_dpp_f16_d(dP[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const int blk = blockIdx.y * gridDim.x + blockIdx.x, nblk = gridDim.x * gridDim.y;
        ws[((size_t)b * nblk + blk) * 12 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

// Deterministic second stage: sum the per-block dP partials of a sample, then dT = K[:3,:]^T dP.
// The partials cancel heavily (sum |terms| >> |sum|), so this stage accumulates in fp64.
__global__ __launch_bounds__(256) void warp_bwd_finish_kernel(const double *__restrict__ ws, const float *__restrict__ K,
                                                              int nblk, float *__restrict__ d_T) {
    __shared__ double red[4][12];
    __shared__ double dP[12];
    const int b = blockIdx.x;
    double acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 256)
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] += ws[((size_t)b * nblk + k) * 12 + i];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        double s = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 12) dP[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 16) {
        const int k = threadIdx.x / 4, j = threadIdx.x % 4;
        double s = 0.0;
        for (int i = 0; i < 3; ++i) s += (double)K[b * 16 + i * 4 + k] * dP[i * 4 + j];
        d_T[b * 16 + threadIdx.x] = (float)s;
    }
}

// ---- disparity pyramid level -> full-resolution depth
__global__ __launch_bounds__(256) void disp_up_fwd_kernel(const float *__restrict__ disp, int h, int w, int H, int W,
                                                          float min_disp, float max_disp, float *__restrict__ depth) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    int x0, x1, y0, y1;
    float lx, ly;
    interp_idx(x, w, W, x0, x1, lx);
    interp_idx(y, h, H, y0, y1, ly);
    const float *s = disp + (size_t)b * h * w;
    {   // the oracle's operations one by one (mdo_resize_bilinear_fwd, mdo_disp_to_depth)
#pragma clang fp contract(off)
        const float v = interp4(s[y0 * w + x0], s[y0 * w + x1], s[y1 * w + x0], s[y1 * w + x1], lx, ly);
        const float sd = min_disp + (max_disp - min_disp) * v;
        depth[((size_t)b * H + y) * W + x] = 1.f / sd;
    }
}

// Gather form of the adjoint (deterministic, no atomics): LPP lanes per low-res pixel scan the full-res pixels whose
// bilinear footprint can touch it (a (3r+3)^2 window for an r-fold upsampling: 729 pixels at r = 8, where one thread per
// low-res pixel meant 11,520 threads looping 729 times: 41 us per call against 5 us for the forward) and meet by shuffles.
template <int LPP>
__global__ __launch_bounds__(256) void disp_up_bwd_kernel(const float *__restrict__ g_depth, const float *__restrict__ disp,
                                                          int B, int h, int w, int H, int W, float min_disp, float max_disp,
                                                          float *__restrict__ d_disp) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long pix = gid / LPP;
    const int sub = (int)(gid % LPP);
    const bool live = pix < (long long)B * h * w;
    float acc = 0.f;
    int b = 0, iy = 0, ix = 0;
    if (live) {
        b = (int)(pix / ((long long)h * w));
        const int rem = (int)(pix - (long long)b * h * w);
        iy = rem / w;
        ix = rem - iy * w;
        const float ry = (float)H / (float)h, rx = (float)W / (float)w;
        const int oy_lo = max(0, (int)floorf(((float)iy - 1.f) * ry) - 1), oy_hi = min(H - 1, (int)ceilf(((float)iy + 2.f) * ry) + 1);
        const int ox_lo = max(0, (int)floorf(((float)ix - 1.f) * rx) - 1), ox_hi = min(W - 1, (int)ceilf(((float)ix + 2.f) * rx) + 1);
        const int fw = ox_hi - ox_lo + 1, n = fw * (oy_hi - oy_lo + 1);
        const float *s = disp + (size_t)b * h * w;
        for (int k = sub; k < n; k += LPP) {
            const int oy = oy_lo + k / fw, ox = ox_lo + k % fw;
            int y0, y1; float ly;
            interp_idx(oy, h, H, y0, y1, ly);
            const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            if (wy == 0.f) continue;
            int x0, x1; float lx;
            interp_idx(ox, w, W, x0, x1, lx);
            const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
            if (wx == 0.f) continue;
            // recompute the forward value at (oy, ox): depth = 1/sd, d depth / d v = -(max-min) / sd^2
            const float v = interp4(s[y0 * w + x0], s[y0 * w + x1], s[y1 * w + x0], s[y1 * w + x1], lx, ly);
            const float sd = min_disp + (max_disp - min_disp) * v;
            const float gv = -g_depth[((size_t)b * H + oy) * W + ox] * (max_disp - min_disp) / (sd * sd);
            acc += gv * wy * wx;
        }
    }
#pragma unroll
    for (int o = 1; o < LPP; o <<= 1) acc += __shfl_xor(acc, o, 64);  // fixed tree: deterministic
    if (live && sub == 0) d_disp[((size_t)b * h + iy) * w + ix] = acc;
}

int check_img(const char *fn, int B, int Ci, int H, int W) {
    MD_REQUIRE(B > 0 && B <= 65535 && Ci > 0 && H > 1 && W > 1, "%s: bad dims B=%d Ci=%d H=%d W=%d", fn, B, Ci, H, W);
    return MD_OK;
}

}  // namespace

extern "C" int md_warp_fwd(const float *img, const float *depth, const float *K, const float *invK, const float *T,
                           int B, int Ci, int H, int W, float *pix, float *out, unsigned char *oob_mask,
                           md_stream_t stream) {
    int rc = check_img("md_warp_fwd", B, Ci, H, W);
    if (rc) return rc;
    MD_REQUIRE(img && depth && K && invK && T && out, "md_warp_fwd: null tensor");
    dim3 grid(md_cdiv(W, 64), md_cdiv(H, 4), B);
    hipLaunchKernelGGL(warp_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, depth, K, invK, T, Ci, H, W, pix,
                       out, oob_mask);
    MD_CHECK_LAUNCH("md_warp_fwd");
    return MD_OK;
}

extern "C" size_t md_warp_bwd_ws_bytes(int B, int H, int W) {
    return sizeof(double) * 12 * (size_t)B * md_cdiv(W, 64) * md_cdiv(H, 4);
}

extern "C" int md_warp_bwd(const float *gout, const float *img, const float *depth, const float *K, const float *invK,
                           const float *T, int B, int Ci, int H, int W, float *d_depth, float *d_T, void *ws,
                           md_stream_t stream) {
    int rc = check_img("md_warp_bwd", B, Ci, H, W);
    if (rc) return rc;
    MD_REQUIRE(gout && img && depth && K && invK && T && d_depth && d_T && ws, "md_warp_bwd: null tensor");
    dim3 grid(md_cdiv(W, 64), md_cdiv(H, 4), B);
    hipLaunchKernelGGL(warp_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, gout, img, depth, K, invK, T, Ci, H, W,
                       d_depth, (double *)ws);
    MD_CHECK_LAUNCH("md_warp_bwd");
    hipLaunchKernelGGL(warp_bwd_finish_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const double *)ws, K,
                       (int)(grid.x * grid.y), d_T);
    MD_CHECK_LAUNCH("md_warp_bwd(finish)");
    return MD_OK;
}

extern "C" int md_disp_to_depth_up_fwd(const float *disp, int B, int h, int w, int H, int W, float min_depth,
                                       float max_depth, float *depth, md_stream_t stream) {
    MD_REQUIRE(disp && depth, "md_disp_to_depth_up_fwd: null tensor");
    MD_REQUIRE(B > 0 && B <= 65535 && h > 0 && w > 0 && H > 0 && W > 0, "md_disp_to_depth_up_fwd: bad dims");
    hipLaunchKernelGGL(disp_up_fwd_kernel, dim3(md_cdiv(W, 64), md_cdiv(H, 4), B), dim3(256), 0, (hipStream_t)stream,
                       disp, h, w, H, W, 1.f / max_depth, 1.f / min_depth, depth);
    MD_CHECK_LAUNCH("md_disp_to_depth_up_fwd");
    return MD_OK;
}

extern "C" int md_disp_to_depth_up_bwd(const float *g_depth, const float *disp, int B, int h, int w, int H, int W,
                                       float min_depth, float max_depth, float *d_disp, md_stream_t stream) {
    MD_REQUIRE(g_depth && disp && d_disp, "md_disp_to_depth_up_bwd: null tensor");
    MD_REQUIRE(B > 0 && B <= 65535 && h > 0 && w > 0 && H > 0 && W > 0, "md_disp_to_depth_up_bwd: bad dims");
    const long long npix = (long long)B * h * w;
    const int r = md_cdiv(H, h) > md_cdiv(W, w) ? md_cdiv(H, h) : md_cdiv(W, w);  // upsampling factor
    const float lo = 1.f / max_depth, hi = 1.f / min_depth;
    hipStream_t st = (hipStream_t)stream;
    if (r >= 8) hipLaunchKernelGGL(disp_up_bwd_kernel<64>, dim3((unsigned)md_cdiv(npix * 64, 256)), dim3(256), 0, st, g_depth, disp, B, h, w, H, W, lo, hi, d_disp);
    else if (r >= 4) hipLaunchKernelGGL(disp_up_bwd_kernel<16>, dim3((unsigned)md_cdiv(npix * 16, 256)), dim3(256), 0, st, g_depth, disp, B, h, w, H, W, lo, hi, d_disp);
    else if (r >= 2) hipLaunchKernelGGL(disp_up_bwd_kernel<4>, dim3((unsigned)md_cdiv(npix * 4, 256)), dim3(256), 0, st, g_depth, disp, B, h, w, H, W, lo, hi, d_disp);
    else hipLaunchKernelGGL(disp_up_bwd_kernel<1>, dim3((unsigned)md_cdiv(npix, 256)), dim3(256), 0, st, g_depth, disp, B, h, w, H, W, lo, hi, d_disp);
    MD_CHECK_LAUNCH("md_disp_to_depth_up_bwd");
    return MD_OK;
}
