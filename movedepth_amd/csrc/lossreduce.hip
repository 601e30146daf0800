// Loss reductions of the photometric path:
//   * min over frames + auto-mask + masked mean (reference trainer.py:687-709 mono, 630-662 MVS,
//     compute_loss_masks 552-567),
//   * edge-aware smoothness on the mean-normalised disparity (layers.py:630-643, trainer.py:712-714).
// All reductions are two-stage and deterministic (per-block partials in the caller's workspace, then a
// single-block finish); no float atomics.  Launch-latency-bound on (B,1,H,W) maps.
#include "md_common.hpp"

namespace {

__device__ __forceinline__ float block_sum(float v, float *red /*[4]*/) {
    v = md_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------ masked min
__global__ __launch_bounds__(256) void masked_min_kernel(const float *__restrict__ reproj, const float *__restrict__ ident,
                                                         const float *__restrict__ noise, const float *__restrict__ ext,
                                                         int N, int HW, int total, int mvs_mode,
                                                         float *__restrict__ mn, float *__restrict__ mask,
                                                         float *__restrict__ ws) {
    __shared__ float red[4];
    float num = 0.f, den = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int b = i / HW, p = i % HW;
        const float *rp = reproj + (size_t)b * N * HW + p;
        float r = rp[0];
        for (int f = 1; f < N; ++f) r = fminf(r, rp[(size_t)f * HW]);
        float m = 1.f;
        if (ident && !mvs_mode) {
            const float *ip = ident + (size_t)b * N * HW + p;
            float id = ip[0];
            for (int f = 1; f < N; ++f) id = fminf(id, ip[(size_t)f * HW]);
            if (noise) id += noise[i];
            m = (r <= id) ? 1.f : 0.f;  // argmin([reproj, identity]) == 0, first index wins ties
        }
        if (ext) m *= ext[i];
        mn[i] = r;
        mask[i] = m;
        num += r * m;
        den += m;
    }
    num = block_sum(num, red);
    den = block_sum(den, red);
    if (threadIdx.x == 0) { ws[blockIdx.x * 2] = num; ws[blockIdx.x * 2 + 1] = den; }
}

__global__ __launch_bounds__(256) void masked_min_finish_kernel(const float *__restrict__ ws, int nblk, float *__restrict__ loss) {
    __shared__ float red[4];
    float num = 0.f, den = 0.f;
    for (int k = threadIdx.x; k < nblk; k += 256) { num += ws[k * 2]; den += ws[k * 2 + 1]; }
    num = block_sum(num, red);
    den = block_sum(den, red);
    if (threadIdx.x == 0) { loss[0] = num / (den + 1e-7f); loss[1] = den; }
}

__global__ __launch_bounds__(256) void masked_min_bwd_kernel(const float *__restrict__ gloss, const float *__restrict__ reproj,
                                                             const float *__restrict__ mask, const float *__restrict__ loss,
                                                             int N, int HW, int total, float *__restrict__ d_reproj) {
    const float scale = gloss[0] / (loss[1] + 1e-7f);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int b = i / HW, p = i % HW;
        const float *rp = reproj + (size_t)b * N * HW + p;
        int am = 0;
        float r = rp[0];
        for (int f = 1; f < N; ++f) {
            const float v = rp[(size_t)f * HW];
            if (v < r) { r = v; am = f; }
        }
        const float g = scale * mask[i];
        for (int f = 0; f < N; ++f) d_reproj[((size_t)b * N + f) * HW + p] = f == am ? g : 0.f;
    }
}

int mm_blocks(int total) {
    int n = md_cdiv(total, 256 * 4);
    return n < 1 ? 1 : (n > 1024 ? 1024 : n);
}

// ------------------------------------------------------------------ smoothness
// blocks per sample (and level) for the pixel passes.  256: with 64 the full-resolution level -- 122,880 pixels per sample, four
// exp() edge weights per pixel in the backward -- ran on 384 workgroups (45 us for the four levels' backward)
constexpr int SM_BLK = 256;
constexpr int SM_MAXS = MD_PHOTO_MAX_SCALES;

// All pyramid levels of one compute_losses call in one launch per pass (grid z = level): 3 + 3 launches per step instead of
// 12 + 12 of 4-16 us kernels over 0.03-1.5 MB maps.
struct SmoothArgs {
    const float *disp[SM_MAXS], *img[SM_MAXS], *gloss[SM_MAXS];
    float *d_disp[SM_MAXS];
    int h[SM_MAXS], w[SM_MAXS];
    int B, Ci, normalize, S;
};
// workspace layout per level (floats): (unused)[B] | part[B*SM_BLK*2] | dots[B*SM_BLK] | msum[B*SM_BLK]
__host__ __device__ inline size_t sm_ws_floats(int B) { return (size_t)B + (size_t)B * SM_BLK * 4; }

// Mean of each sample's disparity, stage 1: SM_BLK partial sums per sample.  (One block per sample, as this was at
// first, is 480 dependent loads per thread at 192x640: 186 us for a 0.5 MB reduction.)  Stage 2 is sample_mean() at the
// top of each consumer, always the same fixed-order sum, so forward and backward see the identical mean.
__global__ __launch_bounds__(256) void smooth_mean_kernel(const SmoothArgs a, float *__restrict__ wsall) {
    __shared__ float red[4];
    const int b = blockIdx.y, lv = blockIdx.z, B = a.B, hw = a.h[lv] * a.w[lv];
    float *ws = wsall + lv * sm_ws_floats(B);
    const float *disp = a.disp[lv];
    // A block without a pixel (the coarse levels fill 8 to 120 of the SM_BLK blocks a sample gets) leaves its zero and goes: run in
    // full -- the mean's 256 loads, three block sums -- the empty blocks were 60 % of the launch's workgroups and most of its time.
    if (blockIdx.x * 256 >= hw) {
        if (threadIdx.x == 0) ws[B + (size_t)B * SM_BLK * 3 + (size_t)b * SM_BLK + blockIdx.x] = 0.f;
        return;
    }
    float s = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < hw; p += SM_BLK * 256) s += disp[(size_t)b * hw + p];
    s = block_sum(s, red);
    if (threadIdx.x == 0) ws[B + (size_t)B * SM_BLK * 3 + (size_t)b * SM_BLK + blockIdx.x] = s;
}

__device__ __forceinline__ float sample_mean(const float *__restrict__ ws, int B, int b, int hw, float *red) {
    const float s = threadIdx.x < SM_BLK ? ws[B + (size_t)B * SM_BLK * 3 + (size_t)b * SM_BLK + threadIdx.x] : 0.f;
    return block_sum(s, red) / (float)hw;
}

__device__ __forceinline__ float edge_w(const float *__restrict__ img, int Ci, size_t hw, size_t p, size_t q) {
    float gi = 0.f;
    for (int c = 0; c < Ci; ++c) gi += fabsf(img[c * hw + p] - img[c * hw + q]);
    return expf(-gi / (float)Ci);
}

__global__ __launch_bounds__(256) void smooth_fwd_kernel(const SmoothArgs a, float *__restrict__ wsall) {
    __shared__ float red[4];
    const int b = blockIdx.y, lv = blockIdx.z, B = a.B, Ci = a.Ci, h = a.h[lv], w = a.w[lv], hw = h * w;
    float *ws = wsall + lv * sm_ws_floats(B);
    if (blockIdx.x * 256 >= hw) {   // no pixel: see smooth_mean_kernel
        if (threadIdx.x == 0) { float *part = ws + B + ((size_t)b * SM_BLK + blockIdx.x) * 2; part[0] = 0.f; part[1] = 0.f; }
        return;
    }
    const float dn = a.normalize ? sample_mean(ws, B, b, hw, red) + 1e-7f : 1.f;
    const float *d = a.disp[lv] + (size_t)b * hw, *im = a.img[lv] + (size_t)b * Ci * hw;
    float sx = 0.f, sy = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < hw; p += SM_BLK * 256) {
        const int x = p % w, y = p / w;
        const float v = d[p] / dn;
        if (x + 1 < w) sx += fabsf(v - d[p + 1] / dn) * edge_w(im, Ci, hw, p, p + 1);
        if (y + 1 < h) sy += fabsf(v - d[p + w] / dn) * edge_w(im, Ci, hw, p, p + w);
    }
    sx = block_sum(sx, red);
    sy = block_sum(sy, red);
    if (threadIdx.x == 0) {
        float *part = ws + B + ((size_t)b * SM_BLK + blockIdx.x) * 2;
        part[0] = sx; part[1] = sy;
    }
}

__global__ __launch_bounds__(256) void smooth_finish_kernel(const SmoothArgs a, const float *__restrict__ wsall, float *__restrict__ loss) {
    __shared__ float red[4];
    const int lv = blockIdx.x, B = a.B, h = a.h[lv], w = a.w[lv];
    const float *ws = wsall + lv * sm_ws_floats(B);
    float sx = 0.f, sy = 0.f;
    for (int k = threadIdx.x; k < B * SM_BLK; k += 256) { sx += ws[B + k * 2]; sy += ws[B + k * 2 + 1]; }
    sx = block_sum(sx, red);
    sy = block_sum(sy, red);
    if (threadIdx.x == 0) loss[lv] = sx / ((float)B * h * (w - 1)) + sy / ((float)B * (h - 1) * w);
}

// pass 1 of the backward: gn = dL/d(normalised disp) (gather form) into d_disp, and per-block dot(gn, disp)
__global__ __launch_bounds__(256) void smooth_bwd_kernel(const SmoothArgs a, float *__restrict__ wsall) {
    __shared__ float red[4];
    const int b = blockIdx.y, lv = blockIdx.z, B = a.B, Ci = a.Ci, h = a.h[lv], w = a.w[lv], hw = h * w;
    float *ws = wsall + lv * sm_ws_floats(B);
    if (blockIdx.x * 256 >= hw) {   // no pixel: see smooth_mean_kernel
        if (threadIdx.x == 0) ws[B + (size_t)B * SM_BLK * 2 + (size_t)b * SM_BLK + blockIdx.x] = 0.f;
        return;
    }
    const float dn = a.normalize ? sample_mean(ws, B, b, hw, red) + 1e-7f : 1.f;
    const float gl = a.gloss[lv] ? a.gloss[lv][0] : 0.f;
    const float cx = gl / ((float)B * h * (w - 1)), cy = gl / ((float)B * (h - 1) * w);
    const float *d = a.disp[lv] + (size_t)b * hw, *im = a.img[lv] + (size_t)b * Ci * hw;
    float *d_disp = a.d_disp[lv];
    float dot = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < hw; p += SM_BLK * 256) {
        const int x = p % w, y = p / w;
        const float v = d[p] / dn;
        float g = 0.f;
        if (x + 1 < w) { const float df = v - d[p + 1] / dn; g += (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * edge_w(im, Ci, hw, p, p + 1) * cx; }
        if (x > 0)     { const float df = d[p - 1] / dn - v; g -= (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * edge_w(im, Ci, hw, p - 1, p) * cx; }
        if (y + 1 < h) { const float df = v - d[p + w] / dn; g += (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * edge_w(im, Ci, hw, p, p + w) * cy; }
        if (y > 0)     { const float df = d[p - w] / dn - v; g -= (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * edge_w(im, Ci, hw, p - w, p) * cy; }
        d_disp[(size_t)b * hw + p] = g;
        dot += g * d[p];
    }
    dot = block_sum(dot, red);
    if (threadIdx.x == 0) ws[B + (size_t)B * SM_BLK * 2 + (size_t)b * SM_BLK + blockIdx.x] = dot;
}

// pass 2: nd = d / (mean + 1e-7)  =>  d_d[q] = gn[q]/dn - dot/(dn^2 hw)
__global__ __launch_bounds__(256) void smooth_bwd_finish_kernel(const SmoothArgs a, const float *__restrict__ wsall) {
    __shared__ float red[4];
    const int b = blockIdx.y, lv = blockIdx.z, B = a.B, hw = a.h[lv] * a.w[lv];
    const float *ws = wsall + lv * sm_ws_floats(B);
    float *d_disp = a.d_disp[lv];
    if (blockIdx.x * 256 >= hw) return;   // no pixel
    const float dn = sample_mean(ws, B, b, hw, red) + 1e-7f;
    const float dot = block_sum(threadIdx.x < SM_BLK ? ws[B + (size_t)B * SM_BLK * 2 + (size_t)b * SM_BLK + threadIdx.x] : 0.f, red);
    const float corr = dot / (dn * dn) / (float)hw;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < hw; p += SM_BLK * 256)
        d_disp[(size_t)b * hw + p] = d_disp[(size_t)b * hw + p] / dn - corr;
}

int smooth_check(const char *fn, const SmoothArgs &a) {
    MD_REQUIRE(a.S >= 1 && a.S <= SM_MAXS, "%s: %d levels (1..%d)", fn, a.S, SM_MAXS);
    MD_REQUIRE(a.B > 0 && a.B <= 65535 && a.Ci > 0, "%s: bad dims", fn);
    for (int s = 0; s < a.S; ++s) {
        MD_REQUIRE(a.disp[s] && a.img[s], "%s: null tensor at level %d", fn, s);
        MD_REQUIRE(a.h[s] > 1 && a.w[s] > 1, "%s: level %d is %dx%d (needs h, w > 1)", fn, s, a.h[s], a.w[s]);
    }
    return MD_OK;
}

int smooth_fwd_launch(const SmoothArgs &a, float *loss, void *ws, hipStream_t st) {
    const char *TN_ = a.S > 1 ? "md_smooth_multi_fwd" : "md_smooth_fwd";
    if (a.normalize) {
        MD_LAUNCH_TIMED(TN_, smooth_mean_kernel, dim3(SM_BLK, a.B, a.S), dim3(256), 0, st, a, (float *)ws);
        MD_CHECK_LAUNCH("md_smooth_fwd(mean)");
    }
    MD_LAUNCH_TIMED(TN_, smooth_fwd_kernel, dim3(SM_BLK, a.B, a.S), dim3(256), 0, st, a, (float *)ws);
    MD_CHECK_LAUNCH("md_smooth_fwd");
    MD_LAUNCH_TIMED(TN_, smooth_finish_kernel, dim3(a.S), dim3(256), 0, st, a, (const float *)ws, loss);
    MD_CHECK_LAUNCH("md_smooth_fwd(finish)");
    return MD_OK;
}

int smooth_bwd_launch(const SmoothArgs &a, void *ws, hipStream_t st) {
    const char *TN_ = a.S > 1 ? "md_smooth_multi_bwd" : "md_smooth_bwd";
    if (a.normalize) {  // recompute the means: the workspace need not survive between forward and backward
        MD_LAUNCH_TIMED(TN_, smooth_mean_kernel, dim3(SM_BLK, a.B, a.S), dim3(256), 0, st, a, (float *)ws);
        MD_CHECK_LAUNCH("md_smooth_bwd(mean)");
    }
    MD_LAUNCH_TIMED(TN_, smooth_bwd_kernel, dim3(SM_BLK, a.B, a.S), dim3(256), 0, st, a, (float *)ws);
    MD_CHECK_LAUNCH("md_smooth_bwd");
    if (a.normalize) {
        MD_LAUNCH_TIMED(TN_, smooth_bwd_finish_kernel, dim3(SM_BLK, a.B, a.S), dim3(256), 0, st, a, (const float *)ws);
        MD_CHECK_LAUNCH("md_smooth_bwd(finish)");
    }
    return MD_OK;
}

}  // namespace

extern "C" size_t md_masked_min_ws_bytes(int B, int H, int W) { return sizeof(float) * 2 * (size_t)mm_blocks(B * H * W); }

extern "C" int md_masked_min_fwd(const float *reproj, const float *ident, const float *noise, const float *ext_mask,
                                 int B, int N, int H, int W, int mvs_mode, float *min_reproj, float *mask, float *loss,
                                 void *ws, md_stream_t stream) {
    MD_REQUIRE(reproj && min_reproj && mask && loss && ws, "md_masked_min_fwd: null tensor");
    MD_REQUIRE(B > 0 && N > 0 && H > 0 && W > 0, "md_masked_min_fwd: bad dims");
    const int total = B * H * W, nblk = mm_blocks(total);
    hipLaunchKernelGGL(masked_min_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, reproj, ident, noise, ext_mask, N,
                       H * W, total, mvs_mode, min_reproj, mask, (float *)ws);
    MD_CHECK_LAUNCH("md_masked_min_fwd");
    hipLaunchKernelGGL(masked_min_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float *)ws, nblk, loss);
    MD_CHECK_LAUNCH("md_masked_min_fwd(finish)");
    return MD_OK;
}

extern "C" int md_masked_min_bwd(const float *gloss, const float *reproj, const float *mask, const float *loss, int B,
                                 int N, int H, int W, float *d_reproj, md_stream_t stream) {
    MD_REQUIRE(gloss && reproj && mask && loss && d_reproj, "md_masked_min_bwd: null tensor");
    MD_REQUIRE(B > 0 && N > 0 && H > 0 && W > 0, "md_masked_min_bwd: bad dims");
    const int total = B * H * W;
    hipLaunchKernelGGL(masked_min_bwd_kernel, dim3(mm_blocks(total)), dim3(256), 0, (hipStream_t)stream, gloss, reproj, mask,
                       loss, N, H * W, total, d_reproj);
    MD_CHECK_LAUNCH("md_masked_min_bwd");
    return MD_OK;
}

extern "C" size_t md_smooth_ws_bytes(int B, int h, int w) {
    (void)h; (void)w;
    return sizeof(float) * sm_ws_floats(B);
}

extern "C" int md_smooth_fwd(const float *disp, const float *img, int B, int Ci, int h, int w, int normalize, float *loss,
                             void *ws, md_stream_t stream) {
    MD_REQUIRE(disp && img && loss && ws, "md_smooth_fwd: null tensor");
    SmoothArgs a = {};
    a.disp[0] = disp; a.img[0] = img; a.h[0] = h; a.w[0] = w; a.B = B; a.Ci = Ci; a.normalize = normalize; a.S = 1;
    int rc = smooth_check("md_smooth_fwd", a);
    if (rc) return rc;
    return smooth_fwd_launch(a, loss, ws, (hipStream_t)stream);
}

extern "C" int md_smooth_bwd(const float *gloss, const float *disp, const float *img, int B, int Ci, int h, int w,
                             int normalize, float *d_disp, void *ws, md_stream_t stream) {
    MD_REQUIRE(gloss && disp && img && d_disp && ws, "md_smooth_bwd: null tensor");
    SmoothArgs a = {};
    a.disp[0] = disp; a.img[0] = img; a.gloss[0] = gloss; a.d_disp[0] = d_disp; a.h[0] = h; a.w[0] = w;
    a.B = B; a.Ci = Ci; a.normalize = normalize; a.S = 1;
    int rc = smooth_check("md_smooth_bwd", a);
    if (rc) return rc;
    return smooth_bwd_launch(a, ws, (hipStream_t)stream);
}

extern "C" size_t md_smooth_multi_ws_bytes(int B, int S) { return sizeof(float) * sm_ws_floats(B) * (size_t)(S > 0 ? S : 1); }

extern "C" int md_smooth_multi_fwd(const float *const *disp, const float *const *img, const int *h, const int *w, int S, int B,
                                   int Ci, int normalize, float *loss, void *ws, md_stream_t stream) {
    MD_REQUIRE(disp && img && h && w && loss && ws, "md_smooth_multi_fwd: null argument");
    MD_REQUIRE(S >= 1 && S <= SM_MAXS, "md_smooth_multi_fwd: %d levels (1..%d)", S, SM_MAXS);
    SmoothArgs a = {};
    for (int s = 0; s < S; ++s) { a.disp[s] = disp[s]; a.img[s] = img[s]; a.h[s] = h[s]; a.w[s] = w[s]; }
    a.B = B; a.Ci = Ci; a.normalize = normalize; a.S = S;
    int rc = smooth_check("md_smooth_multi_fwd", a);
    if (rc) return rc;
    return smooth_fwd_launch(a, loss, ws, (hipStream_t)stream);
}

extern "C" int md_smooth_multi_bwd(const float *const *gloss, const float *const *disp, const float *const *img, const int *h,
                                   const int *w, int S, int B, int Ci, int normalize, float *const *d_disp, void *ws,
                                   md_stream_t stream) {
    MD_REQUIRE(gloss && disp && img && h && w && d_disp && ws, "md_smooth_multi_bwd: null argument");
    MD_REQUIRE(S >= 1 && S <= SM_MAXS, "md_smooth_multi_bwd: %d levels (1..%d)", S, SM_MAXS);
    SmoothArgs a = {};
    for (int s = 0; s < S; ++s) {
        MD_REQUIRE(d_disp[s], "md_smooth_multi_bwd: null d_disp[%d]", s);
        a.disp[s] = disp[s]; a.img[s] = img[s]; a.gloss[s] = gloss[s]; a.d_disp[s] = d_disp[s]; a.h[s] = h[s]; a.w[s] = w[s];
    }
    a.B = B; a.Ci = Ci; a.normalize = normalize; a.S = S;
    int rc = smooth_check("md_smooth_multi_bwd", a);
    if (rc) return rc;
    return smooth_bwd_launch(a, ws, (hipStream_t)stream);
}
