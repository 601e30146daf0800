// Post-volume depth regression ("next" row 8f-1): softmax over D (reference trainer.py:367), entropy
// (layers.py:862-863) and localmax (layers.py:796-812) in one pass over the (B,D,h,w) logits -- the reference
// makes 4-5 separate streaming passes (softmax, clamp/log/mul/sum, argmax, 3 gathers).
// One thread per (sample, pixel); accesses are coalesced along the pixel axis.  localmax's endpoint quirk
// (index d decodes to hypothesis D-1-d) lives in the caller's min_inv/max_inv arguments.
#include "md_common.hpp"

namespace {

struct Soft {
    float mx, den;
    int am;
};

__device__ __forceinline__ Soft softmax_stats(const float *lp, int D, size_t hw) {
    Soft s;
    s.mx = -INFINITY; s.am = 0;
    for (int d = 0; d < D; ++d) {
        const float v = lp[d * hw];
        if (v > s.mx) { s.mx = v; s.am = d; }  // first maximum, like torch.argmax on ties of the probabilities
    }
    s.den = 0.f;
    for (int d = 0; d < D; ++d) s.den += expf(lp[d * hw] - s.mx);
    return s;
}

__global__ __launch_bounds__(256) void sel_fwd_kernel(const float *__restrict__ logits, int D, int hw, int radius,
                                                      const float *__restrict__ min_inv, const float *__restrict__ max_inv,
                                                      float *__restrict__ prob, float *__restrict__ entropy,
                                                      float *__restrict__ depth) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const size_t shw = (size_t)hw;
    const float *lp = logits + (size_t)b * D * shw + p;
    const Soft s = softmax_stats(lp, D, shw);
    float ent = 0.f;
    for (int d = 0; d < D; ++d) {
        const float v = expf(lp[d * shw] - s.mx) / s.den;
        if (prob) prob[(size_t)b * D * shw + d * shw + p] = v;
        ent += -v * logf(fminf(fmaxf(v, 1e-9f), 1.f));
    }
    if (entropy) entropy[(size_t)b * shw + p] = ent;
    float num = 0.f, den = 1e-6f;
    for (int i = -radius; i <= radius; ++i) {
        const int idx = min(max(s.am + i, 0), D - 1);  // clamped: border bins are counted twice
        const float v = expf(lp[idx * shw] - s.mx) / s.den;
        num += (float)idx * v;
        den += v;
    }
    const float nrm = (num / den) / (float)(D - 1);
    const float a = min_inv[(size_t)b * shw + p], bb = max_inv[(size_t)b * shw + p];
    depth[(size_t)b * shw + p] = 1.f / (a + nrm * (bb - a));
}

__global__ __launch_bounds__(256) void sel_bwd_kernel(const float *__restrict__ g_depth, const float *__restrict__ g_entropy,
                                                      const float *__restrict__ logits, int D, int hw, int radius,
                                                      const float *__restrict__ min_inv, const float *__restrict__ max_inv,
                                                      float *__restrict__ d_logits) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const size_t shw = (size_t)hw;
    const float *lp = logits + (size_t)b * D * shw + p;
    float *dl = d_logits + (size_t)b * D * shw + p;
    const Soft s = softmax_stats(lp, D, shw);
    // localmax: r = num/den, depth = 1/(a + r/(D-1) (b-a))
    float num = 0.f, den = 1e-6f;
    for (int i = -radius; i <= radius; ++i) {
        const int idx = min(max(s.am + i, 0), D - 1);
        const float v = expf(lp[idx * shw] - s.mx) / s.den;
        num += (float)idx * v;
        den += v;
    }
    const float r = num / den;
    const float a = min_inv[(size_t)b * shw + p], bb = max_inv[(size_t)b * shw + p];
    const float dep = 1.f / (a + (r / (float)(D - 1)) * (bb - a));
    const float gd = g_depth ? g_depth[(size_t)b * shw + p] : 0.f;
    const float g_r = -gd * dep * dep * (bb - a) / (float)(D - 1);
    const float ge = g_entropy ? g_entropy[(size_t)b * shw + p] : 0.f;
    // gp_d = dL/dprob_d; softmax backward: d_logit_d = p_d (gp_d - sum_k gp_k p_k)
    const int lo = s.am - radius, hi = s.am + radius;
    float dot = 0.f;
    for (int d = 0; d < D; ++d) {
        const float v = expf(lp[d * shw] - s.mx) / s.den;
        float gp = 0.f;
        if (ge != 0.f) gp += ge * ((v < 1e-9f) ? -logf(1e-9f) : (v > 1.f ? 0.f : -logf(v) - 1.f));
        // multiplicity of bin d among the clamped window indices
        int mult = (d >= lo && d <= hi) ? 1 : 0;
        if (d == 0 && lo < 0) mult += -lo;
        if (d == D - 1 && hi > D - 1) mult += hi - (D - 1);
        if (mult) gp += g_r * (float)mult * ((float)d - r) / den;
        dot += gp * v;
        dl[d * shw] = gp;  // stash gp, finished below
    }
    for (int d = 0; d < D; ++d) {
        const float v = expf(lp[d * shw] - s.mx) / s.den;
        dl[d * shw] = v * (dl[d * shw] - dot);
    }
}

// ---- D split over four threads per pixel (D <= 128) --------------------------------------------------------------
// One thread per pixel walks D bins three to four times with dependent loads and, at 48x160, gives 180 workgroups for
// 256 CUs: 74 / 129 us forward / backward for 18-53 MB of traffic.  Here a workgroup is 64 pixels x 4 contiguous
// quarters of D (wave = quarter, so each load is 64 consecutive pixels of one bin); a thread keeps its <= 32 logits
// in registers for all passes (one global read) and the quarters meet through LDS.  Ties of the maximum resolve to the
// first bin as before: quarters are scanned in order with a strict comparison.
constexpr int SEL_NB = 32;  // bins per thread, upper bound

struct SelShared {
    float f[4][4][64];  // [slot][quarter][pixel]
    int am[4][64];
};

__device__ __forceinline__ int sel_mult(int d, int D, int lo, int hi) {  // multiplicity of bin d among the clamped window
    int m = (d >= lo && d <= hi) ? 1 : 0;
    if (d == 0 && lo < 0) m += -lo;
    if (d == D - 1 && hi > D - 1) m += hi - (D - 1);
    return m;
}

template <bool BWD>
__global__ __launch_bounds__(256) void sel4_kernel(const float *__restrict__ g_depth, const float *__restrict__ g_entropy,
                                                   const float *__restrict__ logits, int D, int hw, int radius,
                                                   const float *__restrict__ min_inv, const float *__restrict__ max_inv,
                                                   float *__restrict__ prob, float *__restrict__ entropy,
                                                   float *__restrict__ depth, float *__restrict__ d_logits) {
    __shared__ SelShared sh;
    const int b = blockIdx.y, pl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + pl;
    const bool valid = p < hw;
    const size_t shw = (size_t)hw;
    const int nb = (D + 3) / 4, dbeg = q * nb;
    const float *lp = logits + (size_t)b * D * shw + (valid ? p : 0);
    float v[SEL_NB];
    float mx = -INFINITY;
    int am = 0;
#pragma unroll
    for (int i = 0; i < SEL_NB; ++i) {
        const int d = dbeg + i;
        v[i] = lp[min(d, D - 1) * shw];   // unconditional (clamped) so that all the loads are in flight together
    }
#pragma unroll
    for (int i = 0; i < SEL_NB; ++i) {
        const int d = dbeg + i;
        if (!(i < nb && d < D)) v[i] = -INFINITY;
        if (v[i] > mx) { mx = v[i]; am = d; }
    }
    sh.f[0][q][pl] = mx;
    sh.am[q][pl] = am;
    __syncthreads();
    mx = sh.f[0][0][pl]; am = sh.am[0][pl];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (sh.f[0][k][pl] > mx) { mx = sh.f[0][k][pl]; am = sh.am[k][pl]; }
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < SEL_NB; ++i) {
        v[i] = (i < nb && dbeg + i < D) ? expf(v[i] - mx) : 0.f;
        den += v[i];
    }
    sh.f[1][q][pl] = den;
    __syncthreads();
    den = (sh.f[1][0][pl] + sh.f[1][1][pl]) + (sh.f[1][2][pl] + sh.f[1][3][pl]);
    const int lo = am - radius, hi = am + radius;
    float num = 0.f, wden = 0.f, ent = 0.f;
#pragma unroll
    for (int i = 0; i < SEL_NB; ++i) {
        const int d = dbeg + i;
        if (i < nb && d < D) {
            v[i] = v[i] / den;  // probability
            const int m = sel_mult(d, D, lo, hi);
            num += (float)(m * d) * v[i];
            wden += (float)m * v[i];
            if (!BWD) {
                if (prob && valid) prob[(size_t)b * D * shw + d * shw + p] = v[i];
                ent += -v[i] * logf(fminf(fmaxf(v[i], 1e-9f), 1.f));
            }
        }
    }
    sh.f[2][q][pl] = num; sh.f[3][q][pl] = wden;
    if (!BWD) sh.f[0][q][pl] = ent;  // slot 0 is free again: every thread passed the second barrier after reading it
    __syncthreads();
    num = (sh.f[2][0][pl] + sh.f[2][1][pl]) + (sh.f[2][2][pl] + sh.f[2][3][pl]);
    wden = 1e-6f + ((sh.f[3][0][pl] + sh.f[3][1][pl]) + (sh.f[3][2][pl] + sh.f[3][3][pl]));
    const float r = num / wden;
    const float a = valid ? min_inv[(size_t)b * shw + p] : 1.f, bb = valid ? max_inv[(size_t)b * shw + p] : 1.f;
    const float dep = 1.f / (a + (r / (float)(D - 1)) * (bb - a));
    if (!BWD) {
        if (q == 0 && valid) {
            depth[(size_t)b * shw + p] = dep;
            if (entropy) entropy[(size_t)b * shw + p] = (sh.f[0][0][pl] + sh.f[0][1][pl]) + (sh.f[0][2][pl] + sh.f[0][3][pl]);
        }
        return;
    }
    const float gd = (g_depth && valid) ? g_depth[(size_t)b * shw + p] : 0.f;
    const float g_r = -gd * dep * dep * (bb - a) / (float)(D - 1);
    const float ge = (g_entropy && valid) ? g_entropy[(size_t)b * shw + p] : 0.f;
    float gp[SEL_NB];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < SEL_NB; ++i) {
        const int d = dbeg + i;
        gp[i] = 0.f;
        if (i < nb && d < D) {
            const float pv = v[i];
            if (ge != 0.f) gp[i] += ge * ((pv < 1e-9f) ? -logf(1e-9f) : (pv > 1.f ? 0.f : -logf(pv) - 1.f));
            const int m = sel_mult(d, D, lo, hi);
            if (m) gp[i] += g_r * (float)m * ((float)d - r) / wden;
            dot += gp[i] * pv;
        }
    }
    sh.f[0][q][pl] = dot;
    __syncthreads();
    dot = (sh.f[0][0][pl] + sh.f[0][1][pl]) + (sh.f[0][2][pl] + sh.f[0][3][pl]);
    float *dl = d_logits + (size_t)b * D * shw + p;
#pragma unroll
    for (int i = 0; i < SEL_NB; ++i) {
        const int d = dbeg + i;
        if (i < nb && d < D && valid) dl[d * shw] = v[i] * (gp[i] - dot);
    }
}

}  // namespace

extern "C" int md_softmax_entropy_localmax_fwd(const float *logits, int B, int D, int h, int w, int radius,
                                               const float *min_inv, const float *max_inv, float *prob, float *entropy,
                                               float *depth, md_stream_t stream) {
    MD_REQUIRE(logits && min_inv && max_inv && depth, "md_softmax_entropy_localmax_fwd: null tensor");
    MD_REQUIRE(B > 0 && B <= 65535 && D > 1 && h > 0 && w > 0 && radius >= 0, "md_softmax_entropy_localmax_fwd: bad dims");
    if (D <= 4 * SEL_NB)
        hipLaunchKernelGGL(sel4_kernel<false>, dim3(md_cdiv(h * w, 64), B), dim3(256), 0, (hipStream_t)stream, nullptr, nullptr,
                           logits, D, h * w, radius, min_inv, max_inv, prob, entropy, depth, nullptr);
    else
        hipLaunchKernelGGL(sel_fwd_kernel, dim3(md_cdiv(h * w, 256), B), dim3(256), 0, (hipStream_t)stream, logits, D, h * w,
                           radius, min_inv, max_inv, prob, entropy, depth);
    MD_CHECK_LAUNCH("md_softmax_entropy_localmax_fwd");
    return MD_OK;
}

extern "C" int md_softmax_entropy_localmax_bwd(const float *g_depth, const float *g_entropy, const float *logits, int B,
                                               int D, int h, int w, int radius, const float *min_inv,
                                               const float *max_inv, float *d_logits, md_stream_t stream) {
    MD_REQUIRE(logits && min_inv && max_inv && d_logits, "md_softmax_entropy_localmax_bwd: null tensor");
    MD_REQUIRE(B > 0 && B <= 65535 && D > 1 && h > 0 && w > 0 && radius >= 0, "md_softmax_entropy_localmax_bwd: bad dims");
    if (D <= 4 * SEL_NB)
        hipLaunchKernelGGL(sel4_kernel<true>, dim3(md_cdiv(h * w, 64), B), dim3(256), 0, (hipStream_t)stream, g_depth, g_entropy,
                           logits, D, h * w, radius, min_inv, max_inv, nullptr, nullptr, nullptr, d_logits);
    else
        hipLaunchKernelGGL(sel_bwd_kernel, dim3(md_cdiv(h * w, 256), B), dim3(256), 0, (hipStream_t)stream, g_depth, g_entropy,
                           logits, D, h * w, radius, min_inv, max_inv, d_logits);
    MD_CHECK_LAUNCH("md_softmax_entropy_localmax_bwd");
    return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// Convex upsampling (reference layers.py:200-214): depth (B,h,w), mask (B, 9*s*s, h, w) with s = 2**scale ->
// out (B, s*h, s*w): per fine pixel a softmax over 9 mask logits weights the zero-padded 3x3 coarse neighbourhood.
// The reference materialises softmax(mask) (B,9,s,s,h,w), an unfold and a permuted product; here one thread per
// fine pixel does it in registers (forward), and the backward is the same walk (d_mask per thread, d_depth as a
// gather over the s*s x 9 fine pixels that read a coarse pixel: deterministic, no atomics).
namespace {

__device__ __forceinline__ void cu_softmax9(const float *__restrict__ mask, size_t base, size_t kstride, float (&p)[9]) {
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) { p[k] = mask[base + k * kstride]; mx = fmaxf(mx, p[k]); }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { p[k] = expf(p[k] - mx); den += p[k]; }
#pragma unroll
    for (int k = 0; k < 9; ++k) p[k] /= den;
}

__global__ __launch_bounds__(256) void convex_up_fwd_kernel(const float *__restrict__ depth, const float *__restrict__ mask,
                                                            int h, int w, int s, float *__restrict__ out) {
    const int b = blockIdx.z, H = h * s, W = w * s;
    const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= W || Y >= H) return;
    const int x = X / s, j = X % s, y = Y / s, i = Y % s;
    const size_t hw = (size_t)h * w;
    // mask.view(B, 9, s, s, h, w): channel = (k*s + i)*s + j
    float p[9];
    cu_softmax9(mask, ((size_t)b * 9 * s * s + (size_t)i * s + j) * hw + (size_t)y * w + x, (size_t)s * s * hw, p);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) acc += p[k] * depth[(size_t)b * hw + (size_t)yy * w + xx];
    }
    out[((size_t)b * H + Y) * W + X] = acc;
}

// Backward, both gradients from one pass over the fine pixels.  A thread owns fine pixel (Y, X) of coarse cell (y, x) =
// (Y / s, X / s): soft-max backward for d_mask, and its nine terms g * p[k] of the depth gradient -- tap k of every fine pixel of
// cell (y, x) is coarse pixel (y + k/3 - 1, x + k%3 - 1).  The terms are summed over the cell's s x s fine pixels here (lanes of a
// cell: shuffles; rows of a cell: LDS, fixed order) into cellsum[b][k][y][x]; convex_up_bwd_depth_kernel then gathers nine of them
// per coarse pixel.  (Round 2 gathered straight from gout / mask: 9 x s^2 = 144 soft-max evaluations per coarse pixel on 46,080
// threads, 185 us per step -- the longest kernel of the post-volume path.)
// Block: 64 fine columns x max(4, s) fine rows = whole cells (s a power of two <= 16).
__global__ __launch_bounds__(256) void convex_up_bwd_mask_kernel(const float *__restrict__ gout, const float *__restrict__ depth,
                                                                 const float *__restrict__ mask, int h, int w, int s,
                                                                 float *__restrict__ d_mask, float *__restrict__ cellsum) {
    __shared__ float red[4][9][64];
    const int b = blockIdx.z, H = h * s, W = w * s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows = s > 4 ? s : 4;                 // fine rows of this block
    const int X = blockIdx.x * 64 + lane, Y0 = blockIdx.y * rows;
    const size_t hw = (size_t)h * w, kstride = (size_t)s * s * hw;
    const int x = X / s, j = X % s;
    float t[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k] = 0.f;
    for (int r = wave; r < rows; r += 4) {          // s <= 4: one row per wave; s = 8, 16: the rows of one cell row, 4 at a time
        const int Y = Y0 + r;
        if (X >= W || Y >= H) continue;
        const int y = Y / s, i = Y % s;
        const size_t base = ((size_t)b * 9 * s * s + (size_t)i * s + j) * hw + (size_t)y * w + x;
        float p[9], v[9];
        cu_softmax9(mask, base, kstride, p);
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            v[k] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? depth[(size_t)b * hw + (size_t)yy * w + xx] : 0.f;
            dot += p[k] * v[k];
        }
        const float g = gout[((size_t)b * H + Y) * W + X];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            d_mask[base + k * kstride] = g * p[k] * (v[k] - dot);  // softmax backward
            t[k] += g * p[k];
        }
    }
    // sum over the s lanes (fine columns) of a cell, then over the waves (fine rows) that share a cell row
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float a = t[k];
        for (int o = 1; o < s; o <<= 1) a += __shfl_xor(a, o, 64);
        red[wave][k][lane] = a;
    }
    __syncthreads();
    const int wpc = s >= 4 ? 4 : s;                 // waves per cell row (s = 1: each wave is its own cell row)
    const int cells_x = 64 / s;                     // cells per wave along x
    const int crows = 4 / wpc;                      // cell rows per block (s >= 4: 1)
    for (int idx = threadIdx.x; idx < 9 * cells_x * crows; idx += 256) {
        const int cx = idx % cells_x, k = (idx / cells_x) % 9, cr = idx / (cells_x * 9);
        const int xc = blockIdx.x * cells_x + cx, yc = (s >= 4 ? blockIdx.y : blockIdx.y * crows + cr);
        if (xc >= w || yc >= h) continue;
        float a = 0.f;
        for (int wv = 0; wv < wpc; ++wv) a += red[cr * wpc + wv][k][cx * s];
        cellsum[((size_t)b * 9 + k) * hw + (size_t)yc * w + xc] = a;
    }
}

__global__ __launch_bounds__(256) void convex_up_bwd_depth_kernel(const float *__restrict__ cellsum, int h, int w,
                                                                  float *__restrict__ d_depth) {
    const int b = blockIdx.z;
    const int xx = blockIdx.x * 64 + (threadIdx.x & 63), yy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xx >= w || yy >= h) return;
    const size_t hw = (size_t)h * w;
    float acc = 0.f;
    // coarse pixel (yy,xx) is tap k of the fine pixels of coarse cell (y,x) = (yy - k/3 + 1, xx - k%3 + 1)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int y = yy - (k / 3 - 1), x = xx - (k % 3 - 1);
        if (y >= 0 && y < h && x >= 0 && x < w) acc += cellsum[((size_t)b * 9 + k) * hw + (size_t)y * w + x];
    }
    d_depth[(size_t)b * hw + (size_t)yy * w + xx] = acc;
}

// Standalone geometry (reference layers.py:556-621) for call compatibility; the hot kernels fuse these.
__global__ __launch_bounds__(256) void backproject_kernel(const float *__restrict__ depth, const float *__restrict__ invK,
                                                          int nk, int h, int w, float *__restrict__ cam) {
    const int b = blockIdx.z, hw = h * w, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const float *iK = invK + (nk == 1 ? 0 : b) * 16;
    const float x = (float)(p % w), y = (float)(p / w), d = depth[(size_t)b * hw + p];
    float *c = cam + (size_t)b * 4 * hw;
    c[p] = d * (iK[0] * x + iK[1] * y + iK[2]);
    c[hw + p] = d * (iK[4] * x + iK[5] * y + iK[6]);
    c[2 * hw + p] = d * (iK[8] * x + iK[9] * y + iK[10]);
    c[3 * hw + p] = 1.f;
}

__global__ __launch_bounds__(256) void project3d_kernel(const float *__restrict__ pts, const float *__restrict__ K,
                                                        const float *__restrict__ T, int nk, int h, int w, float eps,
                                                        float *__restrict__ pix) {
    const int b = blockIdx.z, hw = h * w, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const float *Kb = K + (nk == 1 ? 0 : b) * 16, *Tb = T + (nk == 1 ? 0 : b) * 16;
    float P[12];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sacc = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) sacc += Kb[i * 4 + k] * Tb[k * 4 + j];
            P[i * 4 + j] = sacc;
        }
    const float *c = pts + (size_t)b * 4 * hw;
    const float X = c[p], Y = c[hw + p], Z = c[2 * hw + p], Wh = c[3 * hw + p];
    const float c0 = P[0] * X + P[1] * Y + P[2] * Z + P[3] * Wh, c1 = P[4] * X + P[5] * Y + P[6] * Z + P[7] * Wh,
                c2 = P[8] * X + P[9] * Y + P[10] * Z + P[11] * Wh;
    const float zz = c2 + eps;
    pix[((size_t)b * hw + p) * 2] = (c0 / zz / (float)(w - 1) - 0.5f) * 2.f;
    pix[((size_t)b * hw + p) * 2 + 1] = (c1 / zz / (float)(h - 1) - 0.5f) * 2.f;
}

}  // namespace

extern "C" int md_convex_upsample_fwd(const float *depth, const float *mask, int B, int h, int w, int scale, float *out,
                                      md_stream_t stream) {
    MD_REQUIRE(depth && mask && out, "md_convex_upsample_fwd: null tensor");
    MD_REQUIRE(B > 0 && B <= 65535 && h > 0 && w > 0 && scale >= 0 && scale <= 4, "md_convex_upsample_fwd: bad dims");
    const int s = 1 << scale;
    hipLaunchKernelGGL(convex_up_fwd_kernel, dim3(md_cdiv(w * s, 64), md_cdiv(h * s, 4), B), dim3(256), 0, (hipStream_t)stream,
                       depth, mask, h, w, s, out);
    MD_CHECK_LAUNCH("md_convex_upsample_fwd");
    return MD_OK;
}

extern "C" size_t md_convex_upsample_bwd_ws_bytes(int B, int h, int w) { return sizeof(float) * 9 * (size_t)B * h * w; }

extern "C" int md_convex_upsample_bwd(const float *gout, const float *depth, const float *mask, int B, int h, int w,
                                      int scale, float *d_depth, float *d_mask, void *ws, md_stream_t stream) {
    MD_REQUIRE(gout && depth && mask && d_depth && d_mask && ws, "md_convex_upsample_bwd: null tensor");
    MD_REQUIRE(B > 0 && B <= 65535 && h > 0 && w > 0 && scale >= 0 && scale <= 4, "md_convex_upsample_bwd: bad dims");
    const int s = 1 << scale, rows = s > 4 ? s : 4;
    hipLaunchKernelGGL(convex_up_bwd_mask_kernel, dim3(md_cdiv(w * s, 64), md_cdiv(h * s, rows), B), dim3(256), 0,
                       (hipStream_t)stream, gout, depth, mask, h, w, s, d_mask, (float *)ws);
    MD_CHECK_LAUNCH("md_convex_upsample_bwd(mask)");
    hipLaunchKernelGGL(convex_up_bwd_depth_kernel, dim3(md_cdiv(w, 64), md_cdiv(h, 4), B), dim3(256), 0, (hipStream_t)stream,
                       (const float *)ws, h, w, d_depth);
    MD_CHECK_LAUNCH("md_convex_upsample_bwd(depth)");
    return MD_OK;
}

extern "C" int md_backproject(const float *depth, const float *invK, int Bs, int nk, int h, int w, float *cam_points,
                              md_stream_t stream) {
    MD_REQUIRE(depth && invK && cam_points, "md_backproject: null tensor");
    MD_REQUIRE(Bs > 0 && Bs <= 65535 && (nk == 1 || nk == Bs) && h > 0 && w > 0, "md_backproject: bad dims");
    hipLaunchKernelGGL(backproject_kernel, dim3(md_cdiv(h * w, 256), 1, Bs), dim3(256), 0, (hipStream_t)stream, depth, invK, nk,
                       h, w, cam_points);
    MD_CHECK_LAUNCH("md_backproject");
    return MD_OK;
}

extern "C" int md_project3d(const float *points, const float *K, const float *T, int Bs, int nk, int h, int w, float eps,
                            float *pix, md_stream_t stream) {
    MD_REQUIRE(points && K && T && pix, "md_project3d: null tensor");
    MD_REQUIRE(Bs > 0 && Bs <= 65535 && (nk == 1 || nk == Bs) && h > 1 && w > 1, "md_project3d: bad dims");
    hipLaunchKernelGGL(project3d_kernel, dim3(md_cdiv(h * w, 256), 1, Bs), dim3(256), 0, (hipStream_t)stream, points, K, T, nk,
                       h, w, eps, pix);
    MD_CHECK_LAUNCH("md_project3d");
    return MD_OK;
}
