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

}  // namespace

extern "C" int md_softmax_entropy_localmax_fwd(const float *logits, int B, int D, int h, int w, int radius,
                                               const float *min_inv, const float *max_inv, float *prob, float *entropy,
                                               float *depth, md_stream_t stream) {
    MD_REQUIRE(logits && min_inv && max_inv && depth, "md_softmax_entropy_localmax_fwd: null tensor");
    MD_REQUIRE(B > 0 && B <= 65535 && D > 1 && h > 0 && w > 0 && radius >= 0, "md_softmax_entropy_localmax_fwd: bad dims");
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
    hipLaunchKernelGGL(sel_bwd_kernel, dim3(md_cdiv(h * w, 256), B), dim3(256), 0, (hipStream_t)stream, g_depth, g_entropy,
                       logits, D, h * w, radius, min_inv, max_inv, d_logits);
    MD_CHECK_LAUNCH("md_softmax_entropy_localmax_bwd");
    return MD_OK;
}
