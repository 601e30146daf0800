// Confidence-weighted fusion of per-frame grouped cost volumes (reference trainer.py:349-363):
//   m_f[g] = mean_D vol_f[:,g];  w_f = max_g softmax_g(m_f) = 1 / sum_g exp(m_f[g] - max m_f)
//   out = sum_f w_f vol_f / (1e-8 + sum_f w_f)
// One thread per (sample, pixel); every access is coalesced along the pixel axis.  HBM-bound streaming:
// forward = 2 reads + 1 write of each volume.  With ONE lookup frame (BASELINE configs 1-4) the result is
// the input to 1.6e-7 relative (w >= 1/G), and the host skips this kernel (movedepth_amd/ops.py).
#include "md_common.hpp"

namespace {

constexpr int MAXF = 4;

struct FuseArgs {
    const float *vol[MAXF];
    float *dvol[MAXF];
    const float *gout;
    float *out, *weights;
    int N, B, D, G, hw;
    int eval_mode;  // 0: training weight (mean over D, soft-max over G); 1: evaluation weight (mean over G, soft-max over D)
    long long sb, sd, sg, sp;
};

// online max / sum-exp over the G group means of one frame at one pixel
__device__ __forceinline__ void frame_weight(const float *v, const FuseArgs &a, float &M, float &s, int &am) {
    M = -INFINITY; s = 0.f; am = 0;
    for (int g = 0; g < a.G; ++g) {
        float acc = 0.f;
        for (int d = 0; d < a.D; ++d) acc += v[(size_t)d * a.sd + (size_t)g * a.sg];
        const float m = acc / (float)a.D;
        if (m > M) { s = s * expf(M - m) + 1.f; M = m; am = g; }
        else s += expf(m - M);
    }
}

// the evaluation script's weight (evaluate_depth.py:236): cv.mean(2) is the mean over the G groups, the soft-max runs over D
__device__ __forceinline__ void frame_weight_eval(const float *v, const FuseArgs &a, float &M, float &s) {
    M = -INFINITY; s = 0.f;
    for (int d = 0; d < a.D; ++d) {
        float acc = 0.f;
        for (int g = 0; g < a.G; ++g) acc += v[(size_t)d * a.sd + (size_t)g * a.sg];
        const float m = acc / (float)a.G;
        if (m > M) { s = s * expf(M - m) + 1.f; M = m; }
        else s += expf(m - M);
    }
}

__global__ __launch_bounds__(256) void fuse_fwd_kernel(FuseArgs a) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.hw) return;
    float wf[MAXF], wsum = 1e-8f;
#pragma unroll
    for (int f = 0; f < MAXF; ++f) {
        wf[f] = 0.f;
        if (f < a.N) {
            float M, s; int am;
            if (a.eval_mode) frame_weight_eval(a.vol[f] + (size_t)b * a.sb + (size_t)p * a.sp, a, M, s);
            else frame_weight(a.vol[f] + (size_t)b * a.sb + (size_t)p * a.sp, a, M, s, am);
            wf[f] = 1.f / s;
            wsum += wf[f];
            if (a.weights) a.weights[((size_t)f * a.B + b) * a.hw + p] = wf[f];
        }
    }
    for (int d = 0; d < a.D; ++d)
        for (int g = 0; g < a.G; ++g) {
            const size_t o = (size_t)b * a.sb + (size_t)d * a.sd + (size_t)g * a.sg + (size_t)p * a.sp;
            float acc = 0.f;
#pragma unroll
            for (int f = 0; f < MAXF; ++f)
                if (f < a.N) acc += wf[f] * a.vol[f][o];
            a.out[o] = acc / wsum;
        }
}

__global__ __launch_bounds__(256) void fuse_bwd_kernel(FuseArgs a) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.hw) return;
    float wf[MAXF], Mf[MAXF], sf[MAXF], dw[MAXF], wsum = 1e-8f;
    int am[MAXF];
#pragma unroll
    for (int f = 0; f < MAXF; ++f) {
        wf[f] = 0.f; dw[f] = 0.f; Mf[f] = 0.f; sf[f] = 1.f; am[f] = 0;
        if (f < a.N) {
            frame_weight(a.vol[f] + (size_t)b * a.sb + (size_t)p * a.sp, a, Mf[f], sf[f], am[f]);
            wf[f] = 1.f / sf[f];
            wsum += wf[f];
        }
    }
    // dL/dw_f = sum_{d,g} gout * (vol_f - cor) / wsum
    for (int d = 0; d < a.D; ++d)
        for (int g = 0; g < a.G; ++g) {
            const size_t o = (size_t)b * a.sb + (size_t)d * a.sd + (size_t)g * a.sg + (size_t)p * a.sp;
            float v[MAXF], acc = 0.f;
#pragma unroll
            for (int f = 0; f < MAXF; ++f) {
                v[f] = f < a.N ? a.vol[f][o] : 0.f;
                acc += wf[f] * v[f];
            }
            const float go = a.gout[o], cor = acc / wsum;
#pragma unroll
            for (int f = 0; f < MAXF; ++f) dw[f] += go * (v[f] - cor) / wsum;
        }
#pragma unroll
    for (int f = 0; f < MAXF; ++f) {
        if (f >= a.N) continue;
        const float *v = a.vol[f] + (size_t)b * a.sb + (size_t)p * a.sp;
        float *dv = a.dvol[f] + (size_t)b * a.sb + (size_t)p * a.sp;
        const float pstar = wf[f];  // softmax probability of the arg-max group
        for (int g = 0; g < a.G; ++g) {
            float acc = 0.f;
            for (int d = 0; d < a.D; ++d) acc += v[(size_t)d * a.sd + (size_t)g * a.sg];
            const float pg = expf(acc / (float)a.D - Mf[f]) / sf[f];
            const float dm = pstar * ((g == am[f] ? 1.f : 0.f) - pg);
            const float add = dw[f] * dm / (float)a.D;
            for (int d = 0; d < a.D; ++d) {
                const size_t o = (size_t)d * a.sd + (size_t)g * a.sg;
                dv[o] = a.gout[(size_t)b * a.sb + o + (size_t)p * a.sp] * wf[f] / wsum + add;
            }
        }
    }
}

int fill(FuseArgs &a, const char *fn, const float *const *vols, int N, int B, int D, int G, int hw, long long sb,
         long long sd, long long sg, long long sp) {
    MD_REQUIRE(vols, "%s: null volume list", fn);
    MD_REQUIRE(N >= 1 && N <= MAXF, "%s: %d lookup frames unsupported (1..%d)", fn, N, MAXF);
    MD_REQUIRE(B > 0 && B <= 65535 && D > 0 && G > 0 && hw > 0, "%s: bad dims", fn);
    for (int f = 0; f < MAXF; ++f) a.vol[f] = f < N ? vols[f] : nullptr;
    for (int f = 0; f < N; ++f) MD_REQUIRE(vols[f], "%s: null volume %d", fn, f);
    a.N = N; a.B = B; a.D = D; a.G = G; a.hw = hw; a.sb = sb; a.sd = sd; a.sg = sg; a.sp = sp;
    return MD_OK;
}

}  // namespace

extern "C" int md_fuse_fwd(const float *const *vols, int N, int B, int D, int G, int hw, long long sb, long long sd,
                           long long sg, long long sp, int eval_mode, float *out, float *weights, md_stream_t stream) {
    FuseArgs a{};
    int rc = fill(a, "md_fuse_fwd", vols, N, B, D, G, hw, sb, sd, sg, sp);
    if (rc) return rc;
    MD_REQUIRE(out, "md_fuse_fwd: null output");
    a.out = out; a.weights = weights; a.eval_mode = eval_mode != 0;
    hipLaunchKernelGGL(fuse_fwd_kernel, dim3(md_cdiv(hw, 256), B), dim3(256), 0, (hipStream_t)stream, a);
    MD_CHECK_LAUNCH("md_fuse_fwd");
    return MD_OK;
}

extern "C" int md_fuse_bwd(const float *gout, const float *const *vols, int N, int B, int D, int G, int hw,
                           long long sb, long long sd, long long sg, long long sp, float *const *d_vols,
                           md_stream_t stream) {
    FuseArgs a{};
    int rc = fill(a, "md_fuse_bwd", vols, N, B, D, G, hw, sb, sd, sg, sp);
    if (rc) return rc;
    MD_REQUIRE(gout && d_vols, "md_fuse_bwd: null gradient");
    for (int f = 0; f < N; ++f) {
        MD_REQUIRE(d_vols[f], "md_fuse_bwd: null d_vols[%d]", f);
        a.dvol[f] = d_vols[f];
    }
    a.gout = gout;
    hipLaunchKernelGGL(fuse_bwd_kernel, dim3(md_cdiv(hw, 256), B), dim3(256), 0, (hipStream_t)stream, a);
    MD_CHECK_LAUNCH("md_fuse_bwd");
    return MD_OK;
}
