// Training-mode BatchNorm + ReLU (+ residual add) over a channels-last 16-channel volume: the two full-resolution
// normalisations of the 3-D regulariser (reference networks/resnet_encoder.py: conv0 = ConvBnReLU3D(...) :231 /
// module.py, conv11 = Sequential(ConvTranspose3d, BatchNorm3d, ReLU) :249-252, and `x = conv0 + self.conv11(x)` :264).
// As library calls these are, per layer and direction, BatchNorm (2-3 kernels) + ReLU + add: 13 passes over a 283 MB
// tensor forward and backward; fused: 3 forward (statistics; apply) and 5 backward (two sums; dx).
//
//   forward   mean_c, var_c over the N = B*D*H*W voxels (biased variance, as F.batch_norm in training)
//             y = max(0, (x - mean) * invstd * gamma + beta) [+ res]
//   backward  dz = dy * [z > 0]   (z recomputed from x: the pre-activation)
//             dgamma = sum dz * xhat,  dbeta = sum dz
//             dx = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)),   dres = dy
// The per-channel sums come back to the host side as [2][16] floats so that a data-parallel caller can all-reduce
// them between the two kernels (SyncBatchNorm semantics); every reduction is two-stage in a fixed order.
// Lane = (voxel, channel quad): 16-byte accesses, 1 KB contiguous per wave.
#include "md_common.hpp"

namespace {

#ifndef MD_BN_NBLK
#define MD_BN_NBLK 1024
#endif
constexpr int NBLK = MD_BN_NBLK;  // partial sums per launch; channels C = 4*QN with QN in {4, 8, 16}

#ifndef MD_BN_NT
#define MD_BN_NT 1
// The activation / gradient streams (283 MB each at config 2's first layer, read once per launch) carry the non-temporal hint:
// 42.0 -> 41.7 ms per training step (16 launches; bench.py on one box, base 41.91 / 42.03 against 41.62, then 41.66 / 41.70 as the
// default).  MD_BN_NT=0: plain loads (A/B).
#endif
__device__ __forceinline__ float4 ldst(const float4 *p) {
#if MD_BN_NT
    typedef float nt4_t __attribute__((ext_vector_type(4)));
    const nt4_t v = __builtin_nontemporal_load(reinterpret_cast<const nt4_t *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// block-wide sum of per-thread float4 (channels 4q..4q+3 of quad q = tid % QN): result valid in threads 0..QN-1
template <int QN>
__device__ __forceinline__ float4 quad_block_sum(float4 v, float4 *sh /*[4 waves][QN quads]*/) {
#pragma unroll
    for (int o = QN; o < 64; o <<= 1) {
        v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64);
        v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane < QN) sh[wave * QN + lane] = v;
    __syncthreads();
    if (threadIdx.x < QN) v = f4_add(f4_add(sh[threadIdx.x], sh[QN + threadIdx.x]), f4_add(sh[2 * QN + threadIdx.x], sh[3 * QN + threadIdx.x]));
    return v;
}

// partial[blk][0][c] = sum x, partial[blk][1][c] = sum x*x over the block's voxels
template <int QN>
__global__ __launch_bounds__(256) void bn_stats_kernel(const float4 *__restrict__ x, long long npieces, float *__restrict__ partial) {
    __shared__ float4 sh[4 * QN];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), ss = s;
    // consecutive threads take consecutive 16-byte pieces; the stride keeps a thread on one channel quad
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npieces; i += (long long)gridDim.x * 256) {
        const float4 v = ldst(x + i);
        s = f4_add(s, v);
        ss.x = fmaf(v.x, v.x, ss.x); ss.y = fmaf(v.y, v.y, ss.y); ss.z = fmaf(v.z, v.z, ss.z); ss.w = fmaf(v.w, v.w, ss.w);
    }
    s = quad_block_sum<QN>(s, sh);
    ss = quad_block_sum<QN>(ss, sh);
    if (threadIdx.x < QN) {
        reinterpret_cast<float4 *>(partial)[(size_t)blockIdx.x * 2 * QN + threadIdx.x] = s;
        reinterpret_cast<float4 *>(partial)[(size_t)blockIdx.x * 2 * QN + QN + threadIdx.x] = ss;
    }
}

// sums[o] = sum over blocks of partial[blk][o], o < KC = 2*C, fixed order, fp64
template <typename OUT>
__global__ __launch_bounds__(256) void bn_finish_kernel(const float *__restrict__ partial, int nblk, int KC, OUT *__restrict__ sums) {
    __shared__ double sh[256];
    const int o = threadIdx.x % KC, seg = threadIdx.x / KC, nseg = 256 / KC;
    double s = 0.0;
    int i = seg;
    // eight loads in flight per thread: one dependent load per iteration made this single-block kernel 32 us long (2048 partials)
    for (; i + 7 * nseg < nblk; i += 8 * nseg) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(i + u * nseg) * KC + o];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; i < nblk; i += nseg) s += (double)partial[(size_t)i * KC + o];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (seg == 0) {
        double t = 0.0;
        for (int k = 0; k < nseg; ++k) t += sh[k * KC + o];
        sums[o] = (OUT)t;
    }
}

// scale_c = gamma * invstd, shift_c = beta - mean * scale  ->  y = max(0, x * scale + shift) [+ res]
template <int QN>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float4 *__restrict__ x, const float *__restrict__ mean,
                                                       const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, const float4 *__restrict__ res,
                                                       long long npieces, float4 *__restrict__ y) {
    const int q = threadIdx.x % QN;
    float sc[4], sf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = q * 4 + k;
        sc[k] = gamma[c] * invstd[c];
        sf[k] = beta[c] - mean[c] * sc[k];
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npieces; i += (long long)gridDim.x * 256) {
        const float4 v = ldst(x + i);
        float4 o = make_float4(fmaxf(fmaf(v.x, sc[0], sf[0]), 0.f), fmaxf(fmaf(v.y, sc[1], sf[1]), 0.f),
                               fmaxf(fmaf(v.z, sc[2], sf[2]), 0.f), fmaxf(fmaf(v.w, sc[3], sf[3]), 0.f));
        if (res) o = f4_add(o, ldst(res + i));
        y[i] = o;
    }
}

// partial[blk][0][c] = sum dz, partial[blk][1][c] = sum dz * xhat
template <int QN>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ x,
                                                            const float *__restrict__ mean, const float *__restrict__ invstd,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            long long npieces, float *__restrict__ partial) {
    __shared__ float4 sh[4 * QN];
    const int q = threadIdx.x % QN;
    float mu[4], is[4], ga[4], be[4], sc[4], sf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = q * 4 + k;
        mu[k] = mean[c]; is[k] = invstd[c]; ga[k] = gamma[c]; be[k] = beta[c];
        sc[k] = ga[k] * is[k]; sf[k] = be[k] - mu[k] * sc[k];
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), sx = s;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npieces; i += (long long)gridDim.x * 256) {
        const float4 v = ldst(x + i), g = ldst(dy + i);
        const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w};
        float a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xv[k] - mu[k]) * is[k];
            const float dz = fmaf(xv[k], sc[k], sf[k]) > 0.f ? gv[k] : 0.f;   // the forward's own expression (bn_apply_kernel)
            a[k] = dz; b[k] = dz * xh;
        }
        s = f4_add(s, make_float4(a[0], a[1], a[2], a[3]));
        sx = f4_add(sx, make_float4(b[0], b[1], b[2], b[3]));
    }
    s = quad_block_sum<QN>(s, sh);
    sx = quad_block_sum<QN>(sx, sh);
    if (threadIdx.x < QN) {
        reinterpret_cast<float4 *>(partial)[(size_t)blockIdx.x * 2 * QN + threadIdx.x] = s;
        reinterpret_cast<float4 *>(partial)[(size_t)blockIdx.x * 2 * QN + QN + threadIdx.x] = sx;
    }
}

// dx = gamma * invstd * (dz - sum_dz / n - xhat * sum_dz_xhat / n)
template <int QN>
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ x,
                                                        const float *__restrict__ mean, const float *__restrict__ invstd,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        const float *__restrict__ sums, float inv_n, long long npieces,
                                                        float4 *__restrict__ dx) {
    const int q = threadIdx.x % QN;
    float mu[4], is[4], ga[4], be[4], m1[4], m2[4], sc[4], sf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = q * 4 + k;
        mu[k] = mean[c]; is[k] = invstd[c]; ga[k] = gamma[c]; be[k] = beta[c];
        sc[k] = ga[k] * is[k]; sf[k] = be[k] - mu[k] * sc[k];
        m1[k] = sums[c] * inv_n; m2[k] = sums[4 * QN + c] * inv_n;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npieces; i += (long long)gridDim.x * 256) {
        const float4 v = ldst(x + i), g = ldst(dy + i);
        const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xv[k] - mu[k]) * is[k];
            const float dz = fmaf(xv[k], sc[k], sf[k]) > 0.f ? gv[k] : 0.f;   // the forward's own expression (bn_apply_kernel)
            o[k] = ga[k] * is[k] * (dz - m1[k] - xh * m2[k]);
        }
        dx[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// mean, invstd (biased variance) from the two sums, and BatchNorm's running statistics (unbiased variance), in one tiny
// launch instead of a dozen host-side tensor ops per call
__global__ void bn_finalize_kernel(const double *__restrict__ sums, int C, double n, float eps, float momentum, float *__restrict__ mean,
                                   float *__restrict__ invstd, float *__restrict__ running_mean,
                                   float *__restrict__ running_var) {
    const int c = threadIdx.x;
    if (c >= C) return;
    // the two sums arrive in double: E[x^2] - E[x]^2 from sums rounded to float loses (mean/std)^2 * 6e-8 of the variance
    const double m = sums[c] / n;
    double var = sums[C + c] / n - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = running_mean[c] + momentum * ((float)m - running_mean[c]);
    if (running_var) {
        const double unb = var * (n / (n > 1.0 ? n - 1.0 : 1.0));
        running_var[c] = running_var[c] + momentum * ((float)unb - running_var[c]);
    }
}

int bn_check(const char *fn, long long nvox, int Cc) {
    MD_REQUIRE(Cc == 16 || Cc == 32 || Cc == 64, "%s: %d channels unsupported (16, 32 or 64)", fn, Cc);
    MD_REQUIRE(nvox > 0, "%s: empty volume", fn);
    return MD_OK;
}

#define MD_BN_DISPATCH(KERNEL, GRID, ...)                                                              \
    do {                                                                                            \
        if (QN == 4) hipLaunchKernelGGL(KERNEL<4>, GRID, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);      \
        else if (QN == 8) hipLaunchKernelGGL(KERNEL<8>, GRID, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL<16>, GRID, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);             \
    } while (0)

int bn_blocks(long long npieces) {
    long long n = (npieces + 255) / 256;
    return (int)(n < 1 ? 1 : (n > NBLK ? NBLK : n));
}

}  // namespace

extern "C" {

size_t md_bn_relu_ws_bytes(void) { return (size_t)NBLK * 2 * 64 * sizeof(float); }

int md_bn_relu_stats(const float *x, long long nvox, int Cc, double *sums, void *ws, md_stream_t stream) {
    MD_REQUIRE(x && sums && ws, "md_bn_relu_stats: null tensor argument");
    if (int rc = bn_check("md_bn_relu_stats", nvox, Cc)) return rc;
    const int QN = Cc / 4;
    const long long np = nvox * QN;
    const int nb = bn_blocks(np);
    MD_BN_DISPATCH(bn_stats_kernel, dim3(nb), (const float4 *)x, np, (float *)ws);
    MD_CHECK_LAUNCH("md_bn_relu_stats");
    hipLaunchKernelGGL(bn_finish_kernel<double>, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float *)ws, nb, 2 * Cc, sums);
    MD_CHECK_LAUNCH("md_bn_relu_stats(finish)");
    return MD_OK;
}

int md_bn_relu_finalize(const double *sums, long long n_total, int Cc, float eps, float momentum, float *mean, float *invstd,
                        float *running_mean, float *running_var, md_stream_t stream) {
    MD_REQUIRE(sums && mean && invstd, "md_bn_relu_finalize: null tensor argument");
    if (int rc = bn_check("md_bn_relu_finalize", n_total, Cc)) return rc;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, Cc, (double)n_total, eps, momentum, mean,
                       invstd, running_mean, running_var);
    MD_CHECK_LAUNCH("md_bn_relu_finalize");
    return MD_OK;
}

int md_bn_relu_apply(const float *x, const float *mean, const float *invstd, const float *gamma, const float *beta,
                     const float *res, long long nvox, int Cc, float *y, md_stream_t stream) {
    MD_REQUIRE(x && mean && invstd && gamma && beta && y, "md_bn_relu_apply: null tensor argument");
    if (int rc = bn_check("md_bn_relu_apply", nvox, Cc)) return rc;
    const int QN = Cc / 4;
    const long long np = nvox * QN;
    long long nb = (np + 256 * 4 - 1) / (256 * 4);
    if (nb > 65535 * 16) nb = 65535 * 16;
    MD_BN_DISPATCH(bn_apply_kernel, dim3((unsigned)nb), (const float4 *)x, mean, invstd, gamma, beta, (const float4 *)res, np, (float4 *)y);
    MD_CHECK_LAUNCH("md_bn_relu_apply");
    return MD_OK;
}

int md_bn_relu_bwd_reduce(const float *dy, const float *x, const float *mean, const float *invstd, const float *gamma,
                          const float *beta, long long nvox, int Cc, float *sums, void *ws, md_stream_t stream) {
    MD_REQUIRE(dy && x && mean && invstd && gamma && beta && sums && ws, "md_bn_relu_bwd_reduce: null tensor argument");
    if (int rc = bn_check("md_bn_relu_bwd_reduce", nvox, Cc)) return rc;
    const int QN = Cc / 4;
    const long long np = nvox * QN;
    const int nb = bn_blocks(np);
    MD_BN_DISPATCH(bn_bwd_reduce_kernel, dim3(nb), (const float4 *)dy, (const float4 *)x, mean, invstd, gamma, beta, np, (float *)ws);
    MD_CHECK_LAUNCH("md_bn_relu_bwd_reduce");
    hipLaunchKernelGGL(bn_finish_kernel<float>, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float *)ws, nb, 2 * Cc, sums);
    MD_CHECK_LAUNCH("md_bn_relu_bwd_reduce(finish)");
    return MD_OK;
}

int md_bn_relu_bwd_dx(const float *dy, const float *x, const float *mean, const float *invstd, const float *gamma,
                      const float *beta, const float *sums, long long n_total, long long nvox, int Cc, float *dx,
                      md_stream_t stream) {
    MD_REQUIRE(dy && x && mean && invstd && gamma && beta && sums && dx, "md_bn_relu_bwd_dx: null tensor argument");
    if (int rc = bn_check("md_bn_relu_bwd_dx", nvox, Cc)) return rc;
    MD_REQUIRE(n_total >= nvox, "md_bn_relu_bwd_dx: n_total %lld < nvox %lld", n_total, nvox);
    const int QN = Cc / 4;
    const long long np = nvox * QN;
    long long nb = (np + 256 * 4 - 1) / (256 * 4);
    if (nb > 65535 * 16) nb = 65535 * 16;
    MD_BN_DISPATCH(bn_bwd_dx_kernel, dim3((unsigned)nb), (const float4 *)dy, (const float4 *)x, mean, invstd, gamma, beta, sums, 1.f / (float)n_total, np, (float4 *)dx);
    MD_CHECK_LAUNCH("md_bn_relu_bwd_dx");
    return MD_OK;
}

}  // extern "C"
