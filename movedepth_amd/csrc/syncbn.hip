// Training-mode BatchNorm with synchronised statistics for every normalisation layer of the model: 2-D (ResNet encoders, FPN4,
// the mask network) and 3-D (the regulariser), any channel count that is a multiple of 4 up to 4096, optional fused ReLU.
//
// Why: the reference's data-parallel path converts every BatchNorm to torch.nn.SyncBatchNorm (trainer.py:69-135, one all-gather
// of per-layer statistics forward, one all-reduce backward).  torch builds SyncBatchNorm from its native batch-norm kernels,
// which on this GPU are up to 2x slower than the library's on channels-last tensors: a rank of a --ddp run paid 14 % of a step
// (49.5 against 43.5 ms on one GPU) before any collective.  These kernels are that path done once: four passes over the tensor
// per layer (statistics, apply; two sums, dx), each a single launch, the statistics handed to the caller as 2*C sums so that ONE
// all-reduce per layer and direction makes them global.
//
//   forward    sums[0:C] = sum x, sums[C:2C] = sum x^2 (double)            md_bn_stats        -> caller all-reduces sums
//              mean = S1/n, var = S2/n - mean^2 (double), invstd = 1/sqrt(var + eps)
//              y = (x - mean) * invstd * gamma + beta, optionally max(0, .)  md_bn_apply       (also: mean / invstd for the
//                                                                                              backward, running statistics)
//   backward   dz = dy [* (z > 0)], sums[0:C] = sum dz = dbeta, sums[C:2C] = sum dz * xhat = dgamma   md_bn_bwd_reduce
//              dx = gamma * invstd * (dz - S1/n - xhat * S2/n)                                         md_bn_bwd_dx
// Layout: channels-last, x[row * C + c] (torch.channels_last / channels_last_3d / a (N, C) matrix).  A thread owns one
// 16-byte channel quad and strides over rows: every access is 16 bytes, a wave reads 1 KB contiguous.  Reductions are
// deterministic and need no second launch: each block writes its partial sums, takes a ticket, and the block that draws the last
// ticket adds the partials in block order in double (bn_block_finish: how that is done without a cache-flushing fence).
#include "md_common.hpp"

namespace {

constexpr int BN_MAX_PARTIAL_FLOATS = 32768;   // partial sums a finishing block adds up (128 KB: a few microseconds)

struct BnGeo {
    int C, QN, NT;      // channels, channel quads, threads per block of the element-wise kernels (a multiple of QN or QN itself)
    int NTR;            // threads per block of the reduction kernels: 1024 on large tensors (the number of blocks is capped by what
                        // the finishing block can add up, so a block has to carry the occupancy: 256 blocks x 4 waves streamed at
                        // a quarter of the memory bandwidth)
    long long npieces;  // rows * QN 16-byte pieces
    int nblk;           // blocks of the reduction kernels
};

__host__ inline bool bn_geo(long long nrows, int C, BnGeo &g) {
    if (C < 4 || C % 4 != 0 || C > 4096 || nrows <= 0) return false;
    g.C = C; g.QN = C / 4;
    // every block covers whole rows: 256 threads when the quads divide 256, else the next multiple of the quad count <= 1024
    if (256 % g.QN == 0) g.NT = 256;
    else if (g.QN <= 1024) g.NT = ((256 + g.QN - 1) / g.QN) * g.QN <= 1024 ? ((256 + g.QN - 1) / g.QN) * g.QN : g.QN;
    else return false;
    if (g.NT > 1024) return false;
    g.npieces = nrows * g.QN;
    // Reductions: 1024 threads per block wherever the quads divide it -- the finishing block reads nblk * 2C partial sums past
    // every cache (~1.5 us per batch of eight loads per thread), so few fat blocks: at most 16384 / C of them (512 channels: 32
    // blocks = two batches per thread), each streaming with 16 waves
    g.NTR = (1024 % g.QN == 0 && g.npieces >= 4096) ? 1024 : g.NT;
    long long want = (g.npieces + (long long)g.NTR * 8 - 1) / ((long long)g.NTR * 8);   // >= 8 pieces per thread
    long long cap = 16384 / C;
    if (cap < 16) cap = 16;
    if (cap * 2 * C > BN_MAX_PARTIAL_FLOATS + 2 * 4096 * 8) cap = (BN_MAX_PARTIAL_FLOATS + 2 * 4096 * 8) / (2 * C);
    if (cap > 1024) cap = 1024;
    if (want < 1) want = 1;
    g.nblk = (int)(want < cap ? want : cap);
    return true;
}

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// Element types of the activations and their gradients (x, y, dy, dx): float, or the 2-byte types of a mixed-precision step
// (torch.autocast hands BatchNorm its input in bf16 / fp16 and expects the output in the same type; sums, statistics, gamma,
// beta and their gradients stay fp32 / double as in the library's kernels).  A piece = 4 channels = 16 or 8 bytes.
struct bf16_t { unsigned short v; };
struct f16_t { _Float16 v; };
__device__ __forceinline__ float4 ld_stream(const float *p, long long i) {   // read-once streams: non-temporal (DESIGN 4.1 "cache policy")
    typedef float nt4_t __attribute__((ext_vector_type(4)));
    const nt4_t v = __builtin_nontemporal_load(reinterpret_cast<const nt4_t *>(p) + i);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 ld_stream(const bf16_t *p, long long i) {
    typedef unsigned nt2_t __attribute__((ext_vector_type(2)));
    const nt2_t v = __builtin_nontemporal_load(reinterpret_cast<const nt2_t *>(p) + i);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ float4 ld_stream(const f16_t *p, long long i) {
    // (halves taken out of the two dwords one by one: with the dwords bit-cast to 2-vectors of _Float16 the compiler loaded only
    // the first dword of the piece -- channels 2 and 3 of every quad came out wrong)
    typedef unsigned nt2_t __attribute__((ext_vector_type(2)));
    const nt2_t v = __builtin_nontemporal_load(reinterpret_cast<const nt2_t *>(p) + i);
    const unsigned lo = v.x, hi = v.y;
    return make_float4((float)__builtin_bit_cast(_Float16, (unsigned short)(lo & 0xffffu)), (float)__builtin_bit_cast(_Float16, (unsigned short)(lo >> 16)),
                       (float)__builtin_bit_cast(_Float16, (unsigned short)(hi & 0xffffu)), (float)__builtin_bit_cast(_Float16, (unsigned short)(hi >> 16)));
}
__device__ __forceinline__ void st_piece(float *p, long long i, float4 o) { reinterpret_cast<float4 *>(p)[i] = o; }
__device__ __forceinline__ unsigned bf16_rn(float f) {   // round to nearest even (NaN kept quiet), as torch's conversion
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void st_piece(bf16_t *p, long long i, float4 o) {
    reinterpret_cast<uint2 *>(p)[i] = make_uint2(bf16_rn(o.x) | (bf16_rn(o.y) << 16), bf16_rn(o.z) | (bf16_rn(o.w) << 16));
}
__device__ __forceinline__ void st_piece(f16_t *p, long long i, float4 o) {
    const unsigned a = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)o.x) | ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)o.y) << 16);
    const unsigned b = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)o.z) | ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)o.w) << 16);
    reinterpret_cast<uint2 *>(p)[i] = make_uint2(a, b);
}

// Sum of the per-thread values a, b over the threads of the block that own the same channel quad -> partial[blk][0][C] (a) and
// partial[blk][1][C] (b); then the ticket: the last block to arrive adds all partials in block order (double) into out[2C].
template <typename OUT>
__device__ __forceinline__ void bn_block_finish(float4 a, float4 b, int QN, float *__restrict__ partial, unsigned *__restrict__ counter,
                                                OUT *__restrict__ out, OUT *__restrict__ out2 = nullptr) {
    extern __shared__ float4 sh[];   // [2][NT]
    __shared__ bool last;
    const int NT = blockDim.x, tid = threadIdx.x, C = 4 * QN;
    int nper = NT / QN;   // values per quad handed to the serial sum below
    if ((QN & (QN - 1)) == 0 && QN <= 32) {
        // few quads (8 ... 128 channels): the lanes of a wave that share a quad meet by shuffles first (a fixed tree), so that the
        // serial sum below adds one value per wave instead of NT / QN of them -- with 8 channels and 1024 threads that loop was
        // 512 dependent LDS reads by two threads, 20 us of a 39 us launch
        for (int o = QN; o < 64; o <<= 1) {
            a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64); a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
            b.x += __shfl_xor(b.x, o, 64); b.y += __shfl_xor(b.y, o, 64); b.z += __shfl_xor(b.z, o, 64); b.w += __shfl_xor(b.w, o, 64);
        }
        const int lane = tid & 63, wave = tid >> 6;
        nper = NT >> 6;
        if (lane < QN) { sh[wave * QN + lane] = a; sh[NT + wave * QN + lane] = b; }
    } else {
        sh[tid] = a;
        sh[NT + tid] = b;
    }
    __syncthreads();
    // The partial sums travel between workgroups -- possibly on different XCDs, whose L2 caches are not coherent with each other
    // -- as RELAXED AGENT-SCOPE ATOMIC stores / loads: they are performed at the device's coherence point, past the L2.  What a
    // release fence (__threadfence) would add is a write-back of EVERY dirty line of the XCD's L2 (buffer_wbl2), per block: measured
    // 12-15 us per launch on a 1.5 MB tensor and 0.5 TB/s on a 283 MB one.  Ordering instead: a block's stores are complete
    // (acknowledged: s_waitcnt vmcnt(0)) before its thread 0 takes the ticket, and the ticket is an atomic at the same scope.
    if (tid < QN) {   // fixed order within the block
        float4 sa = sh[tid], sb = sh[NT + tid];
        for (int k = 1; k < nper; ++k) { sa = f4add(sa, sh[k * QN + tid]); sb = f4add(sb, sh[NT + k * QN + tid]); }
        float *pa = partial + (size_t)blockIdx.x * 2 * C;
        const float va[4] = {sa.x, sa.y, sa.z, sa.w}, vb[4] = {sb.x, sb.y, sb.z, sb.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            __hip_atomic_store(pa + 4 * tid + k, va[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pa + C + 4 * tid + k, vb[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0): this thread's stores are acknowledged
    __syncthreads();
    if (tid == 0) last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    // Acquire side: drop whatever this XCD's L2 holds (an invalidate, no write-back: cheap, and only this one block pays it).  A
    // partial-sum record of a small layer is shorter than a cache line, so a line is shared by blocks on different XCDs; a block's
    // own write-through store can leave the whole line in its L2 with the neighbour's half stale, and the loads below -- even at
    // agent scope -- would hit it (seen as a 0.5 % error in the first layer's gradients, in some runs only).
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // The finishing block's loads miss every cache by construction (~2 us a round trip): all threads take part -- output o by
    // NT / 2C threads, each a strided share of the blocks, eight loads in flight -- and the shares are added in thread order
    // through LDS (double), so the result does not depend on timing.
    const int KC = 2 * C;
    double *shd = reinterpret_cast<double *>(sh);   // NT doubles: fits the 2 NT float4 of the first phase
    const int nblk = (int)gridDim.x;
    for (int o0 = 0; o0 < KC; o0 += NT) {
        const int nout = KC - o0 < NT ? KC - o0 : NT;        // outputs of this pass
        const int nsh = NT / nout;                           // threads per output (>= 1)
        const int o = o0 + tid % nout, share = tid / nout;
        double s = 0.0;
        if (share < nsh) {
            int blk = share;
            for (; blk + 7 * nsh < nblk; blk += 8 * nsh) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = __hip_atomic_load(partial + (size_t)(blk + u * nsh) * KC + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int u = 0; u < 8; ++u) s += (double)v[u];
            }
            for (; blk < nblk; blk += nsh) s += (double)__hip_atomic_load(partial + (size_t)blk * KC + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        shd[tid] = s;
        __syncthreads();
        if (tid < nout) {
            double t = 0.0;
            for (int k = 0; k < nsh; ++k) t += shd[k * nout + tid];
            out[o0 + tid] = (OUT)t;
            if (out2) out2[o0 + tid] = (OUT)t;   // a second copy for the caller to all-reduce in place (the first stays this rank's)
        }
    }
    if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch on this workspace
}

template <typename IO>
__global__ void bn_stats_kernel(const IO *__restrict__ x, BnGeo g, float *__restrict__ partial, unsigned *__restrict__ counter,
                                double *__restrict__ sums) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), ss = s;
    const long long stride = (long long)gridDim.x * blockDim.x;   // a multiple of QN: the thread stays on one quad
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < g.npieces; i += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld_stream(x, i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s = f4add(s, v[u]);
            ss.x = fmaf(v[u].x, v[u].x, ss.x); ss.y = fmaf(v[u].y, v[u].y, ss.y);
            ss.z = fmaf(v[u].z, v[u].z, ss.z); ss.w = fmaf(v[u].w, v[u].w, ss.w);
        }
    }
    for (; i < g.npieces; i += stride) {
        const float4 v = ld_stream(x, i);
        s = f4add(s, v);
        ss.x = fmaf(v.x, v.x, ss.x); ss.y = fmaf(v.y, v.y, ss.y); ss.z = fmaf(v.z, v.z, ss.z); ss.w = fmaf(v.w, v.w, ss.w);
    }
    bn_block_finish<double>(s, ss, g.QN, partial, counter, sums);
}

// per-thread constants of the channel quad q: scale = gamma * invstd, shift = beta - mean * scale, from the (global) sums
struct BnQuad {
    float mean[4], invstd[4], sc[4], sf[4];
};
__device__ __forceinline__ BnQuad bn_quad_from_sums(const double *__restrict__ sums, int C, int q, double inv_n, float eps,
                                                    const float *__restrict__ gamma, const float *__restrict__ beta) {
    BnQuad r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 4 * q + k;
        // double for the subtraction only: E[x^2] - E[x]^2 from sums rounded to float loses (mean / std)^2 * 6e-8 of the variance.
        // 1/n arrives as a double; the square root is float arithmetic on the well-conditioned result (v_rsq + one Newton step)
        const double m = sums[c] * inv_n;
        double var = fma(sums[C + c], inv_n, -m * m);
        if (var < 0.0) var = 0.0;
        r.mean[k] = (float)m;
        const float ve = (float)(var + (double)eps);
        float is = __builtin_amdgcn_rsqf(ve);
        is = is * fmaf(-0.5f * ve * is, is, 1.5f);
        r.invstd[k] = is;
        r.sc[k] = gamma[c] * r.invstd[k];
        r.sf[k] = beta[c] - r.mean[k] * r.sc[k];
    }
    return r;
}
__device__ __forceinline__ BnQuad bn_quad_from_stat(const float *__restrict__ stat, int C, int q, const float *__restrict__ gamma,
                                                    const float *__restrict__ beta) {
    BnQuad r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 4 * q + k;
        r.mean[k] = stat[c];
        r.invstd[k] = stat[C + c];
        r.sc[k] = gamma[c] * r.invstd[k];
        r.sf[k] = beta[c] - r.mean[k] * r.sc[k];
    }
    return r;
}

template <bool RELU, typename IO>
__global__ void bn_apply_kernel(const IO *__restrict__ x, BnGeo g, const double *__restrict__ sums, double inv_n, double unbias, float eps,
                                float momentum, const float *__restrict__ gamma, const float *__restrict__ beta,
                                float *__restrict__ running_mean, float *__restrict__ running_var, float *__restrict__ stat,
                                IO *__restrict__ y) {
    // mean / invstd in double (divisions, a square root: ~100 instructions per channel): once per block and channel quad, handed
    // to the block's threads through LDS -- per thread, as first written, that arithmetic was a third of the kernel on large grids
    extern __shared__ float4 shc[];   // [2][QN]: scale, shift
    for (int qq = threadIdx.x; qq < g.QN; qq += blockDim.x) {
        const BnQuad cq = bn_quad_from_sums(sums, g.C, qq, inv_n, eps, gamma, beta);
        shc[qq] = make_float4(cq.sc[0], cq.sc[1], cq.sc[2], cq.sc[3]);
        shc[g.QN + qq] = make_float4(cq.sf[0], cq.sf[1], cq.sf[2], cq.sf[3]);
        if (blockIdx.x == 0) {   // what the backward needs, and BatchNorm's running statistics (unbiased variance)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ch = 4 * qq + k;
                stat[ch] = cq.mean[k];
                stat[g.C + ch] = cq.invstd[k];
                if (running_mean) running_mean[ch] = running_mean[ch] + momentum * (cq.mean[k] - running_mean[ch]);
                if (running_var) {
                    const double m = sums[ch] * inv_n;
                    double var = fma(sums[g.C + ch], inv_n, -m * m);
                    if (var < 0.0) var = 0.0;
                    running_var[ch] = running_var[ch] + momentum * ((float)(var * unbias) - running_var[ch]);   // unbias = n / (n - 1)
                }
            }
        }
    }
    __syncthreads();
    const int q = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) % g.QN);
    const float4 sc4 = shc[q], sf4 = shc[g.QN + q];
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < g.npieces; i += stride) {
        const float4 v = ld_stream(x, i);
        float4 o = make_float4(fmaf(v.x, sc4.x, sf4.x), fmaf(v.y, sc4.y, sf4.y), fmaf(v.z, sc4.z, sf4.z), fmaf(v.w, sc4.w, sf4.w));
        if (RELU) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
        st_piece(y, i, o);
    }
}

// evaluation mode: y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta [relu]
template <bool RELU, typename IO>
__global__ void bn_eval_kernel(const IO *__restrict__ x, BnGeo g, const float *__restrict__ rm, const float *__restrict__ rv,
                               float eps, const float *__restrict__ gamma, const float *__restrict__ beta, IO *__restrict__ y) {
    const int q = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) % g.QN);
    float sc[4], sf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ch = 4 * q + k;
        const float is = 1.f / sqrtf(rv[ch] + eps);
        sc[k] = gamma[ch] * is;
        sf[k] = beta[ch] - rm[ch] * sc[k];
    }
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < g.npieces; i += stride) {
        const float4 v = ld_stream(x, i);
        float4 o = make_float4(fmaf(v.x, sc[0], sf[0]), fmaf(v.y, sc[1], sf[1]), fmaf(v.z, sc[2], sf[2]), fmaf(v.w, sc[3], sf[3]));
        if (RELU) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
        st_piece(y, i, o);
    }
}

template <bool RELU, typename IO>
__global__ void bn_bwd_reduce_kernel(const IO *__restrict__ dy, const IO *__restrict__ x, BnGeo g, const float *__restrict__ stat,
                                     const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ partial,
                                     unsigned *__restrict__ counter, float *__restrict__ sums, float *__restrict__ sums_copy) {
    const int q = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) % g.QN);
    const BnQuad c = bn_quad_from_stat(stat, g.C, q, gamma, beta);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), sx = s;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    auto acc = [&](const float4 &v, const float4 &gq) {
        const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {gq.x, gq.y, gq.z, gq.w};
        float a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xv[k] - c.mean[k]) * c.invstd[k];
            const float dz = (!RELU || fmaf(xv[k], c.sc[k], c.sf[k]) > 0.f) ? gv[k] : 0.f;   // the forward's own expression
            a[k] = dz; b[k] = dz * xh;
        }
        s = f4add(s, make_float4(a[0], a[1], a[2], a[3]));
        sx = f4add(sx, make_float4(b[0], b[1], b[2], b[3]));
    };
    for (; i + stride < g.npieces; i += 2 * stride) {
        const float4 v0 = ld_stream(x, i), g0 = ld_stream(dy, i), v1 = ld_stream(x, i + stride), g1 = ld_stream(dy, i + stride);
        acc(v0, g0);
        acc(v1, g1);
    }
    for (; i < g.npieces; i += stride) acc(ld_stream(x, i), ld_stream(dy, i));
    bn_block_finish<float>(s, sx, g.QN, partial, counter, sums, sums_copy);
}

template <bool RELU, typename IO>
__global__ void bn_bwd_dx_kernel(const IO *__restrict__ dy, const IO *__restrict__ x, BnGeo g, const float *__restrict__ stat,
                                 const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ sums,
                                 float inv_n, IO *__restrict__ dx) {
    const int q = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) % g.QN);
    const BnQuad c = bn_quad_from_stat(stat, g.C, q, gamma, beta);
    float m1[4], m2[4], gi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m1[k] = sums[4 * q + k] * inv_n;
        m2[k] = sums[g.C + 4 * q + k] * inv_n;
        gi[k] = gamma[4 * q + k] * c.invstd[k];
    }
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < g.npieces; i += stride) {
        const float4 v = ld_stream(x, i), gq = ld_stream(dy, i);
        const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {gq.x, gq.y, gq.z, gq.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xv[k] - c.mean[k]) * c.invstd[k];
            const float dz = (!RELU || fmaf(xv[k], c.sc[k], c.sf[k]) > 0.f) ? gv[k] : 0.f;
            o[k] = gi[k] * (dz - m1[k] - xh * m2[k]);
        }
        st_piece(dx, i, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// grid of the element-wise kernels: enough blocks to fill the chip, a multiple of what keeps a thread on its quad
__host__ inline unsigned bn_ew_grid(const BnGeo &g) {
    long long nb = (g.npieces + (long long)g.NT * 8 - 1) / ((long long)g.NT * 8);
    if (nb < 1) nb = 1;
    if (nb > 2048) nb = 2048;
    return (unsigned)nb;
}

int bn_args(const char *fn, long long nrows, int C, BnGeo &g) {
    MD_REQUIRE(bn_geo(nrows, C, g), "%s: unsupported shape rows=%lld C=%d (C a multiple of 4, 4..4096, quad count dividing 256 or <= 1024)", fn, nrows, C);
    return MD_OK;
}

}  // namespace

extern "C" {

size_t md_bn_ws_bytes(void) { return 256 + sizeof(float) * (size_t)(BN_MAX_PARTIAL_FLOATS + 2 * 4096 * 8); }

// dtype of x / y / dy / dx: 0 float, 1 bf16, 2 fp16
#define BN_BY_DTYPE(fn, dtype, CALL)                                                                  \
    switch (dtype) {                                                                                    \
        case 0: { using IO = float; CALL; } break;                                                      \
        case 1: { using IO = bf16_t; CALL; } break;                                                     \
        case 2: { using IO = f16_t; CALL; } break;                                                      \
        default: MD_REQUIRE(false, "%s: dtype %d (0 float, 1 bf16, 2 fp16)", fn, dtype);                \
    }

int md_bn_stats(const void *x, int dtype, long long nrows, int C, double *sums, void *ws, md_stream_t stream) {
    MD_REQUIRE(x && sums && ws, "md_bn_stats: null tensor argument");
    MD_REQUIRE(((uintptr_t)x % (dtype ? 8 : 16)) == 0, "md_bn_stats: x must be aligned to a 4-channel piece");
    BnGeo g;
    if (int rc = bn_args("md_bn_stats", nrows, C, g)) return rc;
    unsigned *counter = (unsigned *)ws;
    float *partial = (float *)((char *)ws + 256);
    BN_BY_DTYPE("md_bn_stats", dtype, MD_LAUNCH_TIMED("md_bn_stats", bn_stats_kernel<IO>, dim3(g.nblk), dim3(g.NTR), 2 * g.NTR * sizeof(float4),
                                                      (hipStream_t)stream, (const IO *)x, g, partial, counter, sums))
    MD_CHECK_LAUNCH("md_bn_stats");
    return MD_OK;
}

int md_bn_apply(const void *x, int dtype, const double *sums, long long n_total, float eps, float momentum, const float *gamma,
                const float *beta, int relu, float *running_mean, float *running_var, float *stat, void *y, long long nrows, int C,
                md_stream_t stream) {
    MD_REQUIRE(x && sums && gamma && beta && stat && y, "md_bn_apply: null tensor argument");
    MD_REQUIRE(n_total >= nrows, "md_bn_apply: n_total %lld < rows %lld", n_total, nrows);
    BnGeo g;
    if (int rc = bn_args("md_bn_apply", nrows, C, g)) return rc;
    const dim3 grid(bn_ew_grid(g)), block(g.NT);
    const size_t lds = 2 * (size_t)g.QN * sizeof(float4);
    const double inv_n = 1.0 / (double)n_total, unbias = n_total > 1 ? (double)n_total / (double)(n_total - 1) : 1.0;
    if (relu) { BN_BY_DTYPE("md_bn_apply", dtype, MD_LAUNCH_TIMED("md_bn_apply", (bn_apply_kernel<true, IO>), grid, block, lds, (hipStream_t)stream, (const IO *)x, g, sums,
                                                                  inv_n, unbias, eps, momentum, gamma, beta, running_mean, running_var, stat, (IO *)y)) }
    else { BN_BY_DTYPE("md_bn_apply", dtype, MD_LAUNCH_TIMED("md_bn_apply", (bn_apply_kernel<false, IO>), grid, block, lds, (hipStream_t)stream, (const IO *)x, g, sums,
                                                             inv_n, unbias, eps, momentum, gamma, beta, running_mean, running_var, stat, (IO *)y)) }
    MD_CHECK_LAUNCH("md_bn_apply");
    return MD_OK;
}

int md_bn_eval(const void *x, int dtype, const float *running_mean, const float *running_var, float eps, const float *gamma, const float *beta,
               int relu, void *y, long long nrows, int C, md_stream_t stream) {
    MD_REQUIRE(x && running_mean && running_var && gamma && beta && y, "md_bn_eval: null tensor argument");
    BnGeo g;
    if (int rc = bn_args("md_bn_eval", nrows, C, g)) return rc;
    const dim3 grid(bn_ew_grid(g)), block(g.NT);
    if (relu) { BN_BY_DTYPE("md_bn_eval", dtype, hipLaunchKernelGGL((bn_eval_kernel<true, IO>), grid, block, 0, (hipStream_t)stream, (const IO *)x, g, running_mean, running_var,
                                                                    eps, gamma, beta, (IO *)y)) }
    else { BN_BY_DTYPE("md_bn_eval", dtype, hipLaunchKernelGGL((bn_eval_kernel<false, IO>), grid, block, 0, (hipStream_t)stream, (const IO *)x, g, running_mean, running_var,
                                                               eps, gamma, beta, (IO *)y)) }
    MD_CHECK_LAUNCH("md_bn_eval");
    return MD_OK;
}

int md_bn_bwd_reduce(const void *dy, const void *x, int dtype, const float *stat, const float *gamma, const float *beta, int relu,
                     long long nrows, int C, float *sums, float *sums_copy, void *ws, md_stream_t stream) {
    MD_REQUIRE(dy && x && stat && gamma && beta && sums && ws, "md_bn_bwd_reduce: null tensor argument");
    BnGeo g;
    if (int rc = bn_args("md_bn_bwd_reduce", nrows, C, g)) return rc;
    unsigned *counter = (unsigned *)ws;
    float *partial = (float *)((char *)ws + 256);
    const size_t lds = 2 * g.NTR * sizeof(float4);
    if (relu) { BN_BY_DTYPE("md_bn_bwd_reduce", dtype, MD_LAUNCH_TIMED("md_bn_bwd_reduce", (bn_bwd_reduce_kernel<true, IO>), dim3(g.nblk), dim3(g.NTR), lds, (hipStream_t)stream,
                                                                       (const IO *)dy, (const IO *)x, g, stat, gamma, beta, partial, counter, sums, sums_copy)) }
    else { BN_BY_DTYPE("md_bn_bwd_reduce", dtype, MD_LAUNCH_TIMED("md_bn_bwd_reduce", (bn_bwd_reduce_kernel<false, IO>), dim3(g.nblk), dim3(g.NTR), lds, (hipStream_t)stream,
                                                                  (const IO *)dy, (const IO *)x, g, stat, gamma, beta, partial, counter, sums, sums_copy)) }
    MD_CHECK_LAUNCH("md_bn_bwd_reduce");
    return MD_OK;
}

int md_bn_bwd_dx(const void *dy, const void *x, int dtype, const float *stat, const float *gamma, const float *beta, int relu,
                 const float *sums, long long n_total, long long nrows, int C, void *dx, md_stream_t stream) {
    MD_REQUIRE(dy && x && stat && gamma && beta && sums && dx, "md_bn_bwd_dx: null tensor argument");
    MD_REQUIRE(n_total >= nrows, "md_bn_bwd_dx: n_total %lld < rows %lld", n_total, nrows);
    BnGeo g;
    if (int rc = bn_args("md_bn_bwd_dx", nrows, C, g)) return rc;
    const dim3 grid(bn_ew_grid(g)), block(g.NT);
    if (relu) { BN_BY_DTYPE("md_bn_bwd_dx", dtype, MD_LAUNCH_TIMED("md_bn_bwd_dx", (bn_bwd_dx_kernel<true, IO>), grid, block, 0, (hipStream_t)stream, (const IO *)dy, (const IO *)x, g,
                                                                   stat, gamma, beta, sums, 1.f / (float)n_total, (IO *)dx)) }
    else { BN_BY_DTYPE("md_bn_bwd_dx", dtype, MD_LAUNCH_TIMED("md_bn_bwd_dx", (bn_bwd_dx_kernel<false, IO>), grid, block, 0, (hipStream_t)stream, (const IO *)dy, (const IO *)x, g,
                                                              stat, gamma, beta, sums, 1.f / (float)n_total, (IO *)dx)) }
    MD_CHECK_LAUNCH("md_bn_bwd_dx");
    return MD_OK;
}

}  // extern "C"
