// 3x3x3, stride 1, pad 1, bias-free convolution with ONE output channel over a channels-last (NDHWC) volume:
// reg3d's last layer `prob` (reference networks/resnet_encoder.py:254, called :277), the producer of the logits
// that md_softmax_entropy_localmax consumes (trainer.py:366-371).
//
//   y[b,d,h,w]      = sum_{kd,kh,kw,c} x[b,d+kd-1,h+kh-1,w+kw-1,c] * wt[kd,kh,kw,c]          (zero padding)
//   dx[b,d,h,w,c]   = sum_{kd,kh,kw}   gy[b,d-kd+1,h-kh+1,w-kw+1] * wt[kd,kh,kw,c]
//   dwt[kd,kh,kw,c] = sum_{b,d,h,w}    x[b,d,h,w,c] * gy[b,d-kd+1,h-kh+1,w-kw+1]
//
// All three are memory-bound (3.8 GFLOP against 283 MB of x at B=6, 16x96x48x160); the library's GEMM-shaped
// kernels take 1177 / 285 / 2568 us for them on MI355X (profiles/r01_reg3d_layers_miopen.txt).
//
// Common structure.  A workgroup owns an 8x32 (h,w) column of one sample and marches over a slice of D, one plane
// per step.  Lane = (voxel, channel quad): consecutive lanes hold consecutive 16-byte pieces of memory, so every
// global access of x / dx is a fully coalesced 1 KB per wave, and the 27 x QN per-thread weights (or weight-gradient
// accumulators) live in registers for the whole march.
//   fwd:   the current x plane (+1 halo) is staged in LDS in memory order (double-buffered through registers);
//          a thread reads its 9 in-plane neighbours once and scatters into three running sums (d-1, d, d+1), so
//          LDS traffic is 9 x 64 B per voxel instead of 27 x 64 B; the quads of a voxel meet by wave shuffles.
//   bwd:   a ring of three gy planes (+1 halo, 1.4 KB each) is kept in LDS; 27 scalar LDS reads per thread feed
//          108 FMAs with the register-resident weights (bwd-data) or accumulators (bwd-weight).
//   bwd-weight ends with a shuffle reduction over the lanes of equal quad, an LDS reduction over the 4 waves, one
//   partial [27*C] per workgroup in the caller's workspace and a second kernel that adds the partials in a fixed
//   order (deterministic; no float atomics).
#include "md_common.hpp"

namespace {

constexpr int TH = 8, TW = 32, HW_ = TW + 2, HH_ = TH + 2, CELLS = HH_ * HW_;  // tile and its 1-voxel halo

struct C1Dims {
    int B, C, D, H, W;
    int tiles_x, tiles, dslices, planes;  // planes per D slice
};

__device__ __forceinline__ void c1_item(const C1Dims &dm, int item, int &b, int &ty0, int &tx0, int &d0, int &d1) {
    const int sl = item % dm.dslices;
    const int t = (item / dm.dslices) % dm.tiles;
    b = item / (dm.dslices * dm.tiles);
    tx0 = (t % dm.tiles_x) * TW;
    ty0 = (t / dm.tiles_x) * TH;
    d0 = sl * dm.planes;
    d1 = min(d0 + dm.planes, dm.D);
}

// sum over the QN lanes that hold one voxel's channel quads (lane bits 0..log2(QN)-1)
template <int QN>
__device__ __forceinline__ float quad_sum(float v) {
#pragma unroll
    for (int o = 1; o < QN; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------ forward
template <int QN>
__global__ __launch_bounds__(256) void conv3d_c1_fwd_kernel(const float *__restrict__ x, const float *__restrict__ wt,
                                                            long long wsk, long long wsc, float *__restrict__ y,
                                                            const C1Dims dm) {
    constexpr int NV = QN;                     // voxels per thread: 256 voxels / (256 / QN) voxel lanes
    constexpr int VL = 256 / QN;               // voxel lanes per pass
    constexpr int NLD = (CELLS * QN + 255) / 256;
    __shared__ float4 tile[2][CELLS * QN];
    const int tid = threadIdx.x, q = tid % QN, vl = tid / QN;
    int b, ty0, tx0, d0, d1;
    c1_item(dm, blockIdx.x, b, ty0, tx0, d0, d1);
    if (d0 >= d1) return;
    // weights in LDS, [tap][quad] float4: with them in registers (27 x 4 per thread) the kernel needed 338 VGPRs = one wave
    // per SIMD; read per tap inside a tap-outer / voxel-inner loop they cost 3 registers at a time
    __shared__ float4 wsh[27 * QN];
    if (tid < 27 * QN) {
        const int k = tid / QN, qq = tid % QN;
        const float *p = wt + k * wsk + (long long)(qq * 4) * wsc;
        wsh[tid] = make_float4(p[0], p[wsc], p[2 * wsc], p[3 * wsc]);
    }
    const size_t plane = (size_t)dm.H * dm.W;
    const float4 *xb = reinterpret_cast<const float4 *>(x) + (size_t)b * dm.D * plane * QN;
    // staging roles: piece idx = cell * QN + quad, in memory order along a row
    int lofs[NLD];       // float4 offset inside a plane, or -1 (outside the image / past the tile)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + i * 256, cell = idx / QN, qq = idx % QN;
        const int gy_ = ty0 - 1 + cell / HW_, gx = tx0 - 1 + cell % HW_;
        lofs[i] = (idx < CELLS * QN && gy_ >= 0 && gy_ < dm.H && gx >= 0 && gx < dm.W) ? (gy_ * dm.W + gx) * QN + qq : -1;
    }
    const int p0 = max(d0 - 1, 0), p1 = min(d1 + 1, dm.D);  // planes [p0, p1)
    float4 pre[NLD];
    auto fetch = [&](int p) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            pre[i] = lofs[i] >= 0 ? xb[(size_t)p * plane * QN + lofs[i]] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (tid + i * 256 < CELLS * QN) tile[buf][tid + i * 256] = pre[i];
    };
    fetch(p0);
    stash(p0 & 1);
    __syncthreads();
    // voxel j of this thread: vid = j * VL + vl  ->  (ty, tx) in the tile
    float2 acc[NV][3];
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) acc[j][kd] = make_float2(0.f, 0.f);
    for (int p = p0; p < p1; ++p) {
        if (p + 1 < p1) fetch(p + 1);
        const float4 *tl = tile[p & 1];
        // the 9 in-plane taps as a ROLLED loop: fully unrolled, the compiler hoists all 63 LDS reads of a plane step and
        // needs 370 VGPRs (one wave per SIMD; neither a sched_barrier nor a compiler memory barrier per tap stops it);
        // rolled it is 116
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll 1
            for (int kw = 0; kw < 3; ++kw) {
                float4 w3[3];
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) w3[kd] = wsh[((kd * 3 + kh) * 3 + kw) * QN + q];
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int vid = j * VL + vl, ty = vid / TW, tx = vid % TW;
                    const float4 v = tl[((ty + kh) * HW_ + tx + kw) * QN + q];
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd) {  // plane p is tap kd of output d = p + 1 - kd
                        acc[j][kd].x = fmaf(v.x, w3[kd].x, acc[j][kd].x);
                        acc[j][kd].y = fmaf(v.y, w3[kd].y, acc[j][kd].y);
                        acc[j][kd].x = fmaf(v.z, w3[kd].z, acc[j][kd].x);
                        acc[j][kd].y = fmaf(v.w, w3[kd].w, acc[j][kd].y);
                    }
                }
            }
        // output plane p-1 is complete (its kd=2 tap was plane p); the last plane of the volume completes plane p too
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int d = e ? p : p - 1;
            if (e && p + 1 < dm.D) break;
            if (d < d0 || d >= d1) continue;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const float2 a = e ? acc[j][1] : acc[j][2];
                const float s = quad_sum<QN>(a.x + a.y);
                const int vid = j * VL + vl, gy_ = ty0 + vid / TW, gx = tx0 + vid % TW;
                if (q == 0 && gy_ < dm.H && gx < dm.W) y[((size_t)b * dm.D + d) * plane + (size_t)gy_ * dm.W + gx] = s;
            }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            acc[j][2] = acc[j][1];
            acc[j][1] = acc[j][0];
            acc[j][0] = make_float2(0.f, 0.f);
        }
        if (p + 1 < p1) {
            stash((p + 1) & 1);  // the other buffer: last read during step p-1, separated by the barrier below
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------ gy ring
// Three gy planes (+1 halo) in LDS; plane P lives in slot (P + 3) % 3; planes outside [0, D) and cells outside the
// image are zero.
struct GyRing {
    float *s;  // [3][CELLS]
    __device__ __forceinline__ float *slot(int P) { return s + ((P + 3) % 3) * CELLS; }
};

__device__ __forceinline__ void gy_fetch(const float *__restrict__ gyb, const C1Dims &dm, int P, int ty0, int tx0, float (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int cell = threadIdx.x + i * 256;
        const int gy_ = ty0 - 1 + cell / HW_, gx = tx0 - 1 + cell % HW_;
        const bool ok = cell < CELLS && P >= 0 && P < dm.D && gy_ >= 0 && gy_ < dm.H && gx >= 0 && gx < dm.W;
        r[i] = ok ? gyb[((size_t)P * dm.H + gy_) * dm.W + gx] : 0.f;
    }
}

__device__ __forceinline__ void gy_stash(float *slot, const float (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int cell = threadIdx.x + i * 256;
        if (cell < CELLS) slot[cell] = r[i];
    }
}

// ------------------------------------------------------------------------------------------------ backward (data)
template <int QN>
__global__ __launch_bounds__(256) void conv3d_c1_bwd_data_kernel(const float *__restrict__ gy, const float *__restrict__ wt,
                                                                 long long wsk, long long wsc, float *__restrict__ dx,
                                                                 const C1Dims dm) {
    constexpr int NV = QN, VL = 256 / QN;
    __shared__ float ring[3 * CELLS];
    GyRing R{ring};
    const int tid = threadIdx.x, q = tid % QN, vl = tid / QN;
    int b, ty0, tx0, d0, d1;
    c1_item(dm, blockIdx.x, b, ty0, tx0, d0, d1);
    if (d0 >= d1) return;
    float4 wr[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const float *p = wt + k * wsk + (long long)(q * 4) * wsc;
        wr[k] = make_float4(p[0], p[wsc], p[2 * wsc], p[3 * wsc]);
    }
    const size_t plane = (size_t)dm.H * dm.W;
    const float *gyb = gy + (size_t)b * dm.D * plane;
    float4 *dxb = reinterpret_cast<float4 *>(dx) + (size_t)b * dm.D * plane * QN;
    float r[2];
    gy_fetch(gyb, dm, d0 - 1, ty0, tx0, r); gy_stash(R.slot(d0 - 1), r);
    gy_fetch(gyb, dm, d0, ty0, tx0, r);     gy_stash(R.slot(d0), r);
    gy_fetch(gyb, dm, d0 + 1, ty0, tx0, r);  // plane d+1 travels in registers until the top of step d
    for (int d = d0; d < d1; ++d) {
        gy_stash(R.slot(d + 1), r);
        __syncthreads();
        if (d + 1 < d1) gy_fetch(gyb, dm, d + 2, ty0, tx0, r);
#pragma unroll 1  // rolled: one voxel's 27 taps at a time keeps the register count (and with it the occupancy) in check
        for (int j = 0; j < NV; ++j) {
            const int vid = j * VL + vl, ty = vid / TW, tx = vid % TW;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const float *sl = R.slot(d - kd + 1);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const float g = sl[(ty - kh + 2) * HW_ + tx - kw + 2];
                        const float4 w4 = wr[(kd * 3 + kh) * 3 + kw];
                        a.x = fmaf(g, w4.x, a.x); a.y = fmaf(g, w4.y, a.y);
                        a.z = fmaf(g, w4.z, a.z); a.w = fmaf(g, w4.w, a.w);
                    }
            }
            const int gy_ = ty0 + ty, gx = tx0 + tx;
            if (gy_ < dm.H && gx < dm.W) dxb[((size_t)d * plane + (size_t)gy_ * dm.W + gx) * QN + q] = a;
        }
        __syncthreads();  // slot (d+2)%3 == slot (d-1)%3 is rewritten at the top of the next step
    }
}

// ------------------------------------------------------------------------------------------------ backward (weight)
template <int QN>
__global__ __launch_bounds__(256) void conv3d_c1_bwd_weight_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                   float *__restrict__ partial, const C1Dims dm) {
    constexpr int NV = QN, VL = 256 / QN;
    __shared__ float ring[3 * CELLS];
    __shared__ float4 red[4][27 * QN];
    GyRing R{ring};
    const int tid = threadIdx.x, q = tid % QN, vl = tid / QN;
    int b, ty0, tx0, d0, d1;
    c1_item(dm, blockIdx.x, b, ty0, tx0, d0, d1);
    float4 acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d0 < d1) {
        const size_t plane = (size_t)dm.H * dm.W;
        const float *gyb = gy + (size_t)b * dm.D * plane;
        const float4 *xb = reinterpret_cast<const float4 *>(x) + (size_t)b * dm.D * plane * QN;
        long long xofs[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int vid = j * VL + vl, gy_ = ty0 + vid / TW, gx = tx0 + vid % TW;
            xofs[j] = (gy_ < dm.H && gx < dm.W) ? ((long long)gy_ * dm.W + gx) * QN + q : -1;
        }
        float r[2];
        gy_fetch(gyb, dm, d0 - 1, ty0, tx0, r); gy_stash(R.slot(d0 - 1), r);
        gy_fetch(gyb, dm, d0, ty0, tx0, r);     gy_stash(R.slot(d0), r);
        gy_fetch(gyb, dm, d0 + 1, ty0, tx0, r);
        for (int d = d0; d < d1; ++d) {
            float4 xv[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j)
                xv[j] = xofs[j] >= 0 ? xb[(size_t)d * plane * QN + xofs[j]] : make_float4(0.f, 0.f, 0.f, 0.f);
            gy_stash(R.slot(d + 1), r);
            __syncthreads();
            if (d + 1 < d1) gy_fetch(gyb, dm, d + 2, ty0, tx0, r);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int vid = j * VL + vl, ty = vid / TW, tx = vid % TW;
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    const float *sl = R.slot(d - kd + 1);
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const float g = sl[(ty - kh + 2) * HW_ + tx - kw + 2];
                            float4 &a = acc[(kd * 3 + kh) * 3 + kw];
                            a.x = fmaf(xv[j].x, g, a.x); a.y = fmaf(xv[j].y, g, a.y);
                            a.z = fmaf(xv[j].z, g, a.z); a.w = fmaf(xv[j].w, g, a.w);
                        }
                }
            }
            __syncthreads();
        }
    }
    // lanes of equal quad within the wave, then the 4 waves, then one partial per workgroup
#pragma unroll
    for (int k = 0; k < 27; ++k) {
#pragma unroll
        for (int o = QN; o < 64; o <<= 1) {
            acc[k].x += __shfl_xor(acc[k].x, o, 64); acc[k].y += __shfl_xor(acc[k].y, o, 64);
            acc[k].z += __shfl_xor(acc[k].z, o, 64); acc[k].w += __shfl_xor(acc[k].w, o, 64);
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < QN) {
#pragma unroll
        for (int k = 0; k < 27; ++k) red[wave][k * QN + lane] = acc[k];
    }
    __syncthreads();
    if (tid < 27 * QN) {
        const float4 a0 = red[0][tid], a1 = red[1][tid], a2 = red[2][tid], a3 = red[3][tid];
        float4 s;
        s.x = (a0.x + a1.x) + (a2.x + a3.x); s.y = (a0.y + a1.y) + (a2.y + a3.y);
        s.z = (a0.z + a1.z) + (a2.z + a3.z); s.w = (a0.w + a1.w) + (a2.w + a3.w);
        reinterpret_cast<float4 *>(partial)[(size_t)blockIdx.x * 27 * QN + tid] = s;  // [wg][k][c]
    }
}

// dwt[k*dsk + c*dsc] = sum over workgroups of partial[wg][k*C + c], fixed order, fp64 accumulation
__global__ __launch_bounds__(256) void conv3d_c1_bwd_weight_finish_kernel(const float *__restrict__ partial, int nwg, int KC,
                                                                          int C, long long dsk, long long dsc,
                                                                          float *__restrict__ dwt) {
    __shared__ double sh[256];
    const int o = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nwg; i += 256) s += (double)partial[(size_t)i * KC + o];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) dwt[(o / C) * dsk + (o % C) * dsc] = (float)sh[0];
}

int c1_dims(const char *fn, int B, int C, int D, int H, int W, C1Dims &dm) {
    MD_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "%s: bad dims B=%d D=%d H=%d W=%d", fn, B, D, H, W);
    MD_REQUIRE(C == 8 || C == 16, "%s: C=%d unsupported (8 or 16 input channels)", fn, C);
    MD_REQUIRE((long long)D * H * W * C < (1ll << 31), "%s: one sample must stay below 2^31 elements", fn);
    dm.B = B; dm.C = C; dm.D = D; dm.H = H; dm.W = W;
    dm.tiles_x = md_cdiv(W, TW);
    dm.tiles = dm.tiles_x * md_cdiv(H, TH);
    // D slices: enough workgroups to fill the chip a few times over, at least 8 planes each (a slice re-reads 2
    // halo planes in the forward)
    int ds = 1;
    while (ds * 2 <= D / 8 && (long long)B * dm.tiles * ds < 1536) ds *= 2;
    dm.planes = md_cdiv(D, ds);
    dm.dslices = md_cdiv(D, dm.planes);
    return MD_OK;
}

}  // namespace

extern "C" {

int md_conv3d_c1_fwd(const float *x, const float *wt, long long w_stride_k, long long w_stride_c, float *y, int B, int C,
                     int D, int H, int W, md_stream_t stream) {
    MD_REQUIRE(x && wt && y, "md_conv3d_c1_fwd: null tensor argument");
    MD_REQUIRE(((uintptr_t)x % 16) == 0, "md_conv3d_c1_fwd: x must be 16-byte aligned");
    C1Dims dm;
    if (int rc = c1_dims("md_conv3d_c1_fwd", B, C, D, H, W, dm)) return rc;
    const dim3 grid(B * dm.tiles * dm.dslices), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (C == 8) hipLaunchKernelGGL(conv3d_c1_fwd_kernel<2>, grid, block, 0, s, x, wt, w_stride_k, w_stride_c, y, dm);
    else hipLaunchKernelGGL(conv3d_c1_fwd_kernel<4>, grid, block, 0, s, x, wt, w_stride_k, w_stride_c, y, dm);
    MD_CHECK_LAUNCH("md_conv3d_c1_fwd");
    return MD_OK;
}

int md_conv3d_c1_bwd_data(const float *gy, const float *wt, long long w_stride_k, long long w_stride_c, float *dx, int B,
                          int C, int D, int H, int W, md_stream_t stream) {
    MD_REQUIRE(gy && wt && dx, "md_conv3d_c1_bwd_data: null tensor argument");
    MD_REQUIRE(((uintptr_t)dx % 16) == 0, "md_conv3d_c1_bwd_data: dx must be 16-byte aligned");
    C1Dims dm;
    if (int rc = c1_dims("md_conv3d_c1_bwd_data", B, C, D, H, W, dm)) return rc;
    const dim3 grid(B * dm.tiles * dm.dslices), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (C == 8) hipLaunchKernelGGL(conv3d_c1_bwd_data_kernel<2>, grid, block, 0, s, gy, wt, w_stride_k, w_stride_c, dx, dm);
    else hipLaunchKernelGGL(conv3d_c1_bwd_data_kernel<4>, grid, block, 0, s, gy, wt, w_stride_k, w_stride_c, dx, dm);
    MD_CHECK_LAUNCH("md_conv3d_c1_bwd_data");
    return MD_OK;
}

size_t md_conv3d_c1_bwd_weight_ws_bytes(int B, int C, int D, int H, int W) {
    C1Dims dm;
    if (c1_dims("md_conv3d_c1_bwd_weight_ws_bytes", B, C, D, H, W, dm)) return 0;
    return (size_t)B * dm.tiles * dm.dslices * 27 * C * sizeof(float);
}

int md_conv3d_c1_bwd_weight(const float *x, const float *gy, float *dwt, long long dw_stride_k, long long dw_stride_c,
                            void *ws, size_t ws_bytes, int B, int C, int D, int H, int W, md_stream_t stream) {
    MD_REQUIRE(x && gy && dwt && ws, "md_conv3d_c1_bwd_weight: null tensor argument");
    MD_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)ws % 16) == 0, "md_conv3d_c1_bwd_weight: x and ws must be 16-byte aligned");
    C1Dims dm;
    if (int rc = c1_dims("md_conv3d_c1_bwd_weight", B, C, D, H, W, dm)) return rc;
    const int nwg = B * dm.tiles * dm.dslices;
    MD_REQUIRE(ws_bytes >= (size_t)nwg * 27 * C * sizeof(float), "md_conv3d_c1_bwd_weight: workspace too small (%zu bytes)", ws_bytes);
    const dim3 grid(nwg), block(256);
    hipStream_t s = (hipStream_t)stream;
    float *partial = (float *)ws;
    if (C == 8) hipLaunchKernelGGL(conv3d_c1_bwd_weight_kernel<2>, grid, block, 0, s, x, gy, partial, dm);
    else hipLaunchKernelGGL(conv3d_c1_bwd_weight_kernel<4>, grid, block, 0, s, x, gy, partial, dm);
    MD_CHECK_LAUNCH("md_conv3d_c1_bwd_weight");
    hipLaunchKernelGGL(conv3d_c1_bwd_weight_finish_kernel, dim3(27 * C), block, 0, s, partial, nwg, 27 * C, C, dw_stride_k,
                       dw_stride_c, dwt);
    MD_CHECK_LAUNCH("md_conv3d_c1_bwd_weight(finish)");
    return MD_OK;
}

}  // extern "C"
