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
//
// Second generation, C = 16 (what the trainer runs; the kernels above remain for C = 8 and irregular weight strides):
//   fwd:        lane = voxel (all 16 channels), so the weights are WAVE-UNIFORM and live in SGPRs (scalar loads, v_fma with a
//               scalar operand): no weight registers, no weight LDS traffic, no cross-lane sums.  16x32 tile, x plane staged
//               planar per channel quad ([quad][cell] float4, conflict-free row reads); a thread owns two vertically adjacent
//               voxels and reads 4 rows x 3 columns x 4 quads per plane (24 b128 per voxel against 63 before).
//   bwd-data:   MFMA (v_mfma_f32_16x16x4_f32, exact fp32): D[c][voxel] += A[c][tap] * B[tap][voxel], 7 K-steps over the 27 (+1
//               zero) taps; A = weights (7 registers), B = gy gathered from the LDS ring (one ds_read_b32 per MFMA); a lane's
//               4 results are channels 4q..4q+3 of one voxel = one coalesced float4 store (1 KB per wave).
//   bwd-weight: MFMA: D[tap][c] += A[tap][voxel] * B[voxel][c]; B = x as ONE DWORD PER LANE in memory order (4 voxels x 16
//               channels = 256 contiguous bytes, global -> register -> MFMA), A = gy from the LDS ring, two M tiles (taps 0-15,
//               16-26); per 4 voxels: 1 global load, 2 LDS reads, 2 MFMAs, no VALU arithmetic.
#include <type_traits>

#include "md_common.hpp"

// Experiment switches of earlier rounds (all off in the shipped library; tools/ab_build.sh builds a variant with -D...=1 and
// MOVEDEPTH_HIP_LIB selects it): the library reads nothing from the process environment.
#ifndef MD_CONV3D_C1_XCD
#define MD_CONV3D_C1_XCD 0
#endif
#ifndef MD_CONV3D_C1_GEN1
#define MD_CONV3D_C1_GEN1 0
#endif
#ifndef MD_CONV3D_C1_GLDS
#define MD_CONV3D_C1_GLDS 0
#endif
#ifndef MD_CONV3D_C1_DS
#define MD_CONV3D_C1_DS 0
#endif
#ifndef MD_CONV3D_C1_PF2
#define MD_CONV3D_C1_PF2 0
#endif


namespace {

constexpr int TH = 8, TW = 32, HW_ = TW + 2, HH_ = TH + 2, CELLS = HH_ * HW_;  // tile and its 1-voxel halo

struct C1Dims {
    int B, C, D, H, W;
    int tiles_x, tiles, dslices, planes;  // planes per D slice
    int per_xcd;                          // forward: items per XCD (0 = blockIdx is the item)
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

// ================================================================================================ second generation, C = 16
typedef float f32x4 __attribute__((ext_vector_type(4)));

// The MFMA kernels run WITHOUT workgroup barriers: each wave owns two tile rows (64 voxels) and keeps its own ring of three gy
// planes (its 2 rows + 1 halo row above and below, 34 columns) in LDS.  A wave's LDS operations execute in program order, so
// the wave that wrote a slot can read it back without s_barrier, and the four waves of a workgroup drift apart freely (a
// barrier per plane kept the memory pipeline in lock-step: 70-77 us against 52 us for the bare access pattern,
// tools/micro/c1_load_probe.hip).  gy is 1/16 of the traffic; the doubled halo rows come from L2.
//   RP = row pitch, SP = slot pitch (floats), chosen per kernel so that the operand gathers are bank-conflict-free.
template <int RP, int SP, int NS = 3>
struct WaveRing {
    static constexpr int ROWS = 4, COLS = HW_, N = ROWS * COLS;  // 136 values per plane
    static constexpr int NLD = (N + 63) / 64;
    static_assert(SP >= ((NLD * 64 - 1) / COLS) * RP + COLS, "a slot holds every lane's piece");
    float *base;   // this wave's NS slots
    int lofs[NLD];  // global offset inside a plane, -1 = zero (outside the image / beyond N)
    int sofs[NLD];  // slot-relative LDS offset
    float r[NLD];

    __device__ __forceinline__ void init(float *lds, const C1Dims &dm, int ty0, int tx0, int wave, int lane) {
        base = lds + wave * NS * SP;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = lane + 64 * i, row = idx / COLS, col = idx % COLS;
            const int yy = ty0 + 2 * wave - 1 + row, xx = tx0 - 1 + col;
            lofs[i] = (idx < N && yy >= 0 && yy < dm.H && xx >= 0 && xx < dm.W) ? yy * dm.W + xx : -1;
            sofs[i] = row * RP + col;  // idx >= N: rows 4, 5 of the slot, never read (no store under a branch)
        }
    }
    __device__ __forceinline__ int slot(int P) const { return (NS == 4 ? ((P + 4) & 3) : (P + 3) % 3) * SP; }
    // Unconditional loads from a clamped address, zeroed when they are consumed (stash): a load under a branch makes the
    // compiler wait for vmcnt(0) -- every store in flight included -- where it could count (measured: the data gradient
    // waited for its own stores at every step), and a select right after the load would wait for it on the spot.
    bool rin;
    __device__ __forceinline__ void fetch(const float *__restrict__ gyb, const C1Dims &dm, int P) {
        rin = P >= 0 && P < dm.D;
        const float *pl = gyb + (size_t)min(max(P, 0), dm.D - 1) * dm.H * dm.W;
#pragma unroll
        for (int i = 0; i < NLD; ++i) r[i] = pl[max(lofs[i], 0)];
    }
    __device__ __forceinline__ void stash(int P) {
        float *s = base + slot(P);
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            s[sofs[i]] = (rin && lofs[i] >= 0) ? r[i] : 0.f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
};

// ------------------------------------------------------------------------------------------------ backward (data), MFMA
// Group = 16 consecutive voxels of a tile row.  Lane (n = lane & 15, kq = lane >> 4): A holds w[tap 4kk+kq][channel n],
// B holds gy at voxel n displaced by tap 4kk+kq, D holds dx[voxel n][channels 4kq .. 4kq+3].
//
// gy of the WHOLE slice (planes d0-1 .. d1, tile + halo: (planes + 2) x 10 x 34 floats, 22 KB for 12 planes) is loaded into
// LDS before the first step; the march itself issues no loads.  Why: the kernel is a 283 MB store stream, and a load issued
// into it comes back late (the memory system is busy writing): with gy fetched one plane ahead every step waited for its
// fetch -- 69 us, against 50.6 us with the fetch removed and 53 us for the bare store pattern (tools/micro/c1_load_probe.hip).
// (Two other suspects were cleared first, at no gain: the per-plane workgroup barrier, and a vmcnt(0) that made each step wait
// for its own stores -- loads under branches and a loop-carried prefetch register defeat the compiler's exact vmcnt counting.)
constexpr int BD_RP = 40, BD_PS = HH_ * BD_RP;  // row pitch, plane pitch (floats) of the staged gy
constexpr int BD_MAX_PLANES = 16;

__global__ __launch_bounds__(256) void conv3d_c1_bwd_data_mfma_kernel(const float *__restrict__ gy, const float *__restrict__ wt,
                                                                      long long wsk, long long wsc, float *__restrict__ dx,
                                                                      const C1Dims dm) {
    extern __shared__ float lds[];  // [(d1 - d0 + 2)][HH_][BD_RP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, kq = lane >> 4;
    int b, ty0, tx0, d0, d1;
    c1_item(dm, blockIdx.x, b, ty0, tx0, d0, d1);
    if (d0 >= d1) return;
    const size_t plane = (size_t)dm.H * dm.W;
    const float *gyb = gy + (size_t)b * dm.D * plane;
    {  // stage: 4 values in flight per thread and round
        const int total = (d1 - d0 + 2) * CELLS;
        for (int i0 = tid; i0 < total; i0 += 4 * 256) {
            float v[4];
            int so[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256, p = i / CELLS, cell = i - p * CELLS, row = cell / HW_, col = cell - row * HW_;
                const int P = d0 - 1 + p, yy = ty0 - 1 + row, xx = tx0 - 1 + col;
                const bool ok = i < total && P >= 0 && P < dm.D && yy >= 0 && yy < dm.H && xx >= 0 && xx < dm.W;
                v[u] = ok ? gyb[(size_t)P * plane + (size_t)yy * dm.W + xx] : 0.f;
                so[u] = i < total ? p * BD_PS + row * BD_RP + col : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (so[u] >= 0) lds[so[u]] = v[u];
        }
    }
    float a[7];
    int tofs[7];
#pragma unroll
    for (int kk = 0; kk < 7; ++kk) {
        const int tap = 4 * kk + kq, t = tap < 27 ? tap : 0, kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
        a[kk] = tap < 27 ? wt[t * wsk + n * wsc] : 0.f;
        // output (d, row, col) takes gy(d - kd + 1, row - kh + 1, col - kw + 1): staged plane (d - d0) + 2 - kd, row + 2 - kh, ..
        tofs[kk] = (2 - kd) * BD_PS + (2 - kh) * BD_RP + 2 - kw + n;
    }
    float4 *dxb = reinterpret_cast<float4 *>(dx) + (size_t)b * dm.D * plane * 4;
    // this wave's 4 groups: tile rows 2*wave, 2*wave + 1, column halves 0 / 16
    int goff[4], vofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 2 * wave + (i >> 1), c0 = (i & 1) * 16, yy = ty0 + row, xx = tx0 + c0 + n;
        goff[i] = row * BD_RP + c0;
        vofs[i] = (yy < dm.H && xx < dm.W) ? (yy * dm.W + xx) * 4 + kq : -1;
    }
    __syncthreads();  // the only barrier
    for (int d = d0; d < d1; ++d) {
        const float *pl = lds + (d - d0) * BD_PS;
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], pl[tofs[kk] + goff[i]], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (vofs[i] >= 0) dxb[(size_t)d * plane * 4 + vofs[i]] = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    }
}

// ------------------------------------------------------------------------------------------------ backward (weight), MFMA
// Group = 4 consecutive voxels of a tile row.  Lane (m = lane & 15, k = lane >> 4): B holds x[voxel k][channel m] (memory
// order), A holds gy at voxel k displaced by tap m (tile 0) / tap 16+m (tile 1), D holds dwt[tap 16t + 4k + r][channel m].
template <bool RAGGED>
__global__ __launch_bounds__(256) void conv3d_c1_bwd_weight_mfma_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                        float *__restrict__ partial, const C1Dims dm) {
    // the 16 taps x 4 voxels of an A gather touch 9 (kd, kh) bases x 6 consecutive floats: bases 0, 6, .. 48 mod 64
    constexpr int RP = 70, SP = 466;  // 70 = 6, 466 = 18 (mod 64); 6 rows of 70 fit
    typedef WaveRing<RP, SP> Ring;
    __shared__ float lds[4 * 3 * SP];
    float(*red)[8][64] = reinterpret_cast<float(*)[8][64]>(lds);  // [4][8][64], after the march
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, k = lane >> 4;
    int b, ty0, tx0, d0, d1;
    c1_item(dm, blockIdx.x, b, ty0, tx0, d0, d1);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (d0 < d1) {
        const int tA = m, tB = 16 + m;
        const bool vB = tB < 27;
        const int tBc = vB ? tB : 0;
        const int kdA = tA / 9, offA = (2 - (tA / 3) % 3) * RP + 2 - tA % 3 + k;
        const int kdB = tBc / 9, offB = (2 - (tBc / 3) % 3) * RP + 2 - tBc % 3 + k;
        const size_t plane = (size_t)dm.H * dm.W;
        const float *gyb = gy + (size_t)b * dm.D * plane;
        const float *xb = x + (size_t)b * dm.D * plane * 16;
        // this wave's 16 groups: tile rows 2*wave + (g >> 3), columns (g & 7) * 4 + k
        const int xbase = ((ty0 + 2 * wave) * dm.W + tx0 + k) * 16 + m, xrow = dm.W * 16;
        auto xok = [&](int g) -> bool { return !RAGGED || (ty0 + 2 * wave + (g >> 3) < dm.H && tx0 + (g & 7) * 4 + k < dm.W); };
        auto xload = [&](int d, int g) -> float {  // raw: masked by xok when it is consumed
            const float *p0 = xb + (size_t)d * plane * 16 + xbase;  // two row pointers + immediate offsets when not ragged
            if (!RAGGED) return (g < 8 ? p0 : p0 + xrow)[(g & 7) * 64];
            // a masked lane reads the plane's first voxel (always inside the tensor): p0 itself can lie past the image when the
            // tile is ragged, and past the allocation for the last plane of the last sample
            return xok(g) ? p0[(g >> 3) * xrow + (g & 7) * 64] : xb[(size_t)d * plane * 16 + m];
        };
        Ring R;
        R.init(lds, dm, ty0, tx0, wave, lane);
        R.fetch(gyb, dm, d0 - 1); R.stash(d0 - 1);
        R.fetch(gyb, dm, d0);     R.stash(d0);
        R.fetch(gyb, dm, d0 + 1);
        // x one plane ahead of the MFMAs that consume it, in two register sets that swap roles (a copy at the top of the step
        // lets the compiler rotate it into the previous step, where it waits for loads that were only just issued)
        float xa[16], xb2[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) xa[g] = xload(d0, g);
        auto step = [&](int d, float (&cur)[16], float (&nxt)[16]) {
            R.stash(d + 1);
            R.fetch(gyb, dm, d + 2);  // unconditional (clamped plane, unused after the last step)
            if (d + 1 < d1) {  // a branch is harmless here: the kernel has no stores in flight, the next wait is vmcnt(0) anyway
#pragma unroll
                for (int g = 0; g < 16; ++g) nxt[g] = xload(d + 1, g);
            }
            // Without this the scheduler sinks the 16 loads below the MFMAs and the wait for them follows at once: no
            // prefetch distance at all (measured 74-81 us, 71 us even with the MFMAs replaced by single FMAs).
            __builtin_amdgcn_sched_barrier(0);
            const int s0 = R.slot(d + 1), s1 = R.slot(d), s2 = R.slot(d - 1);
            const float *pA = R.base + (kdA == 0 ? s0 : kdA == 1 ? s1 : s2) + offA;
            const float *pB = R.base + (kdB == 0 ? s0 : kdB == 1 ? s1 : s2) + offB;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int go = (g >> 3) * RP + (g & 7) * 4;
                const float a0 = pA[go];
                float a1 = pB[go];
                a1 = vB ? a1 : 0.f;
                const float xv = xok(g) ? cur[g] : 0.f;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, xv, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, xv, acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);  // the next step's loads stay behind this step's MFMAs
        };
        for (int d = d0; d < d1; d += 2) {
            step(d, xa, xb2);
            if (d + 1 < d1) step(d + 1, xb2, xa);
        }
    }
    // the 4 waves meet in LDS; thread o sums dwt[tap o / 16][channel o % 16] and writes the workgroup's partial
    __syncthreads();  // every wave is done with its ring
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        red[wave][rr][lane] = acc0[rr];
        red[wave][4 + rr][lane] = acc1[rr];
    }
    __syncthreads();
    for (int o = tid; o < 27 * 16; o += 256) {
        const int tap = o >> 4, c = o & 15, t = tap >> 4, tr = tap & 15, sl = (tr >> 2) * 16 + c, rg = t * 4 + (tr & 3);
        partial[(size_t)blockIdx.x * 27 * 16 + o] = (red[0][rg][sl] + red[1][rg][sl]) + (red[2][rg][sl] + red[3][rg][sl]);
    }
}

// ------------------------------------------------------------------------------------------------ forward, scalar weights
namespace fw {
constexpr int FTH = 16, FTW = 32, FHW = FTW + 2, FHH = FTH + 2, FCELLS = FHH * FHW;  // 612 cells
constexpr int PITCH = FCELLS;                                                   // 612 % 16 == 4: staging writes spread over the banks
constexpr int FNLD = (FCELLS * 4 + 255) / 256;
static_assert(PITCH % 16 == 4, "quad planes must sit 4 slots apart");
}  // namespace fw

// WL = 0: weight in channels-last order (tap stride 16, channel stride 1); WL = 1: planar (tap stride 1, channel stride 27).
// PF2: x planes fetched TWO steps ahead (two register sets that swap roles; 3 waves per SIMD) -- MD_CONV3D_C1_PF2=1, an A/B variant
template <int WL, bool PF2 = false>
__global__ __launch_bounds__(256) void conv3d_c1_fwd16_kernel(const float *__restrict__ x, const float *__restrict__ wt,
                                                              float *__restrict__ y, const C1Dims dm) {
    __shared__ float4 tile[fw::FNLD * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, r0 = wave * 4 + (lane >> 5) * 2;  // output voxels (r0, col) and (r0 + 1, col) of the tile
    int b, ty0, tx0, d0, d1;
    {
        // Workgroup w runs on XCD w % 8, each with its own L2.  Neighbouring slices of a tile share two halo planes and
        // neighbouring tiles a two-cell rim, so each XCD gets a contiguous run of items instead of every eighth one.
        int item = blockIdx.x;
        if (dm.per_xcd) {
            item = (blockIdx.x & 7) * dm.per_xcd + (blockIdx.x >> 3);
            if ((int)(blockIdx.x >> 3) >= dm.per_xcd || item >= dm.B * dm.tiles * dm.dslices) return;
        }
        const int sl = item % dm.dslices, t = (item / dm.dslices) % dm.tiles;
        b = item / (dm.dslices * dm.tiles);
        tx0 = (t % dm.tiles_x) * fw::FTW;
        ty0 = (t / dm.tiles_x) * fw::FTH;
        d0 = sl * dm.planes;
        d1 = min(d0 + dm.planes, dm.D);
    }
    if (d0 >= d1) return;
    const size_t plane = (size_t)dm.H * dm.W;
    const float4 *xb = reinterpret_cast<const float4 *>(x) + (size_t)b * dm.D * plane * 4;
    // Staging is unconditional (clamped address, zeroed when written to LDS, pieces beyond the tile into spare LDS).
    // Measured and dropped: storing an output plane one step late, ahead of the next step's loads (84 -> 100 us: the loads
    // queue behind the stores); LDS reads one round ahead (151 registers, 3 waves per SIMD: 100 us).
    int lofs[fw::FNLD], sofs[fw::FNLD];  // global float4 offset inside a plane (-1: zero), LDS slot
#pragma unroll
    for (int i = 0; i < fw::FNLD; ++i) {
        const int idx = tid + i * 256, cell = idx >> 2, qq = idx & 3;
        const int yy = ty0 - 1 + cell / fw::FHW, xx = tx0 - 1 + cell % fw::FHW;
        lofs[i] = (idx < fw::FCELLS * 4 && yy >= 0 && yy < dm.H && xx >= 0 && xx < dm.W) ? (yy * dm.W + xx) * 4 + qq : -1;
        sofs[i] = idx < fw::FCELLS * 4 ? qq * fw::PITCH + cell : idx;
    }
    const int p0 = max(d0 - 1, 0), p1 = min(d1 + 1, dm.D);
    float4 preA[fw::FNLD], preB[fw::FNLD];
    auto fetch = [&](int p, float4 (&pre)[fw::FNLD]) {
#pragma unroll
        for (int i = 0; i < fw::FNLD; ++i) pre[i] = xb[(size_t)p * plane * 4 + max(lofs[i], 0)];
    };
    fetch(p0, preA);
    if (PF2 && p0 + 1 < p1) fetch(p0 + 1, preB);
    float acc[2][3][2];  // [voxel][kd][even / odd channel chain]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) acc[j][kd][0] = acc[j][kd][1] = 0.f;
    const int gy0 = ty0 + r0, gx = tx0 + col;
    auto step = [&](int p, float4 (&pre)[fw::FNLD]) {
        __syncthreads();  // the previous plane's readers are done
#pragma unroll
        for (int i = 0; i < fw::FNLD; ++i) tile[sofs[i]] = lofs[i] >= 0 ? pre[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        if (p + (PF2 ? 2 : 1) < p1) fetch(p + (PF2 ? 2 : 1), pre);
        const float4 *tl = tile + r0 * fw::FHW + col;
        // MODE 0: every kd; 1: the halo plane below the slice (only tap kd = 0, output d0); 2: the one above (only kd = 2)
        auto taps = [&](auto mode) {
            constexpr int MODE = decltype(mode)::value;
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
#pragma unroll 1
                for (int kw = 0; kw < 3; ++kw) {
                    const float *wq = WL == 0 ? wt + kw * 16 + q * 4 : wt + q * 108 + kw;  // wave-uniform: scalar loads
                    float4 v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = tl[q * fw::PITCH + r * fw::FHW + kw];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int kh = r - j;
                            if (kh < 0 || kh > 2) continue;
#pragma unroll
                            for (int kd = 0; kd < 3; ++kd) {
                                if ((MODE == 1 && kd != 0) || (MODE == 2 && kd != 2)) continue;
                                const int tap9 = kd * 9 + kh * 3;
                                const float w0 = WL == 0 ? wq[tap9 * 16 + 0] : wq[0 * 27 + tap9];
                                const float w1 = WL == 0 ? wq[tap9 * 16 + 1] : wq[1 * 27 + tap9];
                                const float w2 = WL == 0 ? wq[tap9 * 16 + 2] : wq[2 * 27 + tap9];
                                const float w3 = WL == 0 ? wq[tap9 * 16 + 3] : wq[3 * 27 + tap9];
                                acc[j][kd][0] = fmaf(v[r].x, w0, acc[j][kd][0]);
                                acc[j][kd][1] = fmaf(v[r].y, w1, acc[j][kd][1]);
                                acc[j][kd][0] = fmaf(v[r].z, w2, acc[j][kd][0]);
                                acc[j][kd][1] = fmaf(v[r].w, w3, acc[j][kd][1]);
                            }
                        }
                }
            }
        };
        if (p < d0) taps(std::integral_constant<int, 1>());
        else if (p >= d1) taps(std::integral_constant<int, 2>());
        else taps(std::integral_constant<int, 0>());
        // plane p is tap kd of output d = p + 1 - kd: output p-1 is complete; the volume's last plane completes output p too
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int d = e ? p : p - 1;
            if (e && p + 1 < dm.D) break;
            if (d < d0 || d >= d1) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float s = e ? acc[j][1][0] + acc[j][1][1] : acc[j][2][0] + acc[j][2][1];
                if (gy0 + j < dm.H && gx < dm.W) y[((size_t)b * dm.D + d) * plane + (size_t)(gy0 + j) * dm.W + gx] = s;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                acc[j][2][c] = acc[j][1][c];
                acc[j][1][c] = acc[j][0][c];
                acc[j][0][c] = 0.f;
            }
    };
    if (PF2) {
        for (int p = p0; p < p1; p += 2) {
            step(p, preA);
            if (p + 1 < p1) step(p + 1, preB);
        }
    } else {
        for (int p = p0; p < p1; ++p) step(p, preA);
    }
}

// ------------------------------------------------------------------------------------------------ forward, LDS-DMA staging
// The same tile, thread mapping and arithmetic as conv3d_c1_fwd16_kernel, with the x plane brought in by global_load_lds_dwordx4
// (memory -> LDS without passing through registers: no staging VGPRs, no ds_write pass -- 612 of the 1380 LDS cycles of a plane).
// The DMA writes wave-uniform base + lane * 16, so a wave instruction fills 64 consecutive slots of one quad plane (cells in
// tile row-major order) and the lane's GLOBAL address is that cell's; quad planes are 640 slots apart (10 instructions of 64).
// Cells outside the image and the 28 surplus slots of a plane read 16 zero bytes (c1_zero16).
//   NBUF = 1: barrier, issue, wait, barrier, compute -- overlap comes from the other workgroups of the CU (4 fit);
//   NBUF = 2: plane p + 1 lands in the other buffer while plane p is computed, one barrier per plane (2 workgroups per CU).
__device__ const float4 c1_zero16 = {0.f, 0.f, 0.f, 0.f};
namespace fwg {
constexpr int GP = 640, NI = 4 * GP / 64 / 4;   // quad-plane pitch (slots), instructions per wave and plane (10)
}

template <int WL, int NBUF>
__global__ __launch_bounds__(256) void conv3d_c1_fwd16g_kernel(const float *__restrict__ x, const float *__restrict__ wt,
                                                               float *__restrict__ y, const C1Dims dm) {
    extern __shared__ float4 gtile[];   // NBUF x 4 x GP
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, r0 = wave * 4 + (lane >> 5) * 2;
    int b, ty0, tx0, d0, d1;
    {
        const int item = blockIdx.x;
        const int sl = item % dm.dslices, t = (item / dm.dslices) % dm.tiles;
        b = item / (dm.dslices * dm.tiles);
        tx0 = (t % dm.tiles_x) * fw::FTW;
        ty0 = (t / dm.tiles_x) * fw::FTH;
        d0 = sl * dm.planes;
        d1 = min(d0 + dm.planes, dm.D);
    }
    if (d0 >= d1) return;
    const size_t plane = (size_t)dm.H * dm.W;
    const float4 *xb = reinterpret_cast<const float4 *>(x) + (size_t)b * dm.D * plane * 4;
    // instruction j of this wave: global instruction k = wave * NI + j -> quad k / 10, cells 64 * (k % 10) + lane
    int lofs[fwg::NI];
#pragma unroll
    for (int j = 0; j < fwg::NI; ++j) {
        const int k = wave * fwg::NI + j, qq = k / 10, cell = (k % 10) * 64 + lane;
        const int yy = ty0 - 1 + cell / fw::FHW, xx = tx0 - 1 + cell % fw::FHW;
        lofs[j] = (cell < fw::FCELLS && yy >= 0 && yy < dm.H && xx >= 0 && xx < dm.W) ? (yy * dm.W + xx) * 4 + qq : -1;
    }
    const int p0 = max(d0 - 1, 0), p1 = min(d1 + 1, dm.D);
    auto stage = [&](int p, int buf) {
        const float4 *xp = xb + (size_t)p * plane * 4;
#pragma unroll
        for (int j = 0; j < fwg::NI; ++j) {
            const float4 *src = lofs[j] >= 0 ? xp + lofs[j] : &c1_zero16;
            float4 *dst = gtile + buf * 4 * fwg::GP + (wave * fwg::NI + j) * 64;   // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    float acc[2][3][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) acc[j][kd][0] = acc[j][kd][1] = 0.f;
    const int gy0 = ty0 + r0, gx = tx0 + col;
    if (NBUF == 2) stage(p0, p0 & 1);
    for (int p = p0; p < p1; ++p) {
        if (NBUF == 1) {
            __syncthreads();  // the previous plane's readers are done
            stage(p, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces have landed
        __syncthreads();                     // ... and everyone else's
        if (NBUF == 2 && p + 1 < p1) stage(p + 1, (p + 1) & 1);
        const float4 *tl = gtile + (NBUF == 2 ? (p & 1) * 4 * fwg::GP : 0) + r0 * fw::FHW + col;
        auto taps = [&](auto mode) {
            constexpr int MODE = decltype(mode)::value;
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
#pragma unroll 1
                for (int kw = 0; kw < 3; ++kw) {
                    const float *wq = WL == 0 ? wt + kw * 16 + q * 4 : wt + q * 108 + kw;  // wave-uniform: scalar loads
                    float4 v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = tl[q * fwg::GP + r * fw::FHW + kw];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int kh = r - j;
                            if (kh < 0 || kh > 2) continue;
#pragma unroll
                            for (int kd = 0; kd < 3; ++kd) {
                                if ((MODE == 1 && kd != 0) || (MODE == 2 && kd != 2)) continue;
                                const int tap9 = kd * 9 + kh * 3;
                                const float w0 = WL == 0 ? wq[tap9 * 16 + 0] : wq[0 * 27 + tap9];
                                const float w1 = WL == 0 ? wq[tap9 * 16 + 1] : wq[1 * 27 + tap9];
                                const float w2 = WL == 0 ? wq[tap9 * 16 + 2] : wq[2 * 27 + tap9];
                                const float w3 = WL == 0 ? wq[tap9 * 16 + 3] : wq[3 * 27 + tap9];
                                acc[j][kd][0] = fmaf(v[r].x, w0, acc[j][kd][0]);
                                acc[j][kd][1] = fmaf(v[r].y, w1, acc[j][kd][1]);
                                acc[j][kd][0] = fmaf(v[r].z, w2, acc[j][kd][0]);
                                acc[j][kd][1] = fmaf(v[r].w, w3, acc[j][kd][1]);
                            }
                        }
                }
            }
        };
        if (p < d0) taps(std::integral_constant<int, 1>());
        else if (p >= d1) taps(std::integral_constant<int, 2>());
        else taps(std::integral_constant<int, 0>());
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int d = e ? p : p - 1;
            if (e && p + 1 < dm.D) break;
            if (d < d0 || d >= d1) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float s = e ? acc[j][1][0] + acc[j][1][1] : acc[j][2][0] + acc[j][2][1];
                if (gy0 + j < dm.H && gx < dm.W) y[((size_t)b * dm.D + d) * plane + (size_t)(gy0 + j) * dm.W + gx] = s;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                acc[j][2][c] = acc[j][1][c];
                acc[j][1][c] = acc[j][0][c];
                acc[j][0][c] = 0.f;
            }
    }
}

int c1_dims(const char *fn, int B, int C, int D, int H, int W, C1Dims &dm) {
    MD_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "%s: bad dims B=%d D=%d H=%d W=%d", fn, B, D, H, W);
    MD_REQUIRE(C == 8 || C == 16, "%s: C=%d unsupported (8 or 16 input channels)", fn, C);
    MD_REQUIRE((long long)D * H * W * C < (1ll << 31), "%s: one sample must stay below 2^31 elements", fn);
    dm.B = B; dm.C = C; dm.D = D; dm.H = H; dm.W = W; dm.per_xcd = 0;
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

// geometry of the second-generation forward (16x32 tiles); a slice re-reads 2 halo planes, so at least 12 planes each
void c1_fwd16_dims(C1Dims &dm) {
    dm.tiles_x = md_cdiv(dm.W, fw::FTW);
    dm.tiles = dm.tiles_x * md_cdiv(dm.H, fw::FTH);
    int ds = 1;
    while (ds * 2 <= dm.D / 12 && (long long)dm.B * dm.tiles * ds < 512) ds *= 2;
    dm.planes = md_cdiv(dm.D, ds);
    dm.dslices = md_cdiv(dm.D, dm.planes);
    // off by default: measured 82.0 / 82.4 us with it against 82.8 / 80.1 us without (the halo re-reads are L2 / MALL hits
    // either way); -DMD_CONV3D_C1_XCD=1 (tools/ab_build.sh) builds it in
    dm.per_xcd = MD_CONV3D_C1_XCD ? md_cdiv(dm.B * dm.tiles * dm.dslices, 8) : 0;
}

constexpr bool c1_gen1() { return MD_CONV3D_C1_GEN1 != 0; }  // -DMD_CONV3D_C1_GEN1=1: the first-generation kernels for every shape (A/B builds)

}  // namespace


extern "C" {

int md_conv3d_c1_fwd(const float *x, const float *wt, long long w_stride_k, long long w_stride_c, float *y, int B, int C,
                     int D, int H, int W, md_stream_t stream) {
    MD_REQUIRE(x && wt && y, "md_conv3d_c1_fwd: null tensor argument");
    MD_REQUIRE(((uintptr_t)x % 16) == 0, "md_conv3d_c1_fwd: x must be 16-byte aligned");
    C1Dims dm;
    if (int rc = c1_dims("md_conv3d_c1_fwd", B, C, D, H, W, dm)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int wl = (w_stride_k == 16 && w_stride_c == 1) ? 0 : (w_stride_k == 1 && w_stride_c == 27) ? 1 : -1;
    if (C == 16 && wl >= 0 && !c1_gen1()) {
        c1_fwd16_dims(dm);
        // -DMD_CONV3D_C1_GLDS=1 / 2: LDS-DMA staging, single / double buffered (conv3d_c1_fwd16g_kernel); -DMD_CONV3D_C1_DS=n: D slices
        constexpr int glds = MD_CONV3D_C1_GLDS, dsl = MD_CONV3D_C1_DS;
        if (dsl > 0) { dm.planes = md_cdiv(D, dsl); dm.dslices = md_cdiv(D, dm.planes); dm.per_xcd = 0; }
        const dim3 grid16(dm.per_xcd ? 8 * dm.per_xcd : B * dm.tiles * dm.dslices);
        if (glds && wl == 0) {
            const dim3 gridg(B * dm.tiles * dm.dslices);
            const size_t lds = (size_t)(glds == 2 ? 2 : 1) * 4 * fwg::GP * sizeof(float4);
            if (glds == 2) {
                static const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3d_c1_fwd16g_kernel<0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                MD_REQUIRE(e2 == hipSuccess, "md_conv3d_c1_fwd: cannot raise the dynamic LDS limit");
                MD_LAUNCH_TIMED("md_conv3d_c1_fwd", (conv3d_c1_fwd16g_kernel<0, 2>), gridg, dim3(256), lds, s, x, wt, y, dm);
            } else {
                MD_LAUNCH_TIMED("md_conv3d_c1_fwd", (conv3d_c1_fwd16g_kernel<0, 1>), gridg, dim3(256), lds, s, x, wt, y, dm);
            }
            MD_CHECK_LAUNCH("md_conv3d_c1_fwd");
            return MD_OK;
        }
        constexpr bool pf2 = MD_CONV3D_C1_PF2 != 0;
        if (wl == 0 && pf2) MD_LAUNCH_TIMED("md_conv3d_c1_fwd", (conv3d_c1_fwd16_kernel<0, true>), grid16, dim3(256), 0, s, x, wt, y, dm);
        else if (wl == 0) MD_LAUNCH_TIMED("md_conv3d_c1_fwd", conv3d_c1_fwd16_kernel<0>, grid16, dim3(256), 0, s, x, wt, y, dm);
        else MD_LAUNCH_TIMED("md_conv3d_c1_fwd", conv3d_c1_fwd16_kernel<1>, grid16, dim3(256), 0, s, x, wt, y, dm);
        MD_CHECK_LAUNCH("md_conv3d_c1_fwd");
        return MD_OK;
    }
    const dim3 grid(B * dm.tiles * dm.dslices), block(256);
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
    else if (c1_gen1()) hipLaunchKernelGGL(conv3d_c1_bwd_data_kernel<4>, grid, block, 0, s, gy, wt, w_stride_k, w_stride_c, dx, dm);
    else {
        // own slicing: the staged gy of a slice must fit LDS
        int ds = dm.dslices;
        while (md_cdiv(D, ds) > BD_MAX_PLANES) ++ds;
        dm.planes = md_cdiv(D, ds);
        dm.dslices = md_cdiv(D, dm.planes);
        const size_t lds_bytes = (size_t)(dm.planes + 2) * BD_PS * sizeof(float);
        MD_LAUNCH_TIMED("md_conv3d_c1_bwd_data", conv3d_c1_bwd_data_mfma_kernel, dim3(B * dm.tiles * dm.dslices), block, lds_bytes, s, gy, wt, w_stride_k,
                           w_stride_c, dx, dm);
    }
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
    else if (c1_gen1()) hipLaunchKernelGGL(conv3d_c1_bwd_weight_kernel<4>, grid, block, 0, s, x, gy, partial, dm);
    else if (H % TH || W % TW) MD_LAUNCH_TIMED("md_conv3d_c1_bwd_weight", conv3d_c1_bwd_weight_mfma_kernel<true>, grid, block, 0, s, x, gy, partial, dm);
    else MD_LAUNCH_TIMED("md_conv3d_c1_bwd_weight", conv3d_c1_bwd_weight_mfma_kernel<false>, grid, block, 0, s, x, gy, partial, dm);
    MD_CHECK_LAUNCH("md_conv3d_c1_bwd_weight");
    hipLaunchKernelGGL(conv3d_c1_bwd_weight_finish_kernel, dim3(27 * C), block, 0, s, partial, nwg, 27 * C, C, dw_stride_k,
                       dw_stride_c, dwt);
    MD_CHECK_LAUNCH("md_conv3d_c1_bwd_weight(finish)");
    return MD_OK;
}

}  // extern "C"
