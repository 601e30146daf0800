// 3x3x3, stride 1, pad 1, bias-free 16 -> 16 channel convolution over channels-last (NDHWC) volumes: reg3d's first
// layer conv0 (reference networks/resnet_encoder.py:231, applied :258), the consumer of the grouped cost volume
// (G = 16 groups = its 16 input channels).  Library times at 6x16x96x48x160 (61 GFLOP per direction,
// profiles/r01_reg3d_layers_miopen.txt): forward 1044 us, data gradient 1383 us, weight gradient 3112 us.
//
//   y[b,d,h,w,co]       = sum_{kd,kh,kw,ci} x[b,d+kd-1,h+kh-1,w+kw-1,ci] * wt[co,ci,kd,kh,kw]
//   dx[b,d,h,w,ci]      = sum_{kd,kh,kw,co} gy[b,d-kd+1,h-kh+1,w-kw+1,co] * wt[co,ci,kd,kh,kw]
//   dwt[co,ci,kd,kh,kw] = sum_{b,d,h,w} x[b,d+kd-1,h+kh-1,w+kw-1,ci] * gy[b,d,h,w,co]
//
// Weight gradient, MFMA mapping (v_mfma_f32_16x16x4_f32, exact fp32, 157 TF/s peak = the bound of this kernel): for one tap,
// D[ci][co] += sum over 4 voxels of A[ci][v] * B[v][co] with A = x at the tap-shifted voxels, B = gy.  Both operands
// are ONE dword per lane in exactly the order the channels-last volumes have in memory (lane = voxel*16 + channel,
// 4 consecutive voxels along w = 256 contiguous bytes), so gy goes global -> register -> MFMA and x goes
// LDS -> register -> MFMA with no shuffles.  Each wave keeps all 27 taps' 16x16 accumulators (108 registers).
// A workgroup owns an 8x32 (h,w) column of one sample and marches over a slice of D with a ring of three x planes
// (+1 halo, zero-filled outside the image = the zero padding) in LDS; per plane a wave does 16 voxel groups x 27
// MFMAs.  The 4 waves are summed through LDS, one partial [27][16][16] per workgroup goes to the caller's workspace
// and a second kernel adds the partials in a fixed order in fp64 (deterministic; no float atomics).
#include "md_common.hpp"

// A/B build switches (tools/ab_build.sh); the library reads nothing from the process environment.
#ifndef MD_C16_DSLICES
#define MD_C16_DSLICES 0   // n >= 1: planes per slice = ceil(D / n) instead of c16_dims' choice
#endif
#ifndef MD_C16_BF3
#define MD_C16_BF3 1       // channels-last forward / data gradient on the bf16 matrix pipe with three-piece operands (0: fp32 MFMA)
#endif
#ifndef MD_C16_BF3_WGRAD
#define MD_C16_BF3_WGRAD 1 // the same for the weight gradient
#endif

namespace {

constexpr int TH = 8, TW = 32, HW_ = TW + 2, HH_ = TH + 2, CELLS = HH_ * HW_;
constexpr int CI = 16, CO = 16, NTAP = 27;
constexpr int PLANE_F = CELLS * CI;              // floats per staged plane
constexpr int NLD = (CELLS * 4 + 255) / 256;     // float4 pieces per thread per plane

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef MD_C16_NT
#define MD_C16_NT 2
// Non-temporal hint on this file's volume loads: 1 = the weight gradient's gy stream, 2 = the staged planes of all three kernels too.
// These kernels are MFMA-bound and do not change (510 / 500 / 611 us either way); what changes is what they leave on the die: with 2
// the weight gradient's 566 MB of reads no longer push the data gradient's dx out of the Infinity Cache before md_costvol_bwd reads
// it -- 72.2 -> 69.4 us for the plane-sweep backward in the training step (bench.py, three runs each over two boxes); 1: no change.
#endif
#ifndef MD_C16_FWD_UNROLL
#define MD_C16_FWD_UNROLL 1      // group loop straight-line: the next group's LDS reads overlap the previous group's stores
#endif
#ifndef MD_C16_FWD_VEC_STORE
#define MD_C16_FWD_VEC_STORE 1   // channels-last output: D[n][voxel] mapping, one 16-byte store per lane (0: four dword stores)
#endif
#ifndef MD_C16_WGRAD_GYQ
#define MD_C16_WGRAD_GYQ 1
#endif

struct C16Dims {
    int B, D, H, W;
    int tiles_x, tiles, dslices, planes;
    // bf16 x 3 kernels only: a 16-channel BLOCK of wider channels-last volumes (md_conv3d_cb_*: Ci, Co multiples of 16 as sums over
    // 16 x 16 blocks).  Record pitch and the block's offset in float4 units, for the input (x / gy) and the output (y / dx) side;
    // acc: the output block is added to, not stored (the second and later input blocks of an output block).  4, 0, 4, 0, 0: plain 16 -> 16.
    int ip4, io4, op4, oo4, acc;
    int nob;   // forward / data gradient: output blocks covered by ONE launch (workgroup -> block blockIdx % nob; oo4 and the weights' offset follow)
};

__device__ __forceinline__ void c16_item(const C16Dims &dm, int item, int &b, int &ty0, int &tx0, int &d0, int &d1) {
    const int sl = item % dm.dslices;
    const int t = (item / dm.dslices) % dm.tiles;
    b = item / (dm.dslices * dm.tiles);
    tx0 = (t % dm.tiles_x) * TW;
    ty0 = (t / dm.tiles_x) * TH;
    d0 = sl * dm.planes;
    d1 = min(d0 + dm.planes, dm.D);
}

// Stages one tile plane (+1 halo, zeros outside the volume) of a 16-channel volume into an LDS slot through registers
// (fetch early, stash after the previous plane's consumers are done).
//   channels-last volume [B,D,H,W,16]: LDS slot = [cell][16], a straight copy in memory order (float4 pieces);
//   planar volume [B,16,D,H,W] (the cost volume's `bgd` layout): LDS slot = [16][CELLS], also a straight copy (dwords);
//     channel stride 340 floats = 20 mod 64 banks, so the MFMA operand reads (16 channels x 4 voxels, or 4 channels x
//     16 voxels per wave) touch 64 distinct banks.
template <bool PLANAR>
struct Stager {
    static constexpr int N = PLANAR ? (CELLS * CI + 255) / 256 : NLD;
    int ofs[N];
    float4 pre4[PLANAR ? 1 : N];
    float pre1[PLANAR ? N : 1];
    size_t pstride;  // elements between consecutive planes (in units of the piece type)

    __device__ __forceinline__ void init(const C16Dims &dm, int ty0, int tx0) {
        const int tid = threadIdx.x;
        const size_t plane = (size_t)dm.H * dm.W;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int idx = tid + i * 256;
            if (PLANAR) {
                const int c = idx / CELLS, cell = idx % CELLS;
                const int yy = ty0 - 1 + cell / HW_, xx = tx0 - 1 + cell % HW_;
                ofs[i] = (idx < CELLS * CI && yy >= 0 && yy < dm.H && xx >= 0 && xx < dm.W)
                             ? (int)(c * dm.D * plane) + yy * dm.W + xx : -1;
            } else {
                const int cell = idx >> 2, qq = idx & 3;
                const int yy = ty0 - 1 + cell / HW_, xx = tx0 - 1 + cell % HW_;
                ofs[i] = (idx < CELLS * 4 && yy >= 0 && yy < dm.H && xx >= 0 && xx < dm.W) ? (yy * dm.W + xx) * 4 + qq : -1;
            }
        }
        pstride = PLANAR ? plane : plane * 4;
    }
    // `base`: the sample's volume (float* for planar, float4* view for channels-last)
    __device__ __forceinline__ void fetch(const float *__restrict__ base, const C16Dims &dm, int P) {
        const bool in = P >= 0 && P < dm.D;
        if (PLANAR) {
#pragma unroll
            for (int i = 0; i < N; ++i) pre1[i] = (in && ofs[i] >= 0) ? base[(size_t)P * pstride + ofs[i]] : 0.f;
        } else {
            const float4 *b4 = reinterpret_cast<const float4 *>(base);
#pragma unroll
            for (int i = 0; i < N; ++i) {
#if MD_C16_NT >= 2
                if (in && ofs[i] >= 0) {
                    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(b4 + (size_t)P * pstride + ofs[i]));
                    pre4[i] = make_float4(v[0], v[1], v[2], v[3]);
                } else pre4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#else
                pre4[i] = (in && ofs[i] >= 0) ? b4[(size_t)P * pstride + ofs[i]] : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
            }
        }
    }
    __device__ __forceinline__ void stash(float *slot) const {
        const int tid = threadIdx.x;
        if (PLANAR) {
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (tid + i * 256 < CELLS * CI) slot[tid + i * 256] = pre1[i];
        } else {
            float4 *s4 = reinterpret_cast<float4 *>(slot);
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (tid + i * 256 < CELLS * 4) s4[tid + i * 256] = pre4[i];
        }
    }
};

template <bool XPLANAR>
__global__ __launch_bounds__(256, 2) void conv3d_c16_bwd_weight_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                      float *__restrict__ partial, const C16Dims dm) {
    __shared__ float ring[3 * PLANE_F];  // plane P in slot (P + 3) % 3; reused for the wave reduction at the end
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = lane & 15, v = lane >> 4;
    int b, ty0, tx0, d0, d1;
    c16_item(dm, blockIdx.x, b, ty0, tx0, d0, d1);
    const size_t plane = (size_t)dm.H * dm.W;
    const float *xb = x + (size_t)b * dm.D * plane * CI;
    const float *gyb = gy + (size_t)b * dm.D * plane * CO;
    Stager<XPLANAR> st;
    st.init(dm, ty0, tx0);
    auto fetch = [&](int P) { st.fetch(xb, dm, P); };
    auto stash = [&](int P) { st.stash(ring + ((P + 3) % 3) * PLANE_F); };
    // gy roles: group g of this wave = tile row 2*wave + g/8, columns (g%8)*4 .. +3; lane holds (voxel v, channel ch)
    auto gy_load = [&](int d, int g) -> float {
        const int yy = ty0 + 2 * wave + (g >> 3), xx = tx0 + (g & 7) * 4 + v;
#if MD_C16_NT >= 1
        return (yy < dm.H && xx < dm.W) ? __builtin_nontemporal_load(gyb + ((size_t)d * plane + (size_t)yy * dm.W + xx) * CO + ch) : 0.f;
#else
        return (yy < dm.H && xx < dm.W) ? gyb[((size_t)d * plane + (size_t)yy * dm.W + xx) * CO + ch] : 0.f;
#endif
    };

    f32x4 acc[NTAP];
#pragma unroll
    for (int k = 0; k < NTAP; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};

    fetch(d0 - 1); stash(d0 - 1);
    fetch(d0);     stash(d0);
    fetch(d0 + 1);
    for (int d = d0; d < d1; ++d) {
        // the plane's 16 gy operands requested up front: one group ahead is 27 MFMAs = 0.4 us, less than a trip to HBM
        // (channels-last x only: the planar variant's wider staging leaves no registers for it)
        constexpr bool GYQ = MD_C16_WGRAD_GYQ && !XPLANAR;
        float gyv[GYQ ? 16 : 1], gcur = 0.f;
        if constexpr (GYQ) {
#pragma unroll
            for (int g = 0; g < 16; ++g) gyv[g] = gy_load(d, g);
        } else {
            gcur = gy_load(d, 0);
        }
        stash(d + 1);
        __syncthreads();
        if (d + 1 < d1) fetch(d + 2);
        // element (cell, channel) of a slot sits at cell*16 + ch (channels-last) or ch*CELLS + cell (planar)
        constexpr int CS = XPLANAR ? 1 : CI;
        const int chofs = XPLANAR ? ch * CELLS : ch;
        const float *s0 = ring + ((d + 2) % 3) * PLANE_F + chofs;  // planes d-1, d, d+1
        const float *s1 = ring + (d % 3) * PLANE_F + chofs;
        const float *s2 = ring + ((d + 1) % 3) * PLANE_F + chofs;
        // operands of group g+1 are read from LDS while the 27 MFMAs of group g run (one group ahead, like gy)
        auto a_load = [&](int g, float (&av)[NTAP]) {
            const int cell = ((2 * wave + (g >> 3)) * HW_ + (g & 7) * 4 + v) * CS;  // this lane's voxel, halo origin
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const float *sl = (kd == 0 ? s0 : kd == 1 ? s1 : s2) + cell;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) av[(kd * 3 + kh) * 3 + kw] = sl[(kh * HW_ + kw) * CS];
            }
        };
        float acur[NTAP];
        a_load(0, acur);
        if constexpr (GYQ) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {   // straight-line: the compiler counts the outstanding loads exactly
            float anext[NTAP];
            if (g + 1 < 16) a_load(g + 1, anext);
#pragma unroll
            for (int k = 0; k < NTAP; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[k], gyv[g], acc[k], 0, 0, 0);
            if (g + 1 < 16) {
#pragma unroll
                for (int k = 0; k < NTAP; ++k) acur[k] = anext[k];
            }
        }
        } else {
#pragma unroll 1
        for (int g = 0; g < 16; ++g) {
            const float gnext = gy_load(d, (g + 1) & 15);  // one group ahead (the wrap-around load is discarded)
            float anext[NTAP];
            a_load((g + 1) & 15, anext);
#pragma unroll
            for (int k = 0; k < NTAP; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[k], gcur, acc[k], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NTAP; ++k) acur[k] = anext[k];
            gcur = gnext;
        }
        }
        __syncthreads();  // slot (d+2)%3 == (d-1)%3 is rewritten at the top of the next step
    }
    // sum the 4 waves through LDS (two waves' accumulators fit at a time), wave 0 writes the workgroup's partial
    f32x4 *red = reinterpret_cast<f32x4 *>(ring);  // [2][NTAP][64]
    if (wave >= 2) {
#pragma unroll
        for (int k = 0; k < NTAP; ++k) red[((wave - 2) * NTAP + k) * 64 + lane] = acc[k];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int k = 0; k < NTAP; ++k) acc[k] += red[(wave * NTAP + k) * 64 + lane];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int k = 0; k < NTAP; ++k) red[k * 64 + lane] = acc[k];
    }
    __syncthreads();
    if (wave == 0) {
        f32x4 *out = reinterpret_cast<f32x4 *>(partial) + (size_t)blockIdx.x * NTAP * 64;
#pragma unroll
        for (int k = 0; k < NTAP; ++k) out[k * 64 + lane] = acc[k] + red[k * 64 + lane];
    }
}

// partial[wg][k][lane][r] holds dW[ci = 4*(lane>>4) + r][co = lane&15] of tap k (the MFMA C/D layout).
// dwt[co*s_co + ci*s_ci + k*s_k] = sum over workgroups, fixed order, fp64.
__global__ __launch_bounds__(256) void conv3d_c16_bwd_weight_finish_kernel(const float *__restrict__ partial, int nwg,
                                                                           long long s_co, long long s_ci, long long s_k,
                                                                           float *__restrict__ dwt) {
    // block = 16 outputs x 16 segments of the workgroup list (432 blocks; 64 outputs x 4 segments on 108 blocks took 47 us
    // for 20 MB of partials)
    __shared__ double sh[16][16];
    const int ol = threadIdx.x & 15, seg = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + ol;  // element of [k][lane][r]
    double s = 0.0;
    for (int i = seg; i < nwg; i += 16) s += (double)partial[(size_t)i * (NTAP * 256) + o];
    sh[seg][ol] = s;
    __syncthreads();
    if (seg == 0) {
        double t = 0.0;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) t += sh[k2][ol];
        const int k = o >> 8, lane = (o >> 2) & 63, r = o & 3;
        const int ci = 4 * (lane >> 4) + r, co = lane & 15;
        dwt[co * s_co + ci * s_ci + k * s_k] = (float)t;
    }
}

// Forward and data gradient (one kernel): out[u][n] = sum_{tap, m} in[u + tap - 1][m] * wt(n, m, tap), where for the
// data gradient `in` is gy, (n, m) = (ci, co) and the taps are mirrored (tap -> 26 - tap).
// MFMA mapping: D[voxel][n] += A[voxel][m-quad] * B[m-quad][n], 16 consecutive voxels along w per group, 4 MFMAs per
// tap (K = 4 input channels each).  Lane (voxel = l&15, kk = l>>4) gets its four A values for one tap with a single
// ds_read_b128 (channels kk*4 .. kk*4+3 of its voxel: the 64 lanes cover 1 KB of LDS contiguously), MFMA s taking
// channel kk*4+s; the matching B values wt(n = l&15, m = kk*4+s, tap) stay in registers for the whole kernel (108).
template <bool IN_PLANAR, bool OUT_PLANAR>
__global__ __launch_bounds__(256, 2) void conv3d_c16_fwd_kernel(const float *__restrict__ in, const float *__restrict__ wt,
                                                               long long s_n, long long s_m, long long s_k, int mirror,
                                                               float *__restrict__ out, const C16Dims dm) {
    __shared__ float ring[3 * PLANE_F];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kk = lane >> 4;
    int b, ty0, tx0, d0, d1;
    c16_item(dm, blockIdx.x, b, ty0, tx0, d0, d1);
    const size_t plane = (size_t)dm.H * dm.W;
    const float *inb = in + (size_t)b * dm.D * plane * CI;
    float *outb = out + (size_t)b * dm.D * plane * CO;
    f32x4 wr[NTAP];  // wr[tap][s] = wt(n, kk*4+s, tap)
#pragma unroll
    for (int k = 0; k < NTAP; ++k) {
        const float *p = wt + n * s_n + (long long)(kk * 4) * s_m + (mirror ? 26 - k : k) * s_k;
        wr[k] = (f32x4){p[0], p[s_m], p[2 * s_m], p[3 * s_m]};
    }
    Stager<IN_PLANAR> st;
    st.init(dm, ty0, tx0);
    auto fetch = [&](int P) { st.fetch(inb, dm, P); };
    auto stash = [&](int P) { st.stash(ring + ((P + 3) % 3) * PLANE_F); };
    fetch(d0 - 1); stash(d0 - 1);
    fetch(d0);     stash(d0);
    fetch(d0 + 1);
    for (int d = d0; d < d1; ++d) {
        stash(d + 1);
        __syncthreads();
        if (d + 1 < d1) fetch(d + 2);
        const float *s0 = ring + ((d + 2) % 3) * PLANE_F;  // planes d-1, d, d+1
        const float *s1 = ring + (d % 3) * PLANE_F;
        const float *s2 = ring + ((d + 1) % 3) * PLANE_F;
#if MD_C16_FWD_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
        for (int g = 0; g < 4; ++g) {  // group = tile row 2*wave + g/2, columns (g%2)*16 .. +15
            const int row = 2 * wave + (g >> 1), col0 = (g & 1) * 16;
            const int cell = row * HW_ + col0 + n;  // this lane's voxel (n doubles as the voxel index), halo origin
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc;  // two chains: a dependent MFMA issues 8 cycles later
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const float *sl = (kd == 0 ? s0 : kd == 1 ? s1 : s2);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        float4 a;  // channels kk*4 .. kk*4+3 of the tap-shifted voxel
                        const int c = cell + kh * HW_ + kw;
                        if (IN_PLANAR) {
                            const float *pc = sl + (kk * 4) * CELLS + c;
                            a = make_float4(pc[0], pc[CELLS], pc[2 * CELLS], pc[3 * CELLS]);
                        } else {
                            a = reinterpret_cast<const float4 *>(sl)[c * 4 + kk];
                        }
                        const f32x4 w4 = wr[(kd * 3 + kh) * 3 + kw];
                        if (OUT_PLANAR || MD_C16_FWD_VEC_STORE) {  // D[n][voxel]: weights as A, input as B
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[0], a.x, acc, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[1], a.y, acc1, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[2], a.z, acc, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[3], a.w, acc1, 0, 0, 0);
                        } else {           // D[voxel][n]
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w4[0], acc, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w4[1], acc1, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w4[2], acc, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w4[3], acc1, 0, 0, 0);
                        }
                    }
            }
            acc += acc1;
            const int yy = ty0 + row;
            if (OUT_PLANAR) {
                // register r of lane l = out[channel 4*(l>>4) + r][voxel l&15]: 64 contiguous bytes per channel
                const int xx = tx0 + col0 + n;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (yy < dm.H && xx < dm.W)
                        outb[((size_t)(4 * kk + r) * dm.D + d) * plane + (size_t)yy * dm.W + xx] = acc[r];
            } else if (MD_C16_FWD_VEC_STORE) {
                // register r of lane l = out[channel 4*(l>>4) + r][voxel l&15]: four consecutive channels of one voxel = one
                // 16-byte store, the wave's 16 voxels x 64 bytes contiguous
                const int xx = tx0 + col0 + n;
                if (yy < dm.H && xx < dm.W)
                    reinterpret_cast<float4 *>(outb)[((size_t)d * plane + (size_t)yy * dm.W + xx) * 4 + kk] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            } else {
                // register r of lane l = out[voxel 4*(l>>4) + r][channel l&15]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int xx = tx0 + col0 + 4 * kk + r;
                    if (yy < dm.H && xx < dm.W) outb[((size_t)d * plane + (size_t)yy * dm.W + xx) * CO + n] = acc[r];
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Forward / data gradient on the bf16 matrix pipe with fp32 operands cut into three bf16 pieces (channels-last volumes).
//
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate (MI355X_MICROARCH.md): the kernel above issues 108 of them (32 cycles
// each) per 16 output voxels and sits at 67-69 % of that pipe's peak.  A float is the exact sum of three bf16 numbers -- its
// mantissa's three bytes: hi = x with the low 16 bits cleared, mid = (x - hi) likewise, lo = the rest; every subtraction exact --
// so x * w = sum of nine bf16 products; the six largest (hi hi, hi mid, mid hi, hi lo, lo hi, mid mid) leave out terms below
// 2^-23 of the product, the size of fp32's own rounding, and v_mfma_f32_16x16x32_bf16 forms each exactly and adds in fp32.
// Per K = 32 slice six instructions of ~17 cycles instead of eight of 32 (tools/micro/mfma_bf16x3_probe.hip).
//
// What makes it pay is doing the split ONCE per input value and tap ROW, not per tap: the three taps along w of a (kd, kh) row
// read the same voxels shifted by one, so the input operand (B: K = 2 tap rows x 16 channels, N = 16 voxels along w) is loaded
// and split once per slice and multiplied with the weights of kw = 0, 1, 2 into three accumulators P_kw[u] = W_kw . x[u]; the
// output is out[v] = P_0[v-1] + P_1[v] + P_2[v+1] -- a rotation by one lane inside each 16-lane row of the accumulator layout
// (D[co][voxel]: lane = voxel), two DPP moves per register.  So that no output needs a voxel outside the loaded ones, a tile
// row is 32 INPUT voxels (two groups of 16, columns tx0-1 .. tx0+30) and 30 outputs; the lanes that close the gap between the
// two groups take their neighbour's term from the other group (both belong to the same wave).  9 tap rows = 4.5 slices: five,
// the last one half empty.  Per 16 input voxels: 5 x 3 x 6 = 90 MFMAs (1530 cycles) against 108 x 32 = 3456, 10 ds_read_b128
// Weights: 5 slices x 3 kw x 3 pieces x 4 registers = 180 VGPRs, resident.
#ifndef MD_C16_BF3_TH
#define MD_C16_BF3_TH 4     // tile rows: 4 (one row per wave, 55 KB ring + 15 KB of weights' `lo` pieces: two workgroups per CU) or 8 (two rows per wave, 92 KB, one per CU)
#endif
constexpr int B3_TH = MD_C16_BF3_TH, B3_TW = 30, B3_XW = 32, B3_XH = B3_TH + 2, B3_CELLS = B3_XH * B3_XW;
constexpr bool B3_TWO = B3_TH == 4;   // two workgroups per CU: 256 registers each, the weights' `lo` pieces in LDS
constexpr int B3_NLD = (B3_CELLS * 4 + 255) / 256;   // float4 pieces per thread per plane
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Bf3 { u32x4 hi, mid, lo; };   // 8 values as packed bf16 pieces (element 2i in the low half of dword i)
// eight floats -> their three bf16 pieces, by truncation: x = hi + mid + lo exactly (8 + 8 + 8 mantissa bits)
__device__ __forceinline__ Bf3 bf3_split(const float *x) {
    Bf3 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h[2], m[2], l[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float v = x[2 * i + e];
            const unsigned hb = __builtin_bit_cast(unsigned, v) & 0xffff0000u;
            const float r1 = v - __builtin_bit_cast(float, hb);
            const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
            const float r2 = r1 - __builtin_bit_cast(float, mb);
            h[e] = hb; m[e] = mb; l[e] = __builtin_bit_cast(unsigned, r2);
        }
        // bytes {a2, a3, b2, b3}: the high halves of a (element 2i) and b (element 2i + 1)
        r.hi[i] = __builtin_amdgcn_perm(h[1], h[0], 0x07060302u);
        r.mid[i] = __builtin_amdgcn_perm(m[1], m[0], 0x07060302u);
        r.lo[i] = __builtin_amdgcn_perm(l[1], l[0], 0x07060302u);
    }
    return r;
}
__device__ __forceinline__ f32x4 bf3_mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the six products of one K = 32 slice: A = weights (rows = output channels), B = input (columns = voxels)
__device__ __forceinline__ f32x4 bf3_mac(u32x4 whi, u32x4 wmid, u32x4 wlo, const Bf3 &x, f32x4 c) {
    c = bf3_mfma(wlo, x.hi, c);    // smallest terms first
    c = bf3_mfma(whi, x.lo, c);
    c = bf3_mfma(wmid, x.mid, c);
    c = bf3_mfma(wmid, x.hi, c);
    c = bf3_mfma(whi, x.mid, c);
    c = bf3_mfma(whi, x.hi, c);
    return c;
}
// lane j of each 16-lane row <- lane j - 1 (row_ror:1) / j + 1 (row_ror:15) of the same row, wrapping: one DPP move per register.
// (The components go through named floats: __builtin_bit_cast(int, v[i]) on an element of the vector type read component 0 for
// every i with this compiler -- ROCm 7.2 -- and the taps along w then mixed the output channels.)
__device__ __forceinline__ float bf3_rot1(float x, int ctrl_is_ror1) {
    const int xi = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, ctrl_is_ror1 ? __builtin_amdgcn_update_dpp(0, xi, 0x121, 0xF, 0xF, false)
                                                  : __builtin_amdgcn_update_dpp(0, xi, 0x12F, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ f32x4 bf3_rot(f32x4 v) {
    const float a = v.x, b = v.y, c = v.z, d = v.w;
    return (f32x4){bf3_rot1(a, CTRL == 0x121), bf3_rot1(b, CTRL == 0x121), bf3_rot1(c, CTRL == 0x121), bf3_rot1(d, CTRL == 0x121)};
}

// The input is cut into its pieces ONCE, when a plane is staged: the LDS image of a plane is three arrays [cell][16 channels] of
// bf16 (hi, mid, lo: 32 bytes per cell each), and a lane's B operand of a slice is three ds_read_b128.  (First version: fp32
// planes in LDS, the split in registers at every use -- nine times per element, 2.4 vector instructions per MFMA on top of the
// MFMAs' own issue slots: the SIMD's issue port, not the matrix pipe, was the limit -- MFMA pipes 45 % busy, 560 us against 580
// for the fp32-MFMA kernel; profiles/r05_conv3d_c16_bf3.txt.)  92 KB of LDS per workgroup: one workgroup per CU, one wave per
// SIMD with the whole register file -- all 180 weight registers resident, the next slices' operands requested ahead.
constexpr int B3_PART_B = B3_CELLS * 32;           // bytes of one piece array of a plane
constexpr int B3_PLANE_B = 3 * B3_PART_B;          // 30,720 bytes per plane
__global__ __launch_bounds__(256, B3_TWO ? 2 : 1) void conv3d_c16_fwd_bf3_kernel(const float *__restrict__ in, const float *__restrict__ wt,
                                                                   long long s_n, long long s_m, long long s_k, int mirror,
                                                                   float *__restrict__ out, const C16Dims dm) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[3 * B3_PLANE_B];   // plane P in slot (P + 3) % 3
    __shared__ u32x4 wlo[B3_TWO ? 15 * 64 : 1];                                    // B3_TWO: the weights' `lo` pieces, [slice * 3 + kw][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4, t = q >> 1, o = q & 1;   // voxel / output channel, K octet: tap row t of the slice, channels 8o..8o+7
    int b, ty0, tx0, d0, d1;
    const int ob = dm.nob > 1 ? (int)(blockIdx.x % dm.nob) : 0;    // neighbouring workgroups: the same input tile, another output block
    c16_item(dm, dm.nob > 1 ? blockIdx.x / dm.nob : blockIdx.x, b, ty0, tx0, d0, d1);      // (dm.tiles_x counts 30-column tiles here)
    tx0 = (tx0 / TW) * B3_TW;
    ty0 = (ty0 / TH) * B3_TH;
    const size_t plane = (size_t)dm.H * dm.W;
    const float *inb = in + ((size_t)b * dm.D * plane * dm.ip4 + dm.io4) * 4;
    float *outb = out + ((size_t)b * dm.D * plane * dm.op4 + dm.oo4 + 4 * ob) * 4;
    wt += (long long)ob * 16 * s_n;
    // weights: A operand of slice s, tap column kw: rows = output channel n, K = (tap row 2s + t, input channels 8o..8o+7)
    Bf3 wr[5][3];
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int p = 2 * s + t;       // tap row kd * 3 + kh; 9 = the empty half of the last slice
            float w8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = min(p, 8) * 3 + kw;
                const float v = wt[n * s_n + (long long)(8 * o + j) * s_m + (mirror ? 26 - k : k) * s_k];
                w8[j] = p < 9 ? v : 0.f;
            }
            wr[s][kw] = bf3_split(w8);
            if (B3_TWO) {   // (every wave holds the same values; published by the first barrier of the march)
                if (wave == 0) wlo[(s * 3 + kw) * 64 + lane] = wr[s][kw].lo;
                wr[s][kw].lo = (u32x4){0u, 0u, 0u, 0u};
            }
        }
    // staging: the plane's (rows + 2) x 32 cells (rows ty0-1 .., columns tx0-1 ..), zeros outside the volume, through registers; a thread's
    // float4 piece (4 channels of a cell) becomes 8 bytes in each of the three piece arrays
    int ofs[B3_NLD];
    float4 pre[B3_NLD];
#pragma unroll
    for (int i = 0; i < B3_NLD; ++i) {
        const int idx = tid + i * 256, cell = idx >> 2, qq = idx & 3;
        const int yy = ty0 - 1 + cell / B3_XW, xx = tx0 - 1 + cell % B3_XW;
        ofs[i] = (idx < B3_CELLS * 4 && yy >= 0 && yy < dm.H && xx >= 0 && xx < dm.W) ? (yy * dm.W + xx) * dm.ip4 + qq : -1;
    }
    auto fetch = [&](int P) {
        const bool inr = P >= 0 && P < dm.D;
        const float4 *b4 = reinterpret_cast<const float4 *>(inb) + (size_t)(inr ? P : 0) * plane * dm.ip4;
#pragma unroll
        for (int i = 0; i < B3_NLD; ++i) {
            if (inr && ofs[i] >= 0) {
                const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(b4 + ofs[i]));
                pre[i] = make_float4(v[0], v[1], v[2], v[3]);
            } else pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](int P) {
        unsigned char *slot = ring + ((P + 3) % 3) * B3_PLANE_B;
#pragma unroll
        for (int i = 0; i < B3_NLD; ++i) {
            const int idx = tid + i * 256;       // piece idx = cell * 4 + quarter: 8 bytes at cell * 32 + quarter * 8 of each array
            if (idx < B3_CELLS * 4) {
                const float x4[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
                unsigned h[4], m[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned hb = __builtin_bit_cast(unsigned, x4[e]) & 0xffff0000u;
                    const float r1 = x4[e] - __builtin_bit_cast(float, hb);
                    const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
                    const float r2 = r1 - __builtin_bit_cast(float, mb);
                    h[e] = hb; m[e] = mb; l[e] = __builtin_bit_cast(unsigned, r2);
                }
                uint2 *dst = reinterpret_cast<uint2 *>(slot + idx * 8);
                dst[0] = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
                dst[B3_PART_B / 8] = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
                dst[2 * (B3_PART_B / 8)] = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
            }
        }
    };
    fetch(d0 - 1); stash(d0 - 1);
    fetch(d0);     stash(d0);
    fetch(d0 + 1);
    for (int d = d0; d < d1; ++d) {
        stash(d + 1);
        __syncthreads();
        if (d + 1 < d1) fetch(d + 2);
        // this lane's tap row of slice s lives in plane d - 1 + kd, window row (tile row) + kh: byte offset of its cell row's octet
        int rowofs[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int p = min(2 * s + t, 8), kd = p / 3, kh = p - 3 * kd;
            rowofs[s] = ((d + 2 + kd) % 3) * B3_PLANE_B + kh * (B3_XW * 32) + 16 * o;
        }
#pragma unroll 1
        for (int rr = 0; rr < B3_TH / 4; ++rr) {     // this wave's tile rows
            const int row = (B3_TH / 4) * wave + rr;
            f32x4 P[2][3];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) P[g][kw] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // five steps = slices, both groups (input voxels 16 g + n of the window row, u = tx0 - 1 + 16 g + n) in each: six
            // independent accumulators in flight (three left the matrix pipe waiting on its own results: 30 cycles per MFMA
            // whatever else was done); a step's six operand reads are asked for one step ahead
            auto load = [&](int sl, int g) {
                const unsigned char *src = ring + rowofs[sl] + (row * B3_XW + 16 * g + n) * 32;
                Bf3 x;
                x.hi = *reinterpret_cast<const u32x4 *>(src);
                x.mid = *reinterpret_cast<const u32x4 *>(src + B3_PART_B);
                x.lo = *reinterpret_cast<const u32x4 *>(src + 2 * B3_PART_B);
                return x;
            };
            Bf3 xa = load(0, 0), xb = load(0, 1);
#pragma unroll
            for (int sl = 0; sl < 5; ++sl) {
                Bf3 na = xa, nb = xb;
                if (sl + 1 < 5) { na = load(sl + 1, 0); nb = load(sl + 1, 1); }
                __builtin_amdgcn_sched_barrier(0);
                u32x4 wl[3];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) wl[kw] = B3_TWO ? wlo[(sl * 3 + kw) * 64 + lane] : wr[sl][kw].lo;
                // product by product across the six accumulators (smallest terms first)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) { P[0][kw] = bf3_mfma(wl[kw], xa.hi, P[0][kw]); P[1][kw] = bf3_mfma(wl[kw], xb.hi, P[1][kw]); }
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) { P[0][kw] = bf3_mfma(wr[sl][kw].hi, xa.lo, P[0][kw]); P[1][kw] = bf3_mfma(wr[sl][kw].hi, xb.lo, P[1][kw]); }
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) { P[0][kw] = bf3_mfma(wr[sl][kw].mid, xa.mid, P[0][kw]); P[1][kw] = bf3_mfma(wr[sl][kw].mid, xb.mid, P[1][kw]); }
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) { P[0][kw] = bf3_mfma(wr[sl][kw].mid, xa.hi, P[0][kw]); P[1][kw] = bf3_mfma(wr[sl][kw].mid, xb.hi, P[1][kw]); }
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) { P[0][kw] = bf3_mfma(wr[sl][kw].hi, xa.mid, P[0][kw]); P[1][kw] = bf3_mfma(wr[sl][kw].hi, xb.mid, P[1][kw]); }
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) { P[0][kw] = bf3_mfma(wr[sl][kw].hi, xa.hi, P[0][kw]); P[1][kw] = bf3_mfma(wr[sl][kw].hi, xb.hi, P[1][kw]); }
                __builtin_amdgcn_sched_barrier(0);
                xa = na; xb = nb;
            }
            // out[v] = P_kw0[v - 1] + P_kw1[v] + P_kw2[v + 1]; group 0's lanes are v = n - 1 (lane 0: the halo column, no
            // output), group 1's v = 15 + n (lane 15: beyond the tile).  Lane 15 of group 0 takes its right neighbour from
            // group 1's lane 0, lane 0 of group 1 its left neighbour from group 0's lane 15: the wrapped lane of the rotation.
            const f32x4 l0 = bf3_rot<0x121>(P[0][0]), l1 = bf3_rot<0x121>(P[1][0]);   // lane j <- j - 1, lane 0 <- 15
            const f32x4 r0 = bf3_rot<0x12F>(P[0][2]), r1 = bf3_rot<0x12F>(P[1][2]);   // lane j <- j + 1, lane 15 <- 0
            f32x4 rsel = r0, lsel = l1;
            if (n == 15) rsel = r1;
            if (n == 0) lsel = l0;
            const f32x4 o0 = P[0][1] + l0 + rsel;
            const f32x4 o1 = P[1][1] + lsel + r1;
            const int yy = ty0 + row;
            if (yy < dm.H) {
                const int x0 = tx0 + n - 1, x1 = tx0 + 15 + n;
                float4 *o4 = reinterpret_cast<float4 *>(outb) + ((size_t)d * plane + (size_t)yy * dm.W) * dm.op4 + q;
                if (n >= 1 && x0 < dm.W) {
                    float4 *p_ = o4 + (size_t)x0 * dm.op4;
                    float4 v_ = make_float4(o0.x, o0.y, o0.z, o0.w);
                    if (dm.acc) { const float4 a_ = *p_; v_.x += a_.x; v_.y += a_.y; v_.z += a_.z; v_.w += a_.w; }
                    *p_ = v_;
                }
                if (n <= 14 && x1 < dm.W) {
                    float4 *p_ = o4 + (size_t)x1 * dm.op4;
                    float4 v_ = make_float4(o1.x, o1.y, o1.z, o1.w);
                    if (dm.acc) { const float4 a_ = *p_; v_.x += a_.x; v_.y += a_.y; v_.z += a_.z; v_.w += a_.w; }
                    *p_ = v_;
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradient on the bf16 matrix pipe, three-piece operands (channels-last volumes).
//
// dW[tap][ci][co] = sum over voxels of x[v + tap][ci] * gy[v][co]: D[ci][co] += A[ci][32 voxels] . B[32 voxels][co], K = the 32 voxels of
// one tile row.  Both operands want, per lane, EIGHT CONSECUTIVE VOXELS OF ONE CHANNEL -- the transpose of what channels-last memory
// holds:
//   * x: the staged plane is written to LDS transposed and already cut into its pieces: three arrays [16 channels][6 rows][40] of
//     bf16 (channel stride 264 elements = 4 banks mod 64: a 16-lane ds_read_b128 touches every bank once).  A lane's operand for tap
//     column kw starts kw elements into its aligned 8-voxel chunk: one ds_read_b128 + the dword behind it per piece, kw = 2 is the
//     upper four dwords, kw = 1 a funnel shift (v_alignbit) of neighbouring dwords -- per tap row (kd, kh) 6 reads and 12 vector
//     instructions feed 3 taps x 6 = 18 MFMAs;
//   * gy: eight dword loads per lane straight from memory (lane = channel, the wave's 16 channels x 32 voxels are 2 KB contiguous),
//     cut in registers once per tile row, the B operand of all 27 taps.
// A wave owns one of the tile's four rows and keeps the 27 taps' 16 x 16 accumulators (108 registers); 162 MFMAs per row and plane.
// The per-workgroup partial goes out in the layout of the fp32-MFMA kernel above and the same fixed-order fp64 finish adds them.
constexpr int W3_TH = 4, W3_XH = W3_TH + 2, W3_XWC = TW + 2, W3_PITCH = 40, W3_CH = W3_XH * W3_PITCH + 24;   // elements
constexpr int W3_PART_B = CI * W3_CH * 2, W3_PLANE_B = 3 * W3_PART_B;     // 8448 / 25344 bytes
constexpr int W3_NLD = (W3_XH * W3_XWC * 4 + 255) / 256;                  // float4 pieces per thread per plane

__global__ __launch_bounds__(256, 2) void conv3d_c16_bwd_weight_bf3_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                          float *__restrict__ partial, const C16Dims dm) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[3 * W3_PLANE_B];   // plane P in slot (P + 3) % 3; reused for the wave reduction
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = lane & 15, q = lane >> 4;      // operand row / column (a channel), K octet: voxels 8q .. 8q+7 of the tile row
    int b, ty0, tx0, d0, d1;
    // md_conv3d_cb_bwd_weight: every (output block, input block) pair in one launch -- pair blockIdx % nob (nob pairs, ip4 / 4 input
    // blocks), its partials behind those of the pairs before it
    const int pair = dm.nob > 1 ? (int)(blockIdx.x % dm.nob) : 0, item = dm.nob > 1 ? (int)(blockIdx.x / dm.nob) : (int)blockIdx.x;
    const int nib = dm.ip4 / 4, pib = dm.nob > 1 ? pair % nib : 0, pob = dm.nob > 1 ? pair / nib : 0;
    c16_item(dm, item, b, ty0, tx0, d0, d1);
    ty0 = (ty0 / TH) * W3_TH;
    const size_t plane = (size_t)dm.H * dm.W;
    const float *xb = x + ((size_t)b * dm.D * plane * dm.ip4 + dm.io4 + 4 * pib) * 4;      // (ip4 / io4: the x block, op4 / oo4: the gy block -- C16Dims)
    const float *gyb = gy + ((size_t)b * dm.D * plane * dm.op4 + dm.oo4 + 4 * pob) * 4;
    partial += (size_t)pair * (gridDim.x / max(dm.nob, 1)) * NTAP * CI * CO;
    const int gyp = dm.op4 * 4;
    // staging of x: cells (rows ty0-1 .., columns tx0-1 ..) x channel quarters through registers; zeros outside the volume
    int ofs[W3_NLD];
    float4 pre[W3_NLD];
#pragma unroll
    for (int i = 0; i < W3_NLD; ++i) {
        const int idx = tid + i * 256, cell = idx >> 2, qq = idx & 3;
        const int yy = ty0 - 1 + cell / W3_XWC, xx = tx0 - 1 + cell % W3_XWC;
        ofs[i] = (idx < W3_XH * W3_XWC * 4 && yy >= 0 && yy < dm.H && xx >= 0 && xx < dm.W) ? (yy * dm.W + xx) * dm.ip4 + qq : -1;
    }
    auto fetch = [&](int P) {
        const bool inr = P >= 0 && P < dm.D;
        const float4 *b4 = reinterpret_cast<const float4 *>(xb) + (size_t)(inr ? P : 0) * plane * dm.ip4;
#pragma unroll
        for (int i = 0; i < W3_NLD; ++i) {
            if (inr && ofs[i] >= 0) {
                const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(b4 + ofs[i]));
                pre[i] = make_float4(v[0], v[1], v[2], v[3]);
            } else pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](int P) {
        unsigned short *slot = reinterpret_cast<unsigned short *>(ring + ((P + 3) % 3) * W3_PLANE_B);
#pragma unroll
        for (int i = 0; i < W3_NLD; ++i) {
            const int idx = tid + i * 256, cell = idx >> 2, qq = idx & 3;
            if (idx < W3_XH * W3_XWC * 4) {
                const int e0 = (4 * qq) * W3_CH + (cell / W3_XWC) * W3_PITCH + cell % W3_XWC;   // element of channel 4 qq in a piece array
                const float x4[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned hb = __builtin_bit_cast(unsigned, x4[e]) & 0xffff0000u;
                    const float r1 = x4[e] - __builtin_bit_cast(float, hb);
                    const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
                    const float r2 = r1 - __builtin_bit_cast(float, mb);
                    slot[e0 + e * W3_CH] = (unsigned short)(hb >> 16);
                    slot[W3_PART_B / 2 + e0 + e * W3_CH] = (unsigned short)(mb >> 16);
                    slot[W3_PART_B + e0 + e * W3_CH] = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
                }
            }
        }
    };
    f32x4 acc[NTAP];
#pragma unroll
    for (int k = 0; k < NTAP; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};

    fetch(d0 - 1); stash(d0 - 1);
    fetch(d0);     stash(d0);
    fetch(d0 + 1);
    const int yy = ty0 + wave;          // this wave's tile row
    for (int d = d0; d < d1; ++d) {
        // the row's gy operand: voxels tx0 + 8q .. + 7 of channel `ch`, requested before the barrier
        float g8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int xx = tx0 + 8 * q + j;
            g8[j] = (yy < dm.H && xx < dm.W) ? __builtin_nontemporal_load(gyb + ((size_t)d * plane + (size_t)yy * dm.W + xx) * gyp + ch) : 0.f;
        }
        stash(d + 1);
        __syncthreads();
        if (d + 1 < d1) fetch(d + 2);
        const Bf3 gs = bf3_split(g8);
        // nine tap rows (kd, kh); a row's raw reads: per piece the aligned 8-voxel chunk of channel `ch` at window row wave + kh,
        // column 8q, and the dword behind it.  (Asked for one row ahead and pinned there with sched_barrier: 544 against 528 us --
        // two waves per SIMD cover the LDS round trips by themselves here, the pinning only costs registers.)
        struct Raw { u32x4 a[3]; unsigned t[3]; };
        auto rload = [&](int p) {
            const int kd = p / 3, kh = p - 3 * kd;
            const unsigned char *src = ring + ((d + 2 + kd) % 3) * W3_PLANE_B + (ch * W3_CH + (wave + kh) * W3_PITCH + 8 * q) * 2;
            Raw r;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                r.a[pc] = *reinterpret_cast<const u32x4 *>(src + pc * W3_PART_B);
                r.t[pc] = *reinterpret_cast<const unsigned *>(src + pc * W3_PART_B + 16);
            }
            return r;
        };
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            const Raw cur = rload(p);
            // tap columns kw = 0, 1, 2: voxels 8q + kw .. + 7 of the window row -- the chunk, a funnel shift, the upper dwords
            u32x4 x0[3], x1[3], x2[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                const u32x4 a = cur.a[pc];
                const unsigned a4 = cur.t[pc];
                x0[pc] = a;
                x1[pc] = (u32x4){__builtin_amdgcn_alignbit(a[1], a[0], 16), __builtin_amdgcn_alignbit(a[2], a[1], 16),
                                 __builtin_amdgcn_alignbit(a[3], a[2], 16), __builtin_amdgcn_alignbit(a4, a[3], 16)};
                x2[pc] = (u32x4){a[1], a[2], a[3], a4};
            }
            const int k0 = p * 3;
            // A = x (rows = input channel), B = gy (columns = output channel); six products per tap (pieces 0 / 1 / 2 = hi / mid / lo),
            // the three taps interleaved, smallest terms first
            acc[k0] = bf3_mfma(x0[2], gs.hi, acc[k0]); acc[k0 + 1] = bf3_mfma(x1[2], gs.hi, acc[k0 + 1]); acc[k0 + 2] = bf3_mfma(x2[2], gs.hi, acc[k0 + 2]);
            acc[k0] = bf3_mfma(x0[0], gs.lo, acc[k0]); acc[k0 + 1] = bf3_mfma(x1[0], gs.lo, acc[k0 + 1]); acc[k0 + 2] = bf3_mfma(x2[0], gs.lo, acc[k0 + 2]);
            acc[k0] = bf3_mfma(x0[1], gs.mid, acc[k0]); acc[k0 + 1] = bf3_mfma(x1[1], gs.mid, acc[k0 + 1]); acc[k0 + 2] = bf3_mfma(x2[1], gs.mid, acc[k0 + 2]);
            acc[k0] = bf3_mfma(x0[1], gs.hi, acc[k0]); acc[k0 + 1] = bf3_mfma(x1[1], gs.hi, acc[k0 + 1]); acc[k0 + 2] = bf3_mfma(x2[1], gs.hi, acc[k0 + 2]);
            acc[k0] = bf3_mfma(x0[0], gs.mid, acc[k0]); acc[k0 + 1] = bf3_mfma(x1[0], gs.mid, acc[k0 + 1]); acc[k0 + 2] = bf3_mfma(x2[0], gs.mid, acc[k0 + 2]);
            acc[k0] = bf3_mfma(x0[0], gs.hi, acc[k0]); acc[k0 + 1] = bf3_mfma(x1[0], gs.hi, acc[k0 + 1]); acc[k0 + 2] = bf3_mfma(x2[0], gs.hi, acc[k0 + 2]);
        }
        __syncthreads();  // slot (d+2)%3 == (d-1)%3 is rewritten at the top of the next step
    }
    // sum the 4 waves through LDS (two waves' accumulators fit at a time), wave 0 writes the workgroup's partial
    f32x4 *red = reinterpret_cast<f32x4 *>(ring);  // [2][NTAP][64]
    if (wave >= 2) {
#pragma unroll
        for (int k = 0; k < NTAP; ++k) red[((wave - 2) * NTAP + k) * 64 + lane] = acc[k];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int k = 0; k < NTAP; ++k) acc[k] += red[(wave * NTAP + k) * 64 + lane];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int k = 0; k < NTAP; ++k) red[k * 64 + lane] = acc[k];
    }
    __syncthreads();
    if (wave == 0) {
        f32x4 *out = reinterpret_cast<f32x4 *>(partial) + (size_t)item * NTAP * 64;
#pragma unroll
        for (int k = 0; k < NTAP; ++k) out[k * 64 + lane] = acc[k] + red[k * 64 + lane];
    }
}

int c16_dims(const char *fn, int B, int Ci, int Co, int D, int H, int W, C16Dims &dm) {
    MD_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "%s: bad dims B=%d D=%d H=%d W=%d", fn, B, D, H, W);
    MD_REQUIRE(Ci == CI && Co == CO, "%s: %d -> %d channels unsupported (16 -> 16 only)", fn, Ci, Co);
    MD_REQUIRE((long long)D * H * W * CI < (1ll << 31), "%s: one sample must stay below 2^31 elements", fn);
    dm.B = B; dm.D = D; dm.H = H; dm.W = W;
    dm.ip4 = 4; dm.io4 = 0; dm.op4 = 4; dm.oo4 = 0; dm.acc = 0; dm.nob = 1;
    dm.tiles_x = md_cdiv(W, TW);
    dm.tiles = dm.tiles_x * md_cdiv(H, TH);
    // two workgroups fit a CU (65 KB of LDS each): at least ~3 per CU in total for balance, >= 8 planes per slice
    int ds = 1;
    while (ds * 2 <= D / 8 && (long long)B * dm.tiles * ds < 700) ds *= 2;
    // Swept at 6x96x48x160 (fwd / data gradient / weight gradient, us): 4 slices (this default, 720 workgroups) 628 / 629 /
    // 701; 6: 656 / 650 / 773; 8: 630 / 624 / 695; 12: 645 / 618 / 788; 16: 663 / 635 / 842 -- the ~60 % of peak is not a
    // tail effect of 720 workgroups on 512 slots.
    if (MD_C16_DSLICES >= 1 && MD_C16_DSLICES <= D) ds = MD_C16_DSLICES;  // A/B builds: planes per slice = ceil(D / value)
    dm.planes = md_cdiv(D, ds);
    dm.dslices = md_cdiv(D, dm.planes);
    return MD_OK;
}

}  // namespace


extern "C" {

static int c16_launch_fwd(const char *fn, const float *in, const float *wt, long long s_n, long long s_m, long long s_k, int mirror,
                          int in_planar, int out_planar, float *out, int B, int Ci, int Co, int D, int H, int W,
                          md_stream_t stream) {
    MD_REQUIRE(in && wt && out, "%s: null tensor argument", fn);
    MD_REQUIRE(in_planar || ((uintptr_t)in % 16) == 0, "%s: a channels-last input volume must be 16-byte aligned", fn);
    C16Dims dm;
    if (int rc = c16_dims(fn, B, Ci, Co, D, H, W, dm)) return rc;
    // channels-last in and out (what the trainer runs): the bf16 x 3 kernel, 30-column tiles; -DMD_C16_BF3=0: the fp32-MFMA kernel (A/B builds)
    constexpr bool bf3 = MD_C16_BF3 != 0;
    if (bf3 && !in_planar && !out_planar && ((uintptr_t)out % 16) == 0) {
        C16Dims d3 = dm;
        d3.tiles_x = md_cdiv(W, B3_TW);
        d3.tiles = d3.tiles_x * md_cdiv(H, B3_TH);
        // D slices as for the fp32-MFMA kernels (c16_dims: ~3 workgroups per slot, >= 8 planes each).  Swept at 6x96x48x160
        // (forward / data gradient, us): 4 slices 492 / 455, 6: 537 / 503, 8: 549 / 481, 12: 597 / 502.
        const dim3 grid3(B * d3.tiles * d3.dslices), block3(256);
        MD_LAUNCH_TIMED(fn, conv3d_c16_fwd_bf3_kernel, grid3, block3, 0, (hipStream_t)stream, in, wt, s_n, s_m, s_k, mirror, out, d3);
        MD_CHECK_LAUNCH(fn);
        return MD_OK;
    }
    const dim3 grid(B * dm.tiles * dm.dslices), block(256);
    hipStream_t s = (hipStream_t)stream;
#define MD_C16_FWD(IP, OP) MD_LAUNCH_TIMED(fn, (conv3d_c16_fwd_kernel<IP, OP>), grid, block, 0, s, in, wt, s_n, s_m, s_k, mirror, out, dm)
    if (in_planar) { if (out_planar) MD_C16_FWD(true, true); else MD_C16_FWD(true, false); }
    else { if (out_planar) MD_C16_FWD(false, true); else MD_C16_FWD(false, false); }
#undef MD_C16_FWD
    MD_CHECK_LAUNCH(fn);
    return MD_OK;
}

int md_conv3d_c16_fwd(const float *x, int x_planar, const float *wt, long long w_stride_co, long long w_stride_ci,
                      long long w_stride_k, float *y, int B, int Ci, int Co, int D, int H, int W, md_stream_t stream) {
    return c16_launch_fwd("md_conv3d_c16_fwd", x, wt, w_stride_co, w_stride_ci, w_stride_k, 0, x_planar, 0, y, B, Ci, Co, D, H,
                          W, stream);
}

int md_conv3d_c16_bwd_data(const float *gy, const float *wt, long long w_stride_co, long long w_stride_ci,
                           long long w_stride_k, float *dx, int dx_planar, int B, int Ci, int Co, int D, int H, int W,
                           md_stream_t stream) {
    return c16_launch_fwd("md_conv3d_c16_bwd_data", gy, wt, w_stride_ci, w_stride_co, w_stride_k, 1, 0, dx_planar, dx, B, Ci, Co,
                          D, H, W, stream);
}

// the bf16 x 3 weight gradient's tiles are 4 rows high: twice the workgroups (and partials) of the fp32-MFMA kernel's
static void c16_dims_w3(C16Dims &dm, int H) { dm.tiles = dm.tiles_x * md_cdiv(H, W3_TH); }
static constexpr bool c16_w3_on() { return MD_C16_BF3_WGRAD != 0; }

size_t md_conv3d_c16_bwd_weight_ws_bytes(int B, int D, int H, int W) {
    C16Dims dm;
    if (c16_dims("md_conv3d_c16_bwd_weight_ws_bytes", B, CI, CO, D, H, W, dm)) return 0;
    c16_dims_w3(dm, H);     // the larger of the two kernels' needs
    return (size_t)B * dm.tiles * dm.dslices * NTAP * CI * CO * sizeof(float);
}

int md_conv3d_c16_bwd_weight(const float *x, int x_planar, const float *gy, float *dwt, long long dw_stride_co,
                             long long dw_stride_ci, long long dw_stride_k, void *ws, size_t ws_bytes, int B, int Ci, int Co,
                             int D, int H, int W, md_stream_t stream) {
    MD_REQUIRE(x && gy && dwt && ws, "md_conv3d_c16_bwd_weight: null tensor argument");
    MD_REQUIRE((x_planar || ((uintptr_t)x % 16) == 0) && ((uintptr_t)ws % 16) == 0,
               "md_conv3d_c16_bwd_weight: a channels-last x and ws must be 16-byte aligned");
    C16Dims dm;
    if (int rc = c16_dims("md_conv3d_c16_bwd_weight", B, Ci, Co, D, H, W, dm)) return rc;
    const bool w3 = c16_w3_on() && !x_planar;   // channels-last x (what the trainer runs): the bf16 x 3 kernel; MD_C16_BF3_WGRAD=0: fp32 MFMA
    if (w3) c16_dims_w3(dm, H);
    const int nwg = B * dm.tiles * dm.dslices;
    MD_REQUIRE(ws_bytes >= (size_t)nwg * NTAP * CI * CO * sizeof(float), "md_conv3d_c16_bwd_weight: workspace too small (%zu bytes)", ws_bytes);
    hipStream_t s = (hipStream_t)stream;
    float *partial = (float *)ws;
    if (w3) MD_LAUNCH_TIMED("md_conv3d_c16_bwd_weight", conv3d_c16_bwd_weight_bf3_kernel, dim3(nwg), dim3(256), 0, s, x, gy, partial, dm);
    else if (x_planar) MD_LAUNCH_TIMED("md_conv3d_c16_bwd_weight", conv3d_c16_bwd_weight_kernel<true>, dim3(nwg), dim3(256), 0, s, x, gy, partial, dm);
    else MD_LAUNCH_TIMED("md_conv3d_c16_bwd_weight", conv3d_c16_bwd_weight_kernel<false>, dim3(nwg), dim3(256), 0, s, x, gy, partial, dm);
    MD_CHECK_LAUNCH("md_conv3d_c16_bwd_weight");
    hipLaunchKernelGGL(conv3d_c16_bwd_weight_finish_kernel, dim3(NTAP * 256 / 16), dim3(256), 0, s, partial, nwg, dw_stride_co,
                       dw_stride_ci, dw_stride_k, dwt);
    MD_CHECK_LAUNCH("md_conv3d_c16_bwd_weight(finish)");
    return MD_OK;
}

// ---- Ci, Co multiples of 16: the regulariser's interior 3 x 3 x 3 layers (32 -> 32 ...) as sums over 16 x 16 channel blocks on the bf16 x 3
// kernels above (C16Dims::ip4 ...): Co/16 x Ci/16 launches per direction, the second and later input blocks of an output block added in
// the epilogue.  Channels-last volumes only.
static int cb_check(const char *fn, int Ci, int Co) {
    MD_REQUIRE(Ci >= 16 && Co >= 16 && Ci % 16 == 0 && Co % 16 == 0 && Ci <= 256 && Co <= 256, "%s: Ci and Co must be multiples of 16 (<= 256), got %d -> %d", fn, Ci, Co);
    return MD_OK;
}

static int cb_launch_fwd(const char *fn, const float *in, int Cin, const float *wt, long long s_n, long long s_m, long long s_k, int mirror,
                         float *out, int Cout, int B, int D, int H, int W, md_stream_t stream) {
    MD_REQUIRE(in && wt && out, "%s: null tensor argument", fn);
    MD_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0, "%s: volumes must be 16-byte aligned", fn);
    C16Dims dm;
    if (int rc = c16_dims(fn, B, CI, CO, D, H, W, dm)) return rc;
    dm.tiles_x = md_cdiv(W, B3_TW);
    dm.tiles = dm.tiles_x * md_cdiv(H, B3_TH);
    // one launch per INPUT block, covering every output block (the workgroups of a tile's output blocks are neighbours: the tile's
    // input planes come from L2 after the first); the later input blocks add to what the earlier ones stored
    const int nob = Cout / 16;
    const dim3 grid(B * dm.tiles * dm.dslices * nob), block(256);
    for (int ib = 0; ib < Cin / 16; ++ib) {
        C16Dims d = dm;
        d.ip4 = Cin / 4; d.io4 = ib * 4; d.op4 = Cout / 4; d.oo4 = 0; d.acc = ib > 0; d.nob = nob;
        const float *wp = wt + (long long)ib * 16 * s_m;
        MD_LAUNCH_TIMED(fn, conv3d_c16_fwd_bf3_kernel, grid, block, 0, (hipStream_t)stream, in, wp, s_n, s_m, s_k, mirror, out, d);
        MD_CHECK_LAUNCH(fn);
    }
    return MD_OK;
}

int md_conv3d_cb_fwd(const float *x, const float *wt, long long w_stride_co, long long w_stride_ci, long long w_stride_k, float *y,
                     int B, int Ci, int Co, int D, int H, int W, md_stream_t stream) {
    if (int rc = cb_check("md_conv3d_cb_fwd", Ci, Co)) return rc;
    return cb_launch_fwd("md_conv3d_cb_fwd", x, Ci, wt, w_stride_co, w_stride_ci, w_stride_k, 0, y, Co, B, D, H, W, stream);
}

int md_conv3d_cb_bwd_data(const float *gy, const float *wt, long long w_stride_co, long long w_stride_ci, long long w_stride_k,
                          float *dx, int B, int Ci, int Co, int D, int H, int W, md_stream_t stream) {
    if (int rc = cb_check("md_conv3d_cb_bwd_data", Ci, Co)) return rc;
    // the same kernel with the taps mirrored and the weight's channel roles swapped: its "output" channels are the layer's inputs
    return cb_launch_fwd("md_conv3d_cb_bwd_data", gy, Co, wt, w_stride_ci, w_stride_co, w_stride_k, 1, dx, Ci, B, D, H, W, stream);
}

size_t md_conv3d_cb_bwd_weight_ws_bytes(int B, int Ci, int Co, int D, int H, int W) {
    if (Ci < 16 || Co < 16 || Ci % 16 || Co % 16) return 0;
    return md_conv3d_c16_bwd_weight_ws_bytes(B, D, H, W) * (size_t)(Ci / 16) * (size_t)(Co / 16);
}

int md_conv3d_cb_bwd_weight(const float *x, const float *gy, float *dwt, long long dw_stride_co, long long dw_stride_ci,
                            long long dw_stride_k, void *ws, size_t ws_bytes, int B, int Ci, int Co, int D, int H, int W,
                            md_stream_t stream) {
    if (int rc = cb_check("md_conv3d_cb_bwd_weight", Ci, Co)) return rc;
    MD_REQUIRE(x && gy && dwt && ws, "md_conv3d_cb_bwd_weight: null tensor argument");
    MD_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)gy % 16) == 0 && ((uintptr_t)ws % 16) == 0, "md_conv3d_cb_bwd_weight: x, gy and ws must be 16-byte aligned");
    C16Dims dm;
    if (int rc = c16_dims("md_conv3d_cb_bwd_weight", B, CI, CO, D, H, W, dm)) return rc;
    c16_dims_w3(dm, H);
    const int nwg = B * dm.tiles * dm.dslices, nib = Ci / 16, npair = nib * (Co / 16);
    const size_t per_pair = (size_t)nwg * NTAP * CI * CO;
    MD_REQUIRE(ws_bytes >= per_pair * npair * sizeof(float), "md_conv3d_cb_bwd_weight: workspace too small (%zu bytes)", ws_bytes);
    hipStream_t s = (hipStream_t)stream;
    float *partial = (float *)ws;
    // every (output block, input block) pair in ONE launch (a tile's pairs are neighbouring workgroups), then a finish per pair
    C16Dims d = dm;
    d.ip4 = Ci / 4; d.io4 = 0; d.op4 = Co / 4; d.oo4 = 0; d.nob = npair;
    MD_LAUNCH_TIMED("md_conv3d_cb_bwd_weight", conv3d_c16_bwd_weight_bf3_kernel, dim3(nwg * npair), dim3(256), 0, s, x, gy, partial, d);
    MD_CHECK_LAUNCH("md_conv3d_cb_bwd_weight");
    for (int pair = 0; pair < npair; ++pair) {
        const int ib = pair % nib, ob = pair / nib;
        hipLaunchKernelGGL(conv3d_c16_bwd_weight_finish_kernel, dim3(NTAP * 256 / 16), dim3(256), 0, s, partial + per_pair * pair, nwg, dw_stride_co,
                           dw_stride_ci, dw_stride_k, dwt + (long long)ob * 16 * dw_stride_co + (long long)ib * 16 * dw_stride_ci);
        MD_CHECK_LAUNCH("md_conv3d_cb_bwd_weight(finish)");
    }
    return MD_OK;
}

}  // extern "C"
