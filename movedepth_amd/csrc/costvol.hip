// Plane-sweep cost volume, forward and backward, for gfx950 (MI355X).
//
// Replaces generate_costvol (reference layers.py:778-794) + the group mean of trainer.py:359.
// The reference materialises (B,D,C,h,w) through repeat -> matmul x3 -> grid_sample -> mul -> stack
// (~4.8 GB of traffic at B=6, 48x160, D=96) and then reduces it; here one kernel writes the grouped
// (B,D,G,h,w) volume once: algorithmic bytes = ref + src + hypotheses (or prior) + volume.
//
// Work decomposition (HBM-write-bound kernel, no MFMA: ~300 flop per 64 output bytes)
//   workgroup = 256 threads = one TW x TH pixel tile of one sample, GS groups (CPW = GS*N channels)
//               and one slice of the D hypotheses;  grid = (tiles, G/GS, B*DSPLIT).
//   thread    = one reference pixel; it walks its D slice, so the hypotheses of a pixel ("per-pixel
//               depth hypotheses") never leave registers and each wave store is a 128/256-byte row piece.
//               With the fused schedule the per-bin interval positions k/(D-1) sit in a small LDS table.
//   LDS       = the source-feature window the tile's epipolar segments can reach, staged ONCE per
//               workgroup channels-last ([pixel][CPW] as 16-byte quads, XOR-swizzled so a wave's
//               ds_read_b128 taps are bank-conflict free), zero-filled outside the image: that is
//               grid_sample's 'zeros' padding for free.  Taps are 4 x CPW/4 ds_read_b128 per hypothesis
//               instead of 4 x CPW scattered global loads.  The window origin comes from the tile's
//               bounding box of tap positions at the first and last hypothesis of the slice; taps that
//               still fall outside (wild poses) take a slow, correct global-memory path (out of line).
//   backward  = same walk; d_ref accumulates in registers.  d_src contributions accumulate in registers
//               while consecutive hypotheses land in the same source cell (an epipolar segment spans
//               ~1-2 px over the depth range) and are scattered with ds_add_f32 into a planar LDS window
//               only when the cell changes -- LDS float atomics are ~50x slower than integer ones on gfx950
//               (measured: 3.1 ms vs 0.22 ms for the same scatter), so they must be rare -- then flushed
//               once per workgroup with global float atomics.
#include <limits.h>

#include <atomic>

#include "md_common.hpp"

// Element type of the feature maps (ref, src) and of the volume (out, gout); everything in between is fp32.
// MD_CV_IO = 0: float (the default build, entry points md_costvol_fwd / md_costvol_bwd);
//            1: bf16, 2: fp16 -- the Makefile compiles this file again with these (entry points *_bf16 / *_f16),
//               BASELINE configs 4 and 5: half the volume bytes, same arithmetic.  d_ref / d_src stay fp32.
#ifndef MD_CV_IO
#define MD_CV_IO 0
#endif
#if MD_CV_IO == 1
#include <hip/hip_bf16.h>
typedef __hip_bfloat16 io_t;
typedef uint16_t abi_io_t;  // the ABI carries bit patterns, no vendor types
#define MD_CV_NAME(x) x##_bf16
#elif MD_CV_IO == 2
#include <hip/hip_fp16.h>
typedef __half io_t;
typedef uint16_t abi_io_t;
#define MD_CV_NAME(x) x##_f16
#else
typedef float io_t;
typedef float abi_io_t;
#define MD_CV_NAME(x) x
#endif
#define MD_CV_STR2(x) #x
#define MD_CV_STR(x) MD_CV_STR2(x)

namespace {

#if MD_CV_IO == 1
__device__ __forceinline__ float ldio(const io_t *p) { return __bfloat162float(*p); }
__device__ __forceinline__ void stio(io_t *p, float v) { *p = __float2bfloat16(v); }
#elif MD_CV_IO == 2
__device__ __forceinline__ float ldio(const io_t *p) { return __half2float(*p); }
__device__ __forceinline__ void stio(io_t *p, float v) { *p = __float2half(v); }
#else
// plain expressions, not functions, in the fp32 build: through a (non-restrict) function parameter the compiler scheduled
// the planar forward kernel 5 % slower
#define ldio(...) (*(__VA_ARGS__))
#define stio(p, v) (*(p) = (v))
#endif
// four consecutive volume elements (p 16-byte aligned for float, 8-byte for the half types)
#if MD_CV_IO == 0
#define stio4(p, v) (*reinterpret_cast<float4 *>(p) = (v))
#else
__device__ __forceinline__ void stio4(io_t *p, float4 v) {
    io_t t[4];
    stio(t, v.x); stio(t + 1, v.y); stio(t + 2, v.z); stio(t + 3, v.w);
    *reinterpret_cast<uint2 *>(p) = *reinterpret_cast<const uint2 *>(t);
}
#endif

// true: follow Project3D's operation order step by step (P @ (depth * ray), /(w-1), -0.5, *2, then grid_sample's
// un-normalise); false: the algebraically identical 3-FMA form.  The two differ by a few ulp of the pixel coordinate
// (<= ~2e-5 px at 160 px), the same order as the reference's own CPU-vs-CPU coordinate noise (SURVEY KAT1: 5e-5).
#ifndef MD_COSTVOL_REFERENCE_OP_ORDER
#define MD_COSTVOL_REFERENCE_OP_ORDER 0
#endif
constexpr bool kReferenceOpOrder = MD_COSTVOL_REFERENCE_OP_ORDER != 0;

constexpr int ITV_MAX = 256;  // hypotheses per D slice served from the LDS interval table

struct CvDims {
    float scale_fac;
    int sched_type;
    int B, C, G, h, w, D;
    int tiles_x, tiles, splits;  // pixel tiles per sample (x, total) and channel splits
    int items;                   // B * tiles * splits work items of D hypotheses each
    int dbg;                     // unused
    // channels-last kernels, two-phase schedule (0 = plain linear shares): workgroups [0, nc) take one of the items' k1 equal
    // slices each, workgroups >= nc take 1/fsub of one of the remaining slices (see launch_cl_inst)
    int k1, nc, fsub;
    unsigned long long *stats;   // md_costvol_stats: per-launch counters (null: off)
    unsigned long long *census;  // backward: this launch's census word (gathered steps / 4 and steps / 4: 18 bits each; windows staged, segments: 14 bits each), or null
    const long long *shares;     // backward: the caller's partition, [lo, hi) of the items x D steps per workgroup (null: the library's)
    unsigned *cost;              // backward: per item, the shader cycles the workgroups that walked it took (null: not recorded)
    int fcl;                     // feature maps and their gradients channels-last [B,h,w,C] (channels-last kernels only)
    int xcd_map;                 // item-aligned slices: contiguous ranges per XCD (cv_share)
    float gslack0;               // fcl: ... and by this factor already at the first (whole-slice) attempt: no halving
    float gslack;                // fcl: a sub-slice whose footprint exceeds the window by this factor is gathered from L2 (cl_stage)
    int shape0;                  // backward: which of the window shapes the fit test tries first (ClShapes, costvol_cl.inc)
    int min_sub;                 // fcl: shortest hypothesis sub-slice a window is staged for; what does not fit then is gathered from L2
    long long sb, sd, sg, sp;
};

// Window = tile + epipolar reach.  The reach shrinks when a workgroup carries 32 channels (128 B per pixel)
// so that two workgroups still fit a CU's 160 KB of LDS.
template <int TW, int CPW>
struct Tile {
    static constexpr int TH = 256 / TW;
    static constexpr int WW = TW + (CPW >= 32 ? 8 : 16);
    static constexpr int WH = TH + (CPW >= 32 ? 4 : 8);
    static constexpr int WP = WW * WH;
};

template <int CPW>
using vecf = float __attribute__((ext_vector_type(CPW)));
using v2f = float __attribute__((ext_vector_type(2)));


// Channel slots of a workgroup (GS groups x N channels each, group j = channels {gbase+j + i*G}).
// Quad map (GS % 4 == 0): 16-byte quad q = m*N + i holds channel i of groups 4m..4m+3, so a group-quad's outputs
// are N packed multiply-adds of whole quads and come out as 4 consecutive groups in aligned registers (no
// shuffling before the wide store).  Otherwise slot k = j*N + i.
template <int GS, int N>
struct Slots {
    static constexpr bool QUADMAP = GS % 4 == 0;
    __device__ __forceinline__ static constexpr int group(int k) { return QUADMAP ? 4 * ((k / 4) / N) + k % 4 : k / N; }
    __device__ __forceinline__ static constexpr int chan_in_group(int k) { return QUADMAP ? (k / 4) % N : k % N; }
    __device__ __forceinline__ static constexpr int slot(int j, int i) { return QUADMAP ? ((j / 4) * N + i) * 4 + j % 4 : j * N + i; }
    __device__ __forceinline__ static int channel(int k, int G, int gbase) { return chan_in_group(k) * G + gbase + group(k); }
};

// A segment = hypotheses [d0, d1) of one work item (sample, pixel tile, channel split).  The grid is sized to
// the chip's workgroup slots and every workgroup takes an equal, contiguous share of the items x D hypothesis
// steps (a workgroup's share may span two items): perfectly balanced for any B, image size and D, no tail round.
struct Seg {
    int b, tile, gs, d0, d1, item;
};
// This workgroup's share [lo, hi) of the items x D hypothesis steps.
__device__ __forceinline__ void cv_share(const CvDims &dm, long long &lo, long long &hi) {
    if (dm.shares) { lo = dm.shares[2 * blockIdx.x]; hi = dm.shares[2 * blockIdx.x + 1]; return; }
    const long long total = (long long)dm.items * dm.D;
    if (dm.k1 <= 0) {
        lo = total * blockIdx.x / gridDim.x;
        hi = total * (blockIdx.x + 1) / gridDim.x;
        return;
    }
    int bid = blockIdx.x;
    if (dm.xcd_map && bid < dm.nc) {
        // workgroup b runs on XCD b % 8 (observed): give every XCD a CONTIGUOUS eighth of the slices, i.e. neighbouring tiles of
        // one sample, so that the d_src lines its atomics touch (and the source lines its windows / gathers read) stay in that
        // XCD's L2 instead of bouncing between the eight of them
        const int x = bid & 7, n8 = dm.nc >> 3, r = dm.nc & 7;
        bid = x * n8 + min(x, r) + (bid >> 3);
    }
    const int sl = bid < dm.nc ? bid : dm.nc + (bid - dm.nc) / dm.fsub;  // coarse slice
    const int item = sl / dm.k1, part = sl - item * dm.k1;
    const long long l0 = (long long)item * dm.D + (long long)part * dm.D / dm.k1;
    const long long l1 = (long long)item * dm.D + (long long)(part + 1) * dm.D / dm.k1;
    if (bid < dm.nc) {
        lo = l0;
        hi = l1;
    } else {
        const int sub = ((int)blockIdx.x - dm.nc) % dm.fsub;
        lo = l0 + (l1 - l0) * sub / dm.fsub;
        hi = l0 + (l1 - l0) * (sub + 1) / dm.fsub;
    }
}

__device__ __forceinline__ bool next_segment(const CvDims &dm, long long &lo, long long hi, Seg &sg) {
    if (lo >= hi) return false;
    const int item = (int)(lo / dm.D);
    sg.item = item;
    sg.d0 = (int)(lo % dm.D);
    const long long left = hi - lo;
    sg.d1 = (long long)sg.d0 + left < dm.D ? sg.d0 + (int)left : dm.D;
    sg.gs = item % dm.splits;
    sg.tile = (item / dm.splits) % dm.tiles;
    sg.b = item / (dm.splits * dm.tiles);
    lo += sg.d1 - sg.d0;
    return true;
}

struct Tap4 {
    int x0, y0;
    float w00, w01, w10, w11;
};

// Per-thread state of the hypothesis walk.
template <bool FUSED>
struct Walk {
    CamMats cam;
    float r0, r1, r2, wm1, hm1, rw, rh;
    float A0, A1, A2, B0, B1, B2;  // c_i(dep) = dep * A_i + B_i
    HypConst hc;
    const float *hyp_p;  // !FUSED: &hyp[b, 0, y, x]
    size_t hyp_stride;   // h*w
    int sched_type, d0;

    __device__ __forceinline__ float hypothesis(const float *itv, int k) const {
        if (!FUSED) return hyp_p[(size_t)k * hyp_stride];
        return md_hyp_eval(hc, itv[k - d0], sched_type);
    }
    // tap range allowing for rounding of the position (see ClWalk::tap_range in costvol_cl.inc)
    __device__ __forceinline__ void tap_range(float dep, int &x_lo, int &x_hi, int &y_lo, int &y_hi) const {
        float ix, iy;
        if (kReferenceOpOrder) {
            md_project_fast(cam, r0, r1, r2, dep, wm1, hm1, rw, rh, ix, iy);
        } else {
            const float rz = md_rcp_nr(fmaf(dep, A2, B2));
            ix = fmaf(dep, A0, B0) * rz;
            iy = fmaf(dep, A1, B1) * rz;
        }
        const float ex = 1e-3f + 4e-6f * fabsf(ix), ey = 1e-3f + 4e-6f * fabsf(iy);
        const bool ok = ix == ix && iy == iy;
        x_lo = ok ? (int)fminf(fmaxf(floorf(ix - ex), -1e9f), 1e9f) : INT_MAX;
        x_hi = ok ? (int)fminf(fmaxf(floorf(ix + ex), -1e9f), 1e9f) : INT_MIN;
        y_lo = ok ? (int)fminf(fmaxf(floorf(iy - ey), -1e9f), 1e9f) : INT_MAX;
        y_hi = ok ? (int)fminf(fmaxf(floorf(iy + ey), -1e9f), 1e9f) : INT_MIN;
    }
    __device__ __forceinline__ Tap4 tap_at(float dep) const {
        float ix, iy;
        if (kReferenceOpOrder) {
            md_project_fast(cam, r0, r1, r2, dep, wm1, hm1, rw, rh, ix, iy);
        } else {
            // projective-linear in depth: c = dep * (P[:, :3] @ ray) + P[:, 3]; the [-1,1] normalise / un-normalise
            // round trip of Project3D + grid_sample is the identity and is dropped
            const float rz = md_rcp_nr(fmaf(dep, A2, B2));
            ix = fmaf(dep, A0, B0) * rz;
            iy = fmaf(dep, A1, B1) * rz;
        }
        float fx = floorf(ix), fy = floorf(iy);
        const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        // NaN -> far outside; the float->int conversion saturates for +-inf / huge values
        fx = (ix == ix) ? fx : -1e9f;
        fy = (iy == iy) ? fy : -1e9f;
        Tap4 t;
        t.x0 = (int)fminf(fmaxf(fx, -1e9f), 1e9f);
        t.y0 = (int)fminf(fmaxf(fy, -1e9f), 1e9f);
        t.w00 = wy0 * wx0; t.w01 = wy0 * wx1; t.w10 = wy1 * wx0; t.w11 = wy1 * wx1;
        return t;
    }
};

// Tap outside the staged window: zero when all four taps miss the image, else bounds-checked global loads.
template <int CPW, int N>
__device__ __noinline__ vecf<CPW> sample_slow(const io_t *__restrict__ srcb, int h, int w, int G, int gbase, int x0,
                                               int y0, float w00, float w01, float w10, float w11) {
    vecf<CPW> S = 0.f;
    if (x0 < -1 || x0 >= w || y0 < -1 || y0 >= h) return S;
    const size_t hw = (size_t)h * w;
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0, vx1 = x1 < w, vy0 = y0 >= 0, vy1 = y1 < h;
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
        const io_t *pl = srcb + (size_t)Slots<CPW / N, N>::channel(k, G, gbase) * hw;
        float s = 0.f;
        if (vx0 && vy0) s += ldio(pl + y0 * w + x0) * w00;
        if (vx1 && vy0) s += ldio(pl + y0 * w + x1) * w01;
        if (vx0 && vy1) s += ldio(pl + y1 * w + x0) * w10;
        if (vx1 && vy1) s += ldio(pl + y1 * w + x1) * w11;
        S[k] = s;
    }
    return S;
}

// d_src scatter for a cell outside the staged window (global float atomics, bounds checked).
template <int CPW, int N>
__device__ __noinline__ void scatter_slow(float *__restrict__ dsrcb, int h, int w, int G, int gbase, int x0, int y0,
                                          vecf<CPW> a0, vecf<CPW> a1, vecf<CPW> a2, vecf<CPW> a3) {
    const size_t hw = (size_t)h * w;
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < w, vx1 = x1 >= 0 && x1 < w, vy0 = y0 >= 0 && y0 < h, vy1 = y1 >= 0 && y1 < h;
    if (!((vx0 || vx1) && (vy0 || vy1))) return;
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
        float *pl = dsrcb + (size_t)Slots<CPW / N, N>::channel(k, G, gbase) * hw;
        if (vx0 && vy0) unsafeAtomicAdd(pl + y0 * w + x0, a0[k]);
        if (vx1 && vy0) unsafeAtomicAdd(pl + y0 * w + x1, a1[k]);
        if (vx0 && vy1) unsafeAtomicAdd(pl + y1 * w + x0, a2[k]);
        if (vx1 && vy1) unsafeAtomicAdd(pl + y1 * w + x1, a3[k]);
    }
}

// Bilinear samples of the CPW staged channels at tap t ('zeros' padding comes from the zero-filled window).
template <int CPW, int N, int TW>
__device__ __forceinline__ vecf<CPW> sample_window(const float4 *win, const Tap4 &t, int ox, int oy,
                                                   const io_t *__restrict__ srcb, int h, int w, int G, int gbase) {
    using T = Tile<TW, CPW>;
    constexpr int QPP = CPW / 4;
    const int lx = t.x0 - ox, ly = t.y0 - oy;
    const bool in_win = (unsigned)lx < (unsigned)(T::WW - 1) && (unsigned)ly < (unsigned)(T::WH - 1);
    vecf<CPW> S;
    if (in_win) {
        const float4 *wp = win + ly * T::WW + lx;
#pragma unroll
        for (int q = 0; q < QPP; ++q) {
            const float4 a00 = wp[q * T::WP], a01 = wp[q * T::WP + 1];
            const float4 a10 = wp[q * T::WP + T::WW], a11 = wp[q * T::WP + T::WW + 1];
            S[q * 4 + 0] = a00.x * t.w00 + a01.x * t.w01 + a10.x * t.w10 + a11.x * t.w11;
            S[q * 4 + 1] = a00.y * t.w00 + a01.y * t.w01 + a10.y * t.w10 + a11.y * t.w11;
            S[q * 4 + 2] = a00.z * t.w00 + a01.z * t.w01 + a10.z * t.w10 + a11.z * t.w11;
            S[q * 4 + 3] = a00.w * t.w00 + a01.w * t.w01 + a10.w * t.w10 + a11.w * t.w11;
        }
    } else {
        S = sample_slow<CPW, N>(srcb, h, w, G, gbase, t.x0, t.y0, t.w00, t.w01, t.w10, t.w11);
    }
    return S;
}

// In-window fast path: bilinear samples of all CPW channels from the LDS window, multiplied by the ref features
// (1/N folded in) and summed per group.  og[j] = group j's output.
template <int GS, int N, int WW_, int WP_>
__device__ __forceinline__ void groups_from_window(const float4 *wp, const Tap4 &t, const vecf<GS * N> &rf, float (&og)[GS]) {
    constexpr int CPW = GS * N, QPP = CPW / 4;
    struct T { enum { WW = WW_, WP = WP_ }; };
    using SL = Slots<GS, N>;
    const v2f w00 = t.w00, w01 = t.w01, w10 = t.w10, w11 = t.w11;  // broadcast pairs -> v_pk_fma_f32
    if (SL::QUADMAP) {
#pragma unroll
        for (int m = 0; m < GS / 4; ++m) {
            v2f olo = 0.f, ohi = 0.f;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int q = m * N + i;
                // (issuing all 4*QPP tap reads up front was tried: 187-256 VGPRs, one wave per SIMD less, slower)
                const float4 a00 = wp[q * T::WP], a01 = wp[q * T::WP + 1];
                const float4 a10 = wp[q * T::WP + T::WW], a11 = wp[q * T::WP + T::WW + 1];
                // two channels per packed instruction (a quad's .xy / .zw sit in aligned register pairs)
                const v2f lo = v2f{a00.x, a00.y} * w00 + v2f{a01.x, a01.y} * w01 + v2f{a10.x, a10.y} * w10 + v2f{a11.x, a11.y} * w11;
                const v2f hi = v2f{a00.z, a00.w} * w00 + v2f{a01.z, a01.w} * w01 + v2f{a10.z, a10.w} * w10 + v2f{a11.z, a11.w} * w11;
                olo += lo * v2f{rf[q * 4 + 0], rf[q * 4 + 1]};
                ohi += hi * v2f{rf[q * 4 + 2], rf[q * 4 + 3]};
            }
            og[4 * m + 0] = olo.x; og[4 * m + 1] = olo.y; og[4 * m + 2] = ohi.x; og[4 * m + 3] = ohi.y;
        }
    } else {
        float S[CPW];
#pragma unroll
        for (int q = 0; q < QPP; ++q) {
            const float4 a00 = wp[q * T::WP], a01 = wp[q * T::WP + 1];
            const float4 a10 = wp[q * T::WP + T::WW], a11 = wp[q * T::WP + T::WW + 1];
            S[q * 4 + 0] = (a00.x * t.w00 + a01.x * t.w01 + a10.x * t.w10 + a11.x * t.w11) * rf[q * 4 + 0];
            S[q * 4 + 1] = (a00.y * t.w00 + a01.y * t.w01 + a10.y * t.w10 + a11.y * t.w11) * rf[q * 4 + 1];
            S[q * 4 + 2] = (a00.z * t.w00 + a01.z * t.w01 + a10.z * t.w10 + a11.z * t.w11) * rf[q * 4 + 2];
            S[q * 4 + 3] = (a00.w * t.w00 + a01.w * t.w01 + a10.w * t.w10 + a11.w * t.w11) * rf[q * 4 + 3];
        }
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < N; ++i) acc += S[SL::slot(j, i)];
            og[j] = acc;
        }
    }
}

// Slow path (tap outside the staged window): group outputs from bounds-checked global loads.
template <int GS, int N>
__device__ __forceinline__ void groups_from_global(const io_t *__restrict__ srcb, int h, int w, int G, int gbase, const Tap4 &t,
                                                   const vecf<GS * N> &rf, float (&og)[GS]) {
    using SL = Slots<GS, N>;
    const vecf<GS * N> S = sample_slow<GS * N, N>(srcb, h, w, G, gbase, t.x0, t.y0, t.w00, t.w01, t.w10, t.w11);
#pragma unroll
    for (int j = 0; j < GS; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) acc += S[SL::slot(j, i)] * rf[SL::slot(j, i)];
        og[j] = acc;
    }
}

// Tile set-up shared by forward and backward: the thread's pixel and walk state, the window origin, and the
// staged source window.  Returns false for threads outside the image (they still helped stage the window).
template <int GS, int N, int TW, bool FUSED>
__device__ __forceinline__ bool tile_setup(const io_t *__restrict__ src, const float *__restrict__ K,
                                           const float *__restrict__ invK, const float *__restrict__ pose,
                                           const float *__restrict__ hyp, const float *__restrict__ prior,
                                           const float *__restrict__ ztrans, const CvDims &dm, const Seg &sg, float4 *win,
                                           int *bb, float *itv, Walk<FUSED> &wk, int &b, int &gbase, int &p, int &d0,
                                           int &d1, int &ox, int &oy) {
    constexpr int CPW = GS * N, QPP = CPW / 4;
    using T = Tile<TW, CPW>;
    const int tid = threadIdx.x;
    b = sg.b;
    gbase = sg.gs * GS;
    const int tx0 = (sg.tile % dm.tiles_x) * TW, ty0 = (sg.tile / dm.tiles_x) * T::TH;
    const int x = tx0 + tid % TW, y = ty0 + tid / TW;
    const bool valid = x < dm.w && y < dm.h;
    p = y * dm.w + x;
    d0 = sg.d0;
    d1 = sg.d1;
    const size_t hw = (size_t)dm.h * dm.w;
    wk.cam = md_load_cam(K + b * 16, invK + b * 16, pose + b * 16);
    md_ray(wk.cam, (float)x, (float)y, wk.r0, wk.r1, wk.r2);
    wk.A0 = wk.cam.P[0] * wk.r0 + wk.cam.P[1] * wk.r1 + wk.cam.P[2] * wk.r2; wk.B0 = wk.cam.P[3];
    wk.A1 = wk.cam.P[4] * wk.r0 + wk.cam.P[5] * wk.r1 + wk.cam.P[6] * wk.r2; wk.B1 = wk.cam.P[7];
    wk.A2 = wk.cam.P[8] * wk.r0 + wk.cam.P[9] * wk.r1 + wk.cam.P[10] * wk.r2; wk.B2 = wk.cam.P[11] + 1e-7f;
    wk.wm1 = (float)(dm.w - 1); wk.hm1 = (float)(dm.h - 1);
    wk.rw = 1.f / wk.wm1; wk.rh = 1.f / wk.hm1;
    wk.sched_type = dm.sched_type;
    wk.d0 = d0;
    wk.hyp_stride = hw;
    wk.hyp_p = nullptr;
    wk.hc.a = 1.f; wk.hc.b = 0.f;
    if (FUSED) {
        const float one_pf = 1.f + (ztrans ? dm.scale_fac * ztrans[b] : dm.scale_fac);
        const float c = valid ? prior[(size_t)b * hw + p] : 1.f;
        wk.hc = md_hyp_const(c, one_pf, dm.sched_type);
        for (int i = tid; i < d1 - d0; i += 256) itv[i] = md_hyp_itv(d0 + i, dm.D, dm.sched_type);
        __syncthreads();
    } else {
        wk.hyp_p = hyp + (size_t)b * dm.D * hw + (valid ? p : 0);
    }
    // bounding box of the north-west taps at both ends of the hypothesis slice (u(d), v(d) are monotone in d
    // between them unless c_z changes sign; stragglers take the global path)
    int mnx = INT_MAX, mny = INT_MAX, mxx = INT_MIN, mxy = INT_MIN;
    if (valid && d1 > d0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // (allowing for rounding: a coordinate sitting on an integer -- static camera -- lands on either side of it from
            // step to step; bounding by the end points' taps alone sent such rows down the global-memory path: 3-7x slower)
            int xl, xh, yl, yh;
            wk.tap_range(wk.hypothesis(itv, e ? d1 - 1 : d0), xl, xh, yl, yh);
            if (xh >= -1 && xl < dm.w && yh >= -1 && yl < dm.h) {
                mnx = min(mnx, max(xl, -1)); mxx = max(mxx, min(xh, dm.w - 1) + 1);
                mny = min(mny, max(yl, -1)); mxy = max(mxy, min(yh, dm.h - 1) + 1);
            }
        }
    }
    mnx = md_wave_min(mnx); mny = md_wave_min(mny); mxx = md_wave_max(mxx); mxy = md_wave_max(mxy);
    const int wave = tid >> 6;
    if ((tid & 63) == 0) { bb[wave * 4] = mnx; bb[wave * 4 + 1] = mny; bb[wave * 4 + 2] = mxx; bb[wave * 4 + 3] = mxy; }
    __syncthreads();
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
        mnx = min(mnx, bb[wv * 4]); mny = min(mny, bb[wv * 4 + 1]);
        mxx = max(mxx, bb[wv * 4 + 2]); mxy = max(mxy, bb[wv * 4 + 3]);
    }
    if (mnx > mxx) { mnx = tx0; mxx = tx0; mny = ty0; mxy = ty0; }  // nothing lands in the image
    const int spanx = mxx - mnx + 1, spany = mxy - mny + 1;
    ox = spanx <= T::WW ? mnx : mnx + (spanx - T::WW) / 2;
    oy = spany <= T::WH ? mny : mny + (spany - T::WH) / 2;
    // stage the window: consecutive threads -> consecutive columns (coalesced per channel plane).  Four quads
    // (16 loads) are issued before the first LDS write so the L2 round trips overlap instead of queueing
    // (the staging loop used to be ~a quarter of the kernel: 15 dependent round trips per workgroup).
    const io_t *srcb = src + (size_t)b * dm.C * hw;
    constexpr int NQ = T::WP * QPP, NIT = (NQ + 255) / 256, UNR = 4;
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += UNR) {
        float4 v[UNR];
        int dst[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = tid + (it0 + u) * 256;
            const int wx = idx % T::WW, rest = idx / T::WW;
            const int wy = rest % T::WH, q = rest / T::WH;
            const int sx = ox + wx, sy = oy + wy;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            dst[u] = (it0 + u < NIT && idx < NQ) ? q * T::WP + wy * T::WW + wx : -1;
            if (dst[u] >= 0 && sx >= 0 && sx < dm.w && sy >= 0 && sy < dm.h) {
                const size_t o = (size_t)sy * dm.w + sx;
                v[u].x = ldio(srcb + (size_t)Slots<GS, N>::channel(q * 4 + 0, dm.G, gbase) * hw + o);
                v[u].y = ldio(srcb + (size_t)Slots<GS, N>::channel(q * 4 + 1, dm.G, gbase) * hw + o);
                v[u].z = ldio(srcb + (size_t)Slots<GS, N>::channel(q * 4 + 2, dm.G, gbase) * hw + o);
                v[u].w = ldio(srcb + (size_t)Slots<GS, N>::channel(q * 4 + 3, dm.G, gbase) * hw + o);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (dst[u] >= 0) win[dst[u]] = v[u];
    }
    __syncthreads();
    return valid;
}

template <int GS, int N, int TW, bool FUSED>
__global__ __launch_bounds__(256) void costvol_fwd_kernel(const io_t *__restrict__ ref, const io_t *__restrict__ src,
                                                          const float *__restrict__ K, const float *__restrict__ invK,
                                                          const float *__restrict__ pose, const float *__restrict__ hyp,
                                                          const float *__restrict__ prior,
                                                          const float *__restrict__ ztrans, io_t *__restrict__ out,
                                                          const CvDims dm) {
    constexpr int CPW = GS * N, QPP = CPW / 4;
    using T = Tile<TW, CPW>;
    __shared__ float4 win[T::WP * QPP];
    __shared__ float itv[ITV_MAX];
    __shared__ int bb[16];
    const long long total = (long long)dm.items * dm.D;
    long long lo = total * blockIdx.x / gridDim.x;
    const long long hi = total * (blockIdx.x + 1) / gridDim.x;
    Seg sg;
    while (next_segment(dm, lo, hi, sg)) {
    Walk<FUSED> wk;
    int b, gbase, p, d0, d1, ox, oy;
    const bool valid = tile_setup<GS, N, TW, FUSED>(src, K, invK, pose, hyp, prior, ztrans, dm, sg, win, bb, itv, wk, b,
                                                    gbase, p, d0, d1, ox, oy);
    const size_t hw = (size_t)dm.h * dm.w;
    const io_t *srcb = src + (size_t)b * dm.C * hw;
    vecf<CPW> rf = 0.f;  // ref features with the 1/N of the group mean folded in
    if (valid) {
#pragma unroll
        for (int k = 0; k < CPW; ++k)
            rf[k] = ldio(ref + ((size_t)b * dm.C + Slots<GS, N>::channel(k, dm.G, gbase)) * hw + p) * (1.f / (float)N);
    }
    io_t *outp = out + (size_t)b * dm.sb + (size_t)gbase * dm.sg + (size_t)d0 * dm.sd + (size_t)p * dm.sp;
    float dnext = valid ? wk.hypothesis(itv, d0) : 1.f;
    for (int d = d0; valid && d < d1; ++d) {
        const float dep = dnext;
        if (d + 1 < d1) dnext = wk.hypothesis(itv, d + 1);
        const Tap4 t = wk.tap_at(dep);
        const int lx = t.x0 - ox, ly = t.y0 - oy;
        float og[GS];
        if ((unsigned)lx < (unsigned)(T::WW - 1) && (unsigned)ly < (unsigned)(T::WH - 1))
            groups_from_window<GS, N, T::WW, T::WP>(win + ly * T::WW + lx, t, rf, og);
        else
            groups_from_global<GS, N>(srcb, dm.h, dm.w, dm.G, gbase, t, rf, og);
#pragma unroll
        for (int j = 0; j < GS; ++j) stio(outp + (size_t)j * dm.sg, og[j]);
        outp += dm.sd;
    }
    __syncthreads();  // the window is restaged by the next segment
    }
}

// Channels-last volume (B,D,h,w,G): the layout MIOpen's fast 3-D convolutions consume.  A pixel's groups are
// contiguous in memory, so a lane owning a pixel would store 16-byte pieces at a 4*G-byte stride: every store
// instruction would touch 64 different 64-byte segments (measured 107 us vs 60 us planar).  Instead each wave
// transposes its (64 pixels x GS groups) result through a private, padded LDS tile and writes it back in memory
// order: lane l of store k writes the 16 bytes at linear position (k*64 + l) of the wave's pixel-major block --
// consecutive lanes, consecutive addresses, 1 KB per instruction.
template <int GS, int N, int TW, bool FUSED>
__global__ __launch_bounds__(256) void costvol_fwd_nhwc_kernel(const io_t *__restrict__ ref, const io_t *__restrict__ src,
                                                               const float *__restrict__ K, const float *__restrict__ invK,
                                                               const float *__restrict__ pose, const float *__restrict__ hyp,
                                                               const float *__restrict__ prior,
                                                               const float *__restrict__ ztrans, io_t *__restrict__ out,
                                                               const CvDims dm) {
    constexpr int CPW = GS * N, QPP = CPW / 4, NCH = GS / 4;  // NCH 16-byte chunks of groups per pixel
    // transpose tile: [64 pixels][NCH chunks] of float4, chunk index XOR-swizzled with the pixel so that both the
    // per-pixel writes (8-lane groups, 64-byte pitch) and the memory-order reads are bank-conflict free without
    // padding (padding would push the workgroup past half a CU's LDS)
    constexpr int PADW = NCH;
    constexpr int SWZ = NCH - 1;  // NCH is 1, 2 or 4
    static_assert(GS % 4 == 0, "channels-last stores need a multiple of 4 groups per workgroup");
    using T = Tile<TW, CPW>;
    __shared__ float4 win[T::WP * QPP];
    __shared__ float4 stage[4 * 64 * PADW];
    __shared__ float itv[ITV_MAX];
    __shared__ int bb[16];
    const long long total = (long long)dm.items * dm.D;
    long long lo = total * blockIdx.x / gridDim.x;
    const long long hi = total * (blockIdx.x + 1) / gridDim.x;
    Seg sg;
    while (next_segment(dm, lo, hi, sg)) {
    Walk<FUSED> wk;
    int b, gbase, p, d0, d1, ox, oy;
    const bool valid = tile_setup<GS, N, TW, FUSED>(src, K, invK, pose, hyp, prior, ztrans, dm, sg, win, bb, itv, wk, b,
                                                    gbase, p, d0, d1, ox, oy);
    const size_t hw = (size_t)dm.h * dm.w;
    const io_t *srcb = src + (size_t)b * dm.C * hw;
    vecf<CPW> rf = 0.f;  // ref features with the 1/N of the group mean folded in
    if (valid) {
#pragma unroll
        for (int k = 0; k < CPW; ++k)
            rf[k] = ldio(ref + ((size_t)b * dm.C + Slots<GS, N>::channel(k, dm.G, gbase)) * hw + p) * (1.f / (float)N);
    }
    // write-back roles: store k of this lane covers 16-byte piece (k*64 + lane) of the wave's block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 *my_stage = stage + wave * 64 * PADW;
    constexpr int RPW = 64 / TW > 0 ? 64 / TW : 1;  // tile rows per wave (TW = 32: 2, TW = 64: 1)
    const int tx0 = (sg.tile % dm.tiles_x) * TW, ty0 = (sg.tile / dm.tiles_x) * T::TH + wave * RPW;
    long long soff[NCH];  // global offsets (floats) of this lane's NCH stores; < 0: nothing to store
    int sidx[NCH];        // where the piece sits in the padded transpose tile
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int L = k * 64 + lane, pw = L / NCH, ch = L % NCH;  // pixel-in-wave, chunk
        const int px = tx0 + pw % TW, py = ty0 + pw / TW;
        sidx[k] = pw * PADW + (ch ^ ((pw >> 1) & SWZ));
        soff[k] = (px < dm.w && py < dm.h)
                      ? (long long)b * dm.sb + (long long)d0 * dm.sd + ((long long)py * dm.w + px) * dm.sp + gbase + ch * 4
                      : -1;
    }
    float dnext = valid ? wk.hypothesis(itv, d0) : 1.f;
    for (int d = d0; d < d1; ++d) {
        float og[GS];
        if (valid) {
            const float dep = dnext;
            if (d + 1 < d1) dnext = wk.hypothesis(itv, d + 1);
            const Tap4 t = wk.tap_at(dep);
            const int lx = t.x0 - ox, ly = t.y0 - oy;
            if ((unsigned)lx < (unsigned)(T::WW - 1) && (unsigned)ly < (unsigned)(T::WH - 1))
                groups_from_window<GS, N, T::WW, T::WP>(win + ly * T::WW + lx, t, rf, og);
            else
                groups_from_global<GS, N>(srcb, dm.h, dm.w, dm.G, gbase, t, rf, og);
        } else {
#pragma unroll
            for (int j = 0; j < GS; ++j) og[j] = 0.f;
        }
        // transpose through the wave's LDS tile (LDS operations of one wave execute in order: no barrier)
#pragma unroll
        for (int c = 0; c < NCH; ++c) my_stage[lane * PADW + (c ^ ((lane >> 1) & SWZ))] = make_float4(og[4 * c], og[4 * c + 1], og[4 * c + 2], og[4 * c + 3]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const float4 v = my_stage[sidx[k]];
            // (nontemporal stores measured the same, cold output: 76.8 / 77.1 / 76.8 us against 76.7 / 77.1 / 77.3)
            if (soff[k] >= 0) stio4(out + soff[k] + (long long)(d - d0) * dm.sd, v);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();  // the window is restaged by the next segment
    }
}

template <int GS, int N, int TW, bool FUSED>
__global__ __launch_bounds__(256) void costvol_bwd_kernel(const io_t *__restrict__ gout, const io_t *__restrict__ ref,
                                                          const io_t *__restrict__ src, const float *__restrict__ K,
                                                          const float *__restrict__ invK, const float *__restrict__ pose,
                                                          const float *__restrict__ hyp, const float *__restrict__ prior,
                                                          const float *__restrict__ ztrans, float *__restrict__ d_ref,
                                                          float *__restrict__ d_src, const CvDims dm) {
    constexpr int CPW = GS * N, QPP = CPW / 4;
    using T = Tile<TW, CPW>;
    constexpr int WP = T::WP;
    __shared__ float4 win[WP * QPP];
    __shared__ float gw[CPW * WP];  // planar d_src window: lanes on neighbouring columns hit neighbouring banks
    __shared__ float itv[ITV_MAX];
    __shared__ int bb[16];
    const int tid = threadIdx.x;
    const long long total = (long long)dm.items * dm.D;
    long long lo = total * blockIdx.x / gridDim.x;
    const long long hi = total * (blockIdx.x + 1) / gridDim.x;
    Seg sg;
    while (next_segment(dm, lo, hi, sg)) {
    for (int i = tid; i < CPW * WP; i += 256) gw[i] = 0.f;
    Walk<FUSED> wk;
    int b, gbase, p, d0, d1, ox, oy;
    const bool valid = tile_setup<GS, N, TW, FUSED>(src, K, invK, pose, hyp, prior, ztrans, dm, sg, win, bb, itv, wk, b,
                                                    gbase, p, d0, d1, ox, oy);  // its barriers publish the zeroed gw
    const size_t hw = (size_t)dm.h * dm.w;
    const io_t *srcb = src + (size_t)b * dm.C * hw;
    float *dsrcb = d_src + (size_t)b * dm.C * hw;
    if (valid) {
        vecf<CPW> rf, dref = 0.f;
#pragma unroll
        for (int k = 0; k < CPW; ++k)
            rf[k] = ldio(ref + ((size_t)b * dm.C + Slots<GS, N>::channel(k, dm.G, gbase)) * hw + p) * (1.f / (float)N);
        vecf<CPW> a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // pending d_src quad at cell (bx, by)
        int bx = INT_MIN, by = INT_MIN;
        const io_t *gp = gout + (size_t)b * dm.sb + (size_t)gbase * dm.sg + (size_t)d0 * dm.sd + (size_t)p * dm.sp;
        float dnext = wk.hypothesis(itv, d0);
        for (int d = d0; d <= d1; ++d) {
            Tap4 t;
            vecf<CPW> gv = 0.f;
            if (d < d1) {
                const float dep = dnext;
                if (d + 1 < d1) dnext = wk.hypothesis(itv, d + 1);
                float gq[GS];
#pragma unroll
                for (int j = 0; j < GS; ++j) gq[j] = ldio(gp + (size_t)j * dm.sg);
                gp += dm.sd;
                t = wk.tap_at(dep);
                const vecf<CPW> S = sample_window<CPW, N, TW>(win, t, ox, oy, srcb, dm.h, dm.w, dm.G, gbase);
#pragma unroll
                for (int k = 0; k < CPW; ++k) {
                    dref[k] += gq[Slots<GS, N>::group(k)] * S[k];
                    gv[k] = gq[Slots<GS, N>::group(k)] * rf[k];
                }
            } else {
                t.x0 = INT_MIN; t.y0 = INT_MIN;  // sentinel iteration: forces the final flush
                t.w00 = t.w01 = t.w10 = t.w11 = 0.f;
            }
            if (t.x0 != bx || t.y0 != by) {
                if (bx != INT_MIN) {
                    const int lx = bx - ox, ly = by - oy;
                    if ((unsigned)lx < (unsigned)(T::WW - 1) && (unsigned)ly < (unsigned)(T::WH - 1)) {
                        float *g = gw + ly * T::WW + lx;
#pragma unroll
                        for (int k = 0; k < CPW; ++k) {
                            atomicAdd(g + k * WP, a0[k]);
                            atomicAdd(g + k * WP + 1, a1[k]);
                            atomicAdd(g + k * WP + T::WW, a2[k]);
                            atomicAdd(g + k * WP + T::WW + 1, a3[k]);
                        }
                    } else {
                        scatter_slow<CPW, N>(dsrcb, dm.h, dm.w, dm.G, gbase, bx, by, a0, a1, a2, a3);
                    }
                }
                bx = t.x0; by = t.y0;
                a0 = 0.f; a1 = 0.f; a2 = 0.f; a3 = 0.f;
            }
            a0 += gv * t.w00; a1 += gv * t.w01; a2 += gv * t.w10; a3 += gv * t.w11;
        }
        float *drp = d_ref + (size_t)b * dm.C * hw + p;
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            float *o = drp + (size_t)Slots<GS, N>::channel(k, dm.G, gbase) * hw;
            const float v = dref[k] * (1.f / (float)N);  // the group mean's 1/N (rf carries it on the d_src side)
            if (d0 == 0 && d1 == dm.D) *o = v;  // this segment is the pixel's only contributor
            else unsafeAtomicAdd(o, v);
        }
    }
    __syncthreads();
    // flush the d_src window (cells outside the image are grid_sample's zero padding: dropped)
    for (int idx = tid; idx < CPW * WP; idx += 256) {
        const float v = gw[idx];
        if (v == 0.f) continue;
        const int k = idx / WP, cell = idx % WP;
        const int sx = ox + cell % T::WW, sy = oy + cell / T::WW;
        if (sx >= 0 && sx < dm.w && sy >= 0 && sy < dm.h)
            unsafeAtomicAdd(dsrcb + (size_t)Slots<GS, N>::channel(k, dm.G, gbase) * hw + (size_t)sy * dm.w + sx, v);
    }
    __syncthreads();  // gw / win are reused by the next segment
    }
}

// Launch-shape switches of earlier rounds' sweeps.  Compile-time only (tools/ab_build.sh NAME -DMD_COSTVOL_...=v builds a variant,
// MOVEDEPTH_HIP_LIB selects it): the library reads nothing from the process environment -- what a caller may choose is an
// argument of the entry point (`flags` of md_costvol_bwd*, include/movedepth_hip.h).
#ifndef MD_COSTVOL_NWG
#define MD_COSTVOL_NWG 0            // > 0: forward grid size (workgroups) instead of the occupancy-derived one
#endif
#ifndef MD_COSTVOL_NWG_BWD
#define MD_COSTVOL_NWG_BWD 0
#endif
#ifndef MD_COSTVOL_TWO_PHASE_FWD
#define MD_COSTVOL_TWO_PHASE_FWD 0  // two-phase schedule (launch_cl_inst): backward only
#endif
#ifndef MD_COSTVOL_TWO_PHASE_BWD
#define MD_COSTVOL_TWO_PHASE_BWD 1
#endif
#ifndef MD_COSTVOL_XCD_MAP
#define MD_COSTVOL_XCD_MAP 0        // contiguous slice ranges per XCD (cv_share): measured no gain
#endif
#ifndef MD_COSTVOL_XCD_MAP_BWD
#define MD_COSTVOL_XCD_MAP_BWD 0
#endif
#ifndef MD_COSTVOL_BWD_SHAPE0
#define MD_COSTVOL_BWD_SHAPE0 0     // which window shape the backward's fit test tries first (ClShapes)
#endif
#ifndef MD_COSTVOL_MIN_SUB
#define MD_COSTVOL_MIN_SUB 24       // shortest hypothesis sub-slice a window is staged for (channels-last features)
#endif
#ifndef MD_COSTVOL_MIN_SUB_BWD
#define MD_COSTVOL_MIN_SUB_BWD 24
#endif
#ifndef MD_COSTVOL_CPW
#define MD_COSTVOL_CPW 0            // first-generation kernels: channels per workgroup (0: 16, or 32 for a channels-last volume)
#endif
#ifndef MD_COSTVOL_CPW_BWD
#define MD_COSTVOL_CPW_BWD 8
#endif
#ifndef MD_COSTVOL_DSPLIT
#define MD_COSTVOL_DSPLIT 0         // first-generation kernels: hypothesis slices per item (0: derived)
#endif
#ifndef MD_COSTVOL_DSPLIT_BWD
#define MD_COSTVOL_DSPLIT_BWD 0
#endif
#ifndef MD_COSTVOL_CL_FILL
#define MD_COSTVOL_CL_FILL 1
#endif
#ifndef MD_COSTVOL_CL
#define MD_COSTVOL_CL 1             // 0: never take the channels-last-volume kernels (planar-era kernels for every layout; A/B builds)
#endif

struct CvPtrs {
    const io_t *gout, *ref, *src;
    const float *K, *invK, *pose, *hyp, *prior, *ztrans;
    io_t *out;
    float *d_ref, *d_src;
    unsigned flags;               // backward: MD_CV_* bits of the entry point
    unsigned long long *census;   // backward: (gathered steps << 32 | steps) of this launch, or null
    const long long *shares;      // backward: caller's partition (n_shares pairs) or null
    int n_shares;
    unsigned *cost;               // backward: per-item cycle counters or null
    int *plan_out;                // md_costvol_bwd_plan: {items, workgroups} of the launch, nothing launched
};

#include "costvol_cl.inc"

// Launch of the channels-last kernels (costvol_cl.inc).  Grid = the chip's resident workgroup slots for the kernel
// (occupancy query), each workgroup taking an equal contiguous share of the items x D hypothesis steps.
template <bool BWD, int N, int LPP, int NW, bool FUSED, bool FCL>
int launch_cl_inst(const CvPtrs &q, const CvDims &dm, hipStream_t stream) {
    CvDims dm2 = dm;
    dm2.k1 = dm2.nc = dm2.fsub = 0;
    const void *fn;   // (if constexpr: only the direction's own kernel is instantiated)
    if constexpr (BWD) fn = (const void *)cl_bwd_kernel<N, LPP, NW, FUSED, FCL>;
    else fn = (const void *)cl_fwd_kernel<N, LPP, NW, FUSED, FCL>;
    const long long total = (long long)dm.items * dm.D;
    long long nwg = BWD ? MD_COSTVOL_NWG_BWD : MD_COSTVOL_NWG;
    if (nwg <= 0) {
        // resident workgroup slots of this kernel on this chip, queried once per instantiation (the query costs tens of
        // microseconds of host time per call: with it in every launch the kernel started ~15 us late inside the training step)
        static long long slots_of[16] = {0};   // per device: a process may drive more than one GPU (static per instantiation)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
        if (slots_of[dev] == 0) {
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * NW, 0) != hipSuccess || per_cu < 1) per_cu = 1;
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            slots_of[dev] = (long long)cus * per_cu;
        }
        const long long slots = slots_of[dev];
        // k equal hypothesis slices per item, the smallest k that fills every slot at least once (>= 8 steps per slice):
        // item-aligned slices stage one window each; an arbitrary grid (e.g. exactly `slots`) makes most workgroups straddle
        // two items and stage twice (B=6, 48x160, D=96: 720 workgroups 67.8 us, 512 72.8 us, 1024 82.7 us)
        long long k = (slots + dm.items - 1) / dm.items;
        if (BWD && dm.items * 10 >= slots * 9) k = 1;  // one almost-full round beats two slices per item in two rounds
        // (Forward with more, shorter slices -- the hardware hands a free slot the next workgroup, which evens out what parallax
        // makes uneven: measured with 1440 instead of 720 workgroups at B=6, 48x160, D=96: moderate poses 72 -> 64 us, driving scene
        // 64 -> 62, but sane poses 59.4 -> 60.4, fp16 41 -> 48, config 4's shape 111 -> 126: a staging per slice is not free, and the
        // sane case is what a training step runs.  Not adopted; profiles/r05_costvol_parallax.txt.)
        // MD_CV_FINE_SLICES (forward `flags`, round 6): twice the slices per item.  The hardware hands a free slot the next workgroup,
        // which evens out what parallax makes uneven -- moderate poses 66 -> 62 us, driving scene 64.8 -> 61.5 at config 2's shape --
        // and costs the sane case a second staging per item (58 -> 59.1 us; fp16 41 -> 48): the CALLER's choice, made by
        // ops.BackwardPolicy from the previous backward's census (sub-slices staged per segment > 1: the tiles' sweeps outgrow
        // their windows), fp32 only.
        if (!BWD && (q.flags & MD_CV_FINE_SLICES)) k *= 2;
        if (k > dm.D / 8) k = dm.D / 8;
        if (k < 1) k = 1;
        nwg = (long long)dm.items * k;
        // Two-phase schedule.  With more slices than slots the last round is partly empty (config 2: 720 slices on 512
        // slots = one full round and one 40 % full, both as long as a slice: the wave-residency counters show every wave
        // resident for half of the kernel's duration).  The slices of that last round are cut into `fsub` pieces so that
        // they fill the slots once more: makespan (rounds - 1) * L + L / fsub instead of rounds * L.  Long shares first.
        dm2.k1 = (int)k;
        dm2.nc = (int)nwg;
        dm2.fsub = 1;
        // Backward only: measured at config 2, backward 97-103 -> 90-91 us stand-alone and 95.6 -> 90.8 us in the training
        // step; the forward (store-bound, 24-step pieces re-stage their window) got slower, 60 -> 63 us, and keeps plain slices.
        if (nwg > slots && (BWD ? MD_COSTVOL_TWO_PHASE_BWD : MD_COSTVOL_TWO_PHASE_FWD)) {
            const long long full = nwg / slots * slots, rest = nwg - full;
            long long f = rest > 0 ? slots / rest : 1;
            while (f > 1 && dm.D / k / f < 8) --f;
            if (rest > 0 && f > 1) {
                dm2.nc = (int)full;
                dm2.fsub = (int)f;
                nwg = full + rest * f;
            }
        }
    }
    if (dm2.k1 > 0 && (dm.D + dm2.k1 - 1) / dm2.k1 > ITV_MAX) dm2.k1 = 0;  // an item-aligned slice (up to ceil(D / k) steps) must fit the interval table: plain shares
    if (nwg > total) { nwg = total; dm2.k1 = 0; }
    if (nwg * ITV_MAX < total) { nwg = (total + ITV_MAX - 1) / ITV_MAX; dm2.k1 = 0; }  // a share fits the interval table
    if (q.plan_out) {   // md_costvol_bwd_plan: the launch geometry, nothing launched
        q.plan_out[0] = dm.items;
        q.plan_out[1] = (int)nwg;
        return MD_OK;
    }
    dm2.shares = nullptr;
    dm2.cost = nullptr;
    bool cost_cleared_census = false;
    if constexpr (BWD) {
        // The caller's partition of the items x D steps (round 6): one [lo, hi) pair per workgroup, from the per-item cycle counts
        // (`cost`) of an earlier launch on similar poses -- all workgroups of this kernel are resident at once, so no hardware
        // scheduler evens out what parallax makes uneven (workgroup lifetimes max / mean 1.4-1.5 on the driving-scene and moderate
        // cases: the launch lasts as long as its slowest tile).  A share may span items (next_segment cuts it at item boundaries).
        if (q.shares) {
            MD_REQUIRE(q.n_shares > 0 && dm.D <= ITV_MAX, "md_costvol_bwd: shares need n_shares > 0 and D <= %d", ITV_MAX);
            nwg = q.n_shares;
            dm2.k1 = dm2.nc = dm2.fsub = 0;
            dm2.shares = q.shares;
        }
        if (q.cost) {
            // (census word directly in front of the counters, as ops.BackwardPolicy lays them out: one fill for both)
            const bool adj = q.census && (char *)q.census + sizeof(unsigned long long) == (char *)q.cost;
            MD_CHECK_HIP(hipMemsetAsync(adj ? (void *)q.census : (void *)q.cost, 0, sizeof(unsigned) * (size_t)dm.items + (adj ? sizeof(unsigned long long) : 0), stream));
            dm2.cost = q.cost;
            cost_cleared_census = adj;
        }
    }
    const dim3 grid((unsigned)nwg), block(64 * NW);
    const char *tname = BWD ? MD_CV_STR(MD_CV_NAME(md_costvol_bwd)) : MD_CV_STR(MD_CV_NAME(md_costvol_fwd));
    if (BWD) {
        // d_src is accumulated with atomics from every workgroup whose window covers a cell: zero it.  d_ref is STORED when a
        // segment is the pixel's only contributor -- true for every segment when each item is one whole-D slice and no other
        // launch shares the samples -- so it only needs the fill otherwise (config 2: 720 items on 768 slots, k = 1).
        const bool whole = dm2.k1 == 1 && dm2.fsub == 1 && nwg == dm.items && !MD_CL_DREF_ATOMIC && !dm2.shares;
        const size_t bytes = sizeof(float) * (size_t)dm.B * dm.C * dm.h * dm.w;
        if (!whole && (char *)q.d_ref + bytes == (char *)q.d_src) {
            MD_CHECK_HIP(hipMemsetAsync(q.d_ref, 0, 2 * bytes, stream));
        } else {
            MD_CHECK_HIP(hipMemsetAsync(q.d_src, 0, bytes, stream));
            if (!whole) MD_CHECK_HIP(hipMemsetAsync(q.d_ref, 0, bytes, stream));
        }
    }
    if (BWD && q.census && !cost_cleared_census) MD_CHECK_HIP(hipMemsetAsync(q.census, 0, sizeof(unsigned long long), stream));
    dm2.census = BWD ? q.census : nullptr;
    hipEvent_t ev0, ev1;
    md_timing_pair(tname, &ev0, &ev1);
    if constexpr (BWD) {
        // MD_CV_GATHER_TABLE in `flags`: the build of the 16 x 4-tile kernel whose gather mode merges a tile's d_src terms per source cell
        // in LDS before they leave the CU (costvol_cl.inc, TAB).  For the wild poses of an untrained pose network, where the launch is
        // bound by the L2's float-atomic rate: 540 -> 314 us at B=6, 48x160, D=96.  Its own instantiation and the CALLER's choice because the
        // table code costs the kernel that carries it: 168 registers + 3 spilled (scratch set-up per launch) and a slower gather walk when
        // the table does not pay -- sane poses 61.7 -> 65 us, driving scene 106 -> 123, moderate 122 -> 138 (profiles/r05_costvol_table.txt).
        // The caller decides from the `census` word of its previous launches (share of the hypothesis steps that ran in gather mode:
        // movedepth_amd/ops.py GatherTablePolicy); the library keeps no state and reads no environment.
        const bool table = (q.flags & MD_CV_GATHER_TABLE) != 0;
        if constexpr (FCL && LPP == 4 && NW == 4 && N <= 2) {
            if (table) {
                hipExtLaunchKernelGGL((cl_bwd_kernel<N, LPP, NW, FUSED, FCL, true>), grid, block, 0, stream, ev0, ev1, 0, q.gout, q.ref, q.src,
                                      q.K, q.invK, q.pose, q.hyp, q.prior, q.ztrans, q.d_ref, q.d_src, dm2);
                MD_CHECK_LAUNCH("md_costvol_bwd (channels-last, cell table)");
                return MD_OK;
            }
        }
        hipExtLaunchKernelGGL((cl_bwd_kernel<N, LPP, NW, FUSED, FCL>), grid, block, 0, stream, ev0, ev1, 0, q.gout, q.ref, q.src,
                              q.K, q.invK, q.pose, q.hyp, q.prior, q.ztrans, q.d_ref, q.d_src, dm2);
    } else
        hipExtLaunchKernelGGL((cl_fwd_kernel<N, LPP, NW, FUSED, FCL>), grid, block, 0, stream, ev0, ev1, 0, q.ref, q.src, q.K,
                              q.invK, q.pose, q.hyp, q.prior, q.ztrans, q.out, dm2);
    MD_CHECK_LAUNCH(BWD ? "md_costvol_bwd (channels-last)" : "md_costvol_fwd (channels-last)");
    return MD_OK;
}

template <bool BWD>
int launch_gen1(const CvPtrs &q, CvDims dm, hipStream_t stream, const char *tname);

template <bool BWD>
int launch_cl(const CvPtrs &q, CvDims dm, hipStream_t stream) {
    const int N = dm.C / dm.G, LPP = dm.G / 4, TW = 64 / LPP;
    // waves per workgroup: forward 8 (16 x 8 pixel tile), backward 4 (its LDS is 3x the source window per workgroup; measured at
    // B=6, 48x160, D=96: backward 131 us with 4 against 152 with 8, forward 68 against 62).  Only these two are instantiated.
#ifndef MD_CL_FWD_NW
#define MD_CL_FWD_NW 8
#endif
    constexpr int NW = BWD ? 4 : MD_CL_FWD_NW;
    constexpr int WX = BWD ? 1 : MD_CL_FWD_WX;   // forward tile (TW * WX) x (NW / WX), costvol_cl.inc
    dm.tiles_x = md_cdiv(dm.w, TW * WX);
    dm.tiles = dm.tiles_x * md_cdiv(dm.h, NW / WX);
    dm.splits = 1;
    dm.items = dm.B * dm.tiles;
    dm.dbg = 0;
    dm.stats = md_stats_buffer();
    // (Planar feature maps with wild poses: rounds 3-4 routed such samples to the first-generation backward through a pose
    // pre-pass whose per-launch flags lived in a ring of device globals -- hidden state, one stream only.  Gone: the trainer's path
    // is channels-last features, where the kernel itself gathers what does not fit its window; planar features keep the per-tap
    // miss path of this kernel for such samples.)
#define MD_CL_F(N_, LPP_)                                                                                        \
    do {                                                                                                         \
        if (dm.fcl)                                                                                              \
            rc = q.hyp ? launch_cl_inst<BWD, N_, LPP_, NW, false, true>(q, dm, stream)                           \
                       : launch_cl_inst<BWD, N_, LPP_, NW, true, true>(q, dm, stream);                           \
        else                                                                                                     \
            rc = q.hyp ? launch_cl_inst<BWD, N_, LPP_, NW, false, false>(q, dm, stream)                          \
                       : launch_cl_inst<BWD, N_, LPP_, NW, true, false>(q, dm, stream);                          \
    } while (0)
    int rc = MD_EINVAL;
    bool found = true;
    switch (N * 10 + LPP) {
        case 14: MD_CL_F(1, 4); break;
        case 24: MD_CL_F(2, 4); break;
        case 44: MD_CL_F(4, 4); break;
        case 12: MD_CL_F(1, 2); break;
        case 22: MD_CL_F(2, 2); break;
        case 42: MD_CL_F(4, 2); break;
        default: found = false;
    }
#undef MD_CL_F
    if (!found) {
        md_set_error("costvol: no channels-last kernel for C=%d G=%d", dm.C, dm.G);
        return MD_EINVAL;
    }
    return rc;
}

template <bool BWD>
int launch(const CvPtrs &q, CvDims dm, hipStream_t stream) {
    dm.gslack = MD_CL_GATHER_SLACK;
    dm.xcd_map = BWD ? MD_COSTVOL_XCD_MAP_BWD : MD_COSTVOL_XCD_MAP;
    dm.gslack0 = MD_CL_GATHER_SLACK0;
    dm.shape0 = MD_COSTVOL_BWD_SHAPE0 >= 0 && MD_COSTVOL_BWD_SHAPE0 < 3 ? MD_COSTVOL_BWD_SHAPE0 : 0;
    // 24: a window staged for fewer steps costs more than gathering them (a staging + a d_src window flush are ~25 k cycles of a
    // backward workgroup, ~10 of a forward one).  At B=6, 48x160, D=96, 8 -> 24: driving scene backward 133 -> 105 us, forward at 2 m
    // per frame 83 -> 71, moderate poses 73 -> 66 / 132 -> 126; sane and white-noise priors never get there; wild poses 530 -> 540
    // (profiles/r05_costvol_parallax.txt)
    dm.min_sub = BWD ? MD_COSTVOL_MIN_SUB_BWD : MD_COSTVOL_MIN_SUB;
    if (dm.min_sub < 4) dm.min_sub = 4;
    if (cl_eligible(dm, BWD ? (const void *)q.gout : (const void *)q.out)) return launch_cl<BWD>(q, dm, stream);
    if (q.plan_out) { q.plan_out[0] = q.plan_out[1] = 0; return MD_OK; }   // (no caller-side partition for the planar-era kernels)
    if (BWD && (q.shares || q.cost)) {
        md_set_error("md_costvol_bwd: shares / cost are taken by the channels-last-volume kernels only");
        return MD_EINVAL;
    }
    if (dm.fcl) {
        md_set_error("costvol: channels-last feature maps need the channels-last volume kernels (volume (B,D,h,w,G) with G = 8 or 16, "
                     "C/G = 1, 2 or 4, 16-byte aligned): got C=%d G=%d strides (%lld,%lld,%lld,%lld)", dm.C, dm.G, dm.sb, dm.sd, dm.sg, dm.sp);
        return MD_EINVAL;
    }
    if (BWD) {   // first-generation backward: both gradients accumulate with atomics
        if (q.census) MD_CHECK_HIP(hipMemsetAsync(q.census, 0, sizeof(unsigned long long), stream));   // (no gather mode, no sub-slices in these kernels)
        const size_t bytes = sizeof(float) * (size_t)dm.B * dm.C * dm.h * dm.w;
        if ((char *)q.d_ref + bytes == (char *)q.d_src) {
            MD_CHECK_HIP(hipMemsetAsync(q.d_ref, 0, 2 * bytes, stream));
        } else {
            MD_CHECK_HIP(hipMemsetAsync(q.d_src, 0, bytes, stream));
            MD_CHECK_HIP(hipMemsetAsync(q.d_ref, 0, bytes, stream));
        }
    }
    return launch_gen1<BWD>(q, dm, stream, BWD ? MD_CV_STR(MD_CV_NAME(md_costvol_bwd)) : MD_CV_STR(MD_CV_NAME(md_costvol_fwd)));
}

// First-generation kernels (planar volumes, other channel groupings)
template <bool BWD>
int launch_gen1(const CvPtrs &q, CvDims dm, hipStream_t stream, const char *tname) {
    const int N = dm.C / dm.G;
    if (N != 1 && N != 2 && N != 4 && N != 8) {
        md_set_error("costvol: unsupported channel grouping C=%d G=%d (C/G must be 1, 2, 4 or 8)", dm.C, dm.G);
        return MD_EINVAL;
    }
    // channels per workgroup: coordinates are computed once per (pixel, hypothesis) for all of them, but LDS per
    // workgroup grows with it; 16 measured best for the forward at 48x160 (8: 68 us, 16: 61 us, 32: 70 us).  The
    // backward keeps 4 x CPW scatter accumulators in registers, so it carries 8.
    // channels-last output: all of a pixel's groups must come from one workgroup to write whole 64-byte lines
    const bool cl_out = !BWD && dm.sg == 1;
    int cpw_target = BWD ? MD_COSTVOL_CPW_BWD : (MD_COSTVOL_CPW > 0 ? MD_COSTVOL_CPW : (cl_out ? 32 : 16));
    int GS = 0;
    for (int cpw = cpw_target; cpw >= 4 && !GS; cpw /= 2)
        if (cpw % N == 0 && dm.G % (cpw / N) == 0) GS = cpw / N;
    for (int cpw = cpw_target * 2; cpw <= 32 && !GS; cpw *= 2)
        if (cpw % N == 0 && dm.G % (cpw / N) == 0) GS = cpw / N;
    if (GS == 0) {
        md_set_error("costvol: no channel split for C=%d G=%d (need a multiple of 4 channels per workgroup)", dm.C, dm.G);
        return MD_EINVAL;
    }
    const int CPW = GS * N;
    constexpr int TW = 32;  // 64-wide tiles measured slower at every setting (w=160 wastes 17 % of the lanes)
    const int TH = 256 / TW;
    dm.tiles_x = md_cdiv(dm.w, TW);
    const int tiles = dm.tiles_x * md_cdiv(dm.h, TH);
    const int splits = dm.G / GS;
    // Grid: every work item (sample, tile, channel split) is cut into `dsplit` equal hypothesis slices, one
    // workgroup each, with dsplit the largest divisor of D that keeps all workgroups resident at once (slots =
    // 256 CUs x workgroups per CU by LDS, <= 8 by waves) and leaves >= 8 hypotheses per slice.  Measured at
    // B=6, 48x160, D=96: slices that straddle two items (a perfectly balanced linear split) cost more in window
    // re-staging than the balance wins (72 vs 60 us), so slices never cross items unless MD_COSTVOL_NWG forces it
    // (exception: the channels-last forward, below).
    dm.tiles = tiles;
    dm.splits = splits;
    dm.items = dm.B * tiles * splits;
    const long long total = (long long)dm.items * dm.D;
    long long nwg = BWD ? MD_COSTVOL_NWG_BWD : MD_COSTVOL_NWG;
    if (nwg <= 0) {
        const int wp = (TW + (CPW >= 32 ? 8 : 16)) * (TH + (CPW >= 32 ? 4 : 8));
        const long long lds = (long long)wp * CPW * 4 * (BWD ? 2 : 1) + ITV_MAX * 4 + 64 +
                              ((!BWD && dm.sg == 1) ? 4 * 64 * (GS / 4) * 16 : 0);
        long long per_cu = (160 * 1024) / lds;
        if (per_cu > 8) per_cu = 8;
        if (per_cu < 1) per_cu = 1;
        const long long slots = 256 * per_cu;
        int dsplit = BWD ? MD_COSTVOL_DSPLIT_BWD : MD_COSTVOL_DSPLIT;
        if (dsplit <= 0) {
            dsplit = 1;
            for (int c = 2; c <= dm.D / 8; ++c)
                if (dm.D % c == 0 && (long long)dm.items * c <= slots) dsplit = c;
        }
        while (dm.D % dsplit != 0) --dsplit;
        nwg = (long long)dm.items * dsplit;
        // Channels-last forward only: LDS allows 2 workgroups per CU, and item-aligned slices rarely fill the slots
        // evenly (B=6, 48x160: 360 workgroups on 512 slots, so 104 CUs carry two and 152 carry one).  A linear
        // split over exactly `slots` workgroups measured 70.2 us against 74.5 us (same-buffer medians of 8 interleaved runs,
        // 7 of 8 pairs faster) despite ~1/3 of the workgroups staging two windows.  The planar kernels measured
        // the opposite (72 vs 60 us) and keep item-aligned slices.
        if (cl_out && nwg < slots && total >= slots * 8 && MD_COSTVOL_CL_FILL) nwg = slots;
    }
    if (nwg > total) nwg = total;
    if (nwg * ITV_MAX < total) nwg = (total + ITV_MAX - 1) / ITV_MAX;  // a share fits the interval table
    const dim3 grid((unsigned)nwg), block(256);
    // channels-last volume (sg == 1): 16-byte stores of 4 consecutive groups, given 16-byte alignment
    const bool nhwc = !BWD && dm.sg == 1 && dm.G % 4 == 0 && dm.sb % 4 == 0 && dm.sd % 4 == 0 && dm.sp % 4 == 0 &&
                      ((uintptr_t)q.out % 16) == 0;
    (void)nhwc;
    bool launched = true;
    hipEvent_t ev0, ev1;
    md_timing_pair(tname, &ev0, &ev1);

#define MD_CV_LAUNCH(GS_, N_, TW_, F_)                                                                                \
    do {                                                                                                              \
        if (BWD)                                                                                                      \
            hipExtLaunchKernelGGL((costvol_bwd_kernel<GS_, N_, TW_, F_>), grid, block, 0, stream, ev0, ev1, 0, q.gout,   \
                                  q.ref, q.src, q.K, q.invK, q.pose, q.hyp, q.prior, q.ztrans, q.d_ref, q.d_src, dm);   \
        else if (nhwc && (GS_) % 4 == 0)                                                                              \
            hipExtLaunchKernelGGL((costvol_fwd_nhwc_kernel<((GS_) % 4 == 0 ? (GS_) : 4), N_, TW_, F_>), grid, block, 0, \
                               stream, ev0, ev1, 0, q.ref, q.src, q.K, q.invK, q.pose, q.hyp, q.prior, q.ztrans, q.out, dm);       \
        else                                                                                                          \
            hipExtLaunchKernelGGL((costvol_fwd_kernel<GS_, N_, TW_, F_>), grid, block, 0, stream, ev0, ev1, 0, q.ref, q.src, q.K,     \
                               q.invK, q.pose, q.hyp, q.prior, q.ztrans, q.out, dm);                                  \
    } while (0)
#define MD_CV_F(GS_, N_, TW_)                        \
    do {                                             \
        if (q.hyp) MD_CV_LAUNCH(GS_, N_, TW_, false); \
        else MD_CV_LAUNCH(GS_, N_, TW_, true);       \
    } while (0)
#define MD_CV_TW(GS_, N_) MD_CV_F(GS_, N_, 32)

    switch (N * 100 + CPW) {
        case 104: MD_CV_TW(4, 1); break;
        case 108: MD_CV_TW(8, 1); break;
        case 116: MD_CV_TW(16, 1); break;
        case 132: MD_CV_TW(32, 1); break;
        case 204: MD_CV_TW(2, 2); break;
        case 208: MD_CV_TW(4, 2); break;
        case 216: MD_CV_TW(8, 2); break;
        case 232: MD_CV_TW(16, 2); break;
        case 404: MD_CV_TW(1, 4); break;
        case 408: MD_CV_TW(2, 4); break;
        case 416: MD_CV_TW(4, 4); break;
        case 432: MD_CV_TW(8, 4); break;
        case 808: MD_CV_TW(1, 8); break;
        case 816: MD_CV_TW(2, 8); break;
        case 832: MD_CV_TW(4, 8); break;
        default: launched = false;
    }
    if (!launched) {
        md_set_error("costvol: no kernel for C/G=%d with %d channels per workgroup", N, CPW);
        return MD_EINVAL;
    }
#undef MD_CV_TW
#undef MD_CV_F
#undef MD_CV_LAUNCH
    MD_CHECK_LAUNCH(BWD ? "md_costvol_bwd" : "md_costvol_fwd");
    return MD_OK;
}

int check_common(const char *fn, const void *ref, const void *src, const void *K, const void *invK, const void *pose,
                 const void *hyp, const void *prior, int sched_type, int B, int C, int G, int h, int w, int D) {
    MD_REQUIRE(ref && src && K && invK && pose, "%s: null tensor argument", fn);
    MD_REQUIRE(hyp || prior, "%s: need either hyp [B,D,h,w] or prior [B,1,h,w]", fn);
    MD_REQUIRE(B > 0 && C > 0 && G > 0 && h > 1 && w > 1 && D > 0, "%s: bad dims B=%d C=%d G=%d h=%d w=%d D=%d", fn, B,
               C, G, h, w, D);
    MD_REQUIRE(C % G == 0, "%s: C=%d not divisible by G=%d", fn, C, G);
    MD_REQUIRE(hyp || D > 1, "%s: the fused schedule needs D > 1", fn);
    MD_REQUIRE(sched_type >= 0 && sched_type <= 2, "%s: bad schedule type %d", fn, sched_type);
    MD_REQUIRE(B <= 16384, "%s: batch too large", fn);
    return MD_OK;
}

}  // namespace

extern "C" int MD_CV_NAME(md_costvol_fwd)(const abi_io_t *ref_, const abi_io_t *src_, const float *K, const float *invK,
                              const float *pose, const float *hyp, const float *prior, const float *ztrans,
                              float scale_fac, int sched_type, int B, int C, int G, int h, int w, int D, int feat_cl,
                              abi_io_t *out_, long long out_sb, long long out_sd, long long out_sg, long long out_sp,
                              unsigned flags, md_stream_t stream) {
    const io_t *ref = reinterpret_cast<const io_t *>(ref_), *src = reinterpret_cast<const io_t *>(src_);
    io_t *out = reinterpret_cast<io_t *>(out_);
    int rc = check_common("md_costvol_fwd", ref, src, K, invK, pose, hyp, prior, sched_type, B, C, G, h, w, D);
    if (rc) return rc;
    MD_REQUIRE(out, "md_costvol_fwd: null output");
    MD_REQUIRE((flags & ~(unsigned)MD_CV_FINE_SLICES) == 0, "md_costvol_fwd: unknown flag bits 0x%x", flags);
    CvPtrs q{};
    q.flags = flags;
    q.ref = ref; q.src = src; q.K = K; q.invK = invK; q.pose = pose; q.hyp = hyp; q.prior = prior; q.ztrans = ztrans;
    q.out = out;
    CvDims dm{};
    dm.scale_fac = scale_fac; dm.sched_type = sched_type;
    dm.B = B; dm.C = C; dm.G = G; dm.h = h; dm.w = w; dm.D = D;
    dm.sb = out_sb; dm.sd = out_sd; dm.sg = out_sg; dm.sp = out_sp;
    dm.fcl = feat_cl != 0;
    return launch<false>(q, dm, (hipStream_t)stream);
}

extern "C" int MD_CV_NAME(md_costvol_bwd)(const abi_io_t *gout_, long long g_sb, long long g_sd, long long g_sg, long long g_sp,
                              const abi_io_t *ref_,
                              const abi_io_t *src_, const float *K, const float *invK, const float *pose,
                              const float *hyp, const float *prior, const float *ztrans, float scale_fac,
                              int sched_type, int B, int C, int G, int h, int w, int D, int feat_cl, float *d_ref,
                              float *d_src, unsigned flags, unsigned long long *census, const long long *shares, int n_shares,
                              unsigned *cost, md_stream_t stream) {
    const io_t *gout = reinterpret_cast<const io_t *>(gout_), *ref = reinterpret_cast<const io_t *>(ref_),
               *src = reinterpret_cast<const io_t *>(src_);
    int rc = check_common("md_costvol_bwd", ref, src, K, invK, pose, hyp, prior, sched_type, B, C, G, h, w, D);
    if (rc) return rc;
    MD_REQUIRE(gout && d_ref && d_src, "md_costvol_bwd: null gradient tensor");
    MD_REQUIRE((flags & ~(unsigned)MD_CV_GATHER_TABLE) == 0, "md_costvol_bwd: unknown flag bits 0x%x", flags);
    MD_REQUIRE(((uintptr_t)census % 8) == 0, "md_costvol_bwd: census must be 8-byte aligned");
    CvPtrs q{};
    MD_REQUIRE(((uintptr_t)shares % 8) == 0 && ((uintptr_t)cost % 4) == 0 && n_shares >= 0, "md_costvol_bwd: misaligned shares / cost");
    q.flags = flags; q.census = census;
    q.shares = n_shares > 0 ? shares : nullptr; q.n_shares = n_shares; q.cost = cost;
    q.gout = gout; q.ref = ref; q.src = src; q.K = K; q.invK = invK; q.pose = pose; q.hyp = hyp; q.prior = prior;
    q.ztrans = ztrans; q.d_ref = d_ref; q.d_src = d_src;
    CvDims dm{};
    dm.scale_fac = scale_fac; dm.sched_type = sched_type;
    dm.B = B; dm.C = C; dm.G = G; dm.h = h; dm.w = w; dm.D = D;
    dm.sb = g_sb; dm.sd = g_sd; dm.sg = g_sg; dm.sp = g_sp;
    dm.fcl = feat_cl != 0;
    // the launch paths zero what they accumulate into (one fill when the two gradients sit back to back, as ops.py allocates
    // them; none for d_ref when the launch stores it)
    return launch<true>(q, dm, (hipStream_t)stream);
}

#if MD_CV_IO == 0
extern "C" int md_costvol_bwd_plan(int B, int C, int G, int h, int w, int D, int feat_cl, int *items, int *workgroups) {
    MD_REQUIRE(items && workgroups, "md_costvol_bwd_plan: null output");
    MD_REQUIRE(B > 0 && C > 0 && G > 0 && C % G == 0 && h > 1 && w > 1 && D > 0, "md_costvol_bwd_plan: bad dims");
    CvPtrs q{};
    int out[2] = {0, 0};
    q.plan_out = out;
    CvDims dm{};
    dm.B = B; dm.C = C; dm.G = G; dm.h = h; dm.w = w; dm.D = D;
    dm.sg = 1; dm.sp = G; dm.sd = (long long)h * w * G; dm.sb = (long long)D * h * w * G;   // the channels-last volume (B,D,h,w,G)
    dm.fcl = feat_cl != 0;
    const int rc = launch<true>(q, dm, nullptr);
    *items = out[0];
    *workgroups = out[1];
    return rc;
}
#endif
