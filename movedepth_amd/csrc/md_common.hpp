// Shared host/device helpers for libmovedepth_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/movedepth_hip.h"

// ---------------------------------------------------------------- error plumbing
void md_set_error(const char *fmt, ...);

#define MD_REQUIRE(cond, ...)         \
    do {                              \
        if (!(cond)) {                \
            md_set_error(__VA_ARGS__); \
            return MD_EINVAL;         \
        }                             \
    } while (0)

#define MD_CHECK_LAUNCH(name)                                                   \
    do {                                                                        \
        hipError_t e_ = hipGetLastError();                                      \
        if (e_ != hipSuccess) {                                                 \
            md_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return MD_ELAUNCH;                                                  \
        }                                                                       \
    } while (0)

#define MD_CHECK_HIP(expr)                                                    \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) {                                               \
            md_set_error("%s failed: %s", #expr, hipGetErrorString(e_));      \
            return MD_ELAUNCH;                                                \
        }                                                                     \
    } while (0)

unsigned long long *md_stats_buffer();  // capi.hip: device counters for md_costvol_stats, null when off
// kernel timing hook (capi.hip): start / stop events for hipExtLaunchKernelGGL, null unless md_kernel_timing_enable(1)
void md_timing_pair(const char *name, hipEvent_t *start, hipEvent_t *stop);

// Launch with the measurement hook of capi.hip: when md_kernel_timing_enable(1) is on, a start / stop event pair is tied to THIS dispatch
// (its own begin / end timestamps, what rocprofv3's kernel trace reads) and recorded under `tname`; otherwise a plain launch.
#define MD_LAUNCH_TIMED(tname, kern, grid, block, lds, s, ...)                                   \
    do {                                                                                         \
        hipEvent_t e0_, e1_;                                                                     \
        md_timing_pair(tname, &e0_, &e1_);                                                       \
        hipExtLaunchKernelGGL(kern, grid, block, lds, s, e0_, e1_, 0, __VA_ARGS__);              \
    } while (0)

static inline int md_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- geometry (device)
// Per-sample camera constants, wave-uniform (live in SGPRs).
struct CamMats {
    float P[12];  // (K @ T)[:3,:]          layers.py:608
    float iK[9];  // inv_K[:3,:3]           layers.py:582
};

__device__ __forceinline__ CamMats md_load_cam(const float *__restrict__ K, const float *__restrict__ invK,
                                               const float *__restrict__ T) {
    CamMats m;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s = fmaf(K[i * 4 + k], T[k * 4 + j], s);  // explicit: every kernel forms the same P
            m.P[i * 4 + j] = s;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) m.iK[i * 3 + j] = invK[i * 4 + j];
    return m;
}

// Operation order of the reference's three matrix products (layers.py:582, 608, 610 are torch.matmul calls; what arithmetic
// that is, is the BLAS's business).  The committed fixtures -- the reference's own outputs -- decide: of all combinations of
// {rounded product-and-sum, fused multiply-add chain ascending / descending} exactly one reproduces the 20,720 pixel coordinates
// of tests/golden/{warp_small, warp_border, geometry, losses_mono}.npz bit for bit (tools/diag/op_order_search.py):
//   P = K @ T                                   -- every product and sum rounded on its own (md_load_cam_plain);
//   inv_K[:3,:3] @ (x,y,1), P @ (X,Y,Z,1)       -- acc = a0 b0, then acc = fma(a_k, b_k, acc), k ascending (md_ray, md_project*).
// The photometric kernels (warp.hip, photo.hip) and the CPU oracle (project_pixel) both do exactly that, so their sample
// positions -- and with them every texel decision of the border-mode grid_sample -- are BIT-EQUAL to each other and to the
// reference's (asserted at 192x640 in tests/test_hip_parity.py, tests/test_photo_fused.py, and on the fixtures).
__device__ __forceinline__ CamMats md_load_cam_plain(const float *__restrict__ K, const float *__restrict__ invK,
                                                     const float *__restrict__ T) {
#pragma clang fp contract(off)
    CamMats m;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s = s + K[i * 4 + k] * T[k * 4 + j];
            m.P[i * 4 + j] = s;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) m.iK[i * 3 + j] = invK[i * 4 + j];
    return m;
}

// Ray through pixel (x, y): inv_K[:3,:3] @ (x, y, 1).  No contraction (here and in md_project*): only the fused operations
// written out, in the reference's order (see above).
__device__ __forceinline__ void md_ray(const CamMats &m, float x, float y, float &r0, float &r1, float &r2) {
#pragma clang fp contract(off)
    r0 = fmaf(m.iK[1], y, m.iK[0] * x) + m.iK[2];
    r1 = fmaf(m.iK[4], y, m.iK[3] * x) + m.iK[5];
    r2 = fmaf(m.iK[7], y, m.iK[6] * x) + m.iK[8];
}
// row i of P @ (X, Y, Z, 1) in the reference's order (see above)
__device__ __forceinline__ float md_dot4(const float *P, float X, float Y, float Z) {
#pragma clang fp contract(off)
    return fmaf(P[2], Z, fmaf(P[1], Y, P[0] * X)) + P[3];
}

// Backproject at depth d and project: returns the un-normalised sample position (ix, iy) grid_sample uses,
// following the reference's operation order (layers.py:583, 610-620, then grid_sample's un-normalise).
struct Proj {
    float ix, iy;  // sample position in source pixels (before any border clipping)
    float gx, gy;  // normalised grid coordinates in [-1,1]
    float u, v, zz;
    float X, Y, Z;
};

__device__ __forceinline__ Proj md_project(const CamMats &m, float r0, float r1, float r2, float d, int w, int h) {
#pragma clang fp contract(off)
    Proj p;
    p.X = d * r0;
    p.Y = d * r1;
    p.Z = d * r2;
    float c0 = md_dot4(m.P, p.X, p.Y, p.Z);
    float c1 = md_dot4(m.P + 4, p.X, p.Y, p.Z);
    float c2 = md_dot4(m.P + 8, p.X, p.Y, p.Z);
    p.zz = c2 + 1e-7f;
    p.u = c0 / p.zz;
    p.v = c1 / p.zz;
    p.gx = (p.u / (float)(w - 1) - 0.5f) * 2.f;
    p.gy = (p.v / (float)(h - 1) - 0.5f) * 2.f;
    p.ix = ((p.gx + 1.f) / 2.f) * (float)(w - 1);
    p.iy = ((p.gy + 1.f) / 2.f) * (float)(h - 1);
    return p;
}

// md_project with the two divisions by the wave-uniform (w-1), (h-1) done through their correctly rounded reciprocals
// rw = 1/(w-1), rh = 1/(h-1) (Markstein: q = x r, RN(q + (x - q y) r) = the IEEE quotient): same values, 3 instructions each.
__device__ __forceinline__ Proj md_project_r(const CamMats &m, float r0, float r1, float r2, float d, float wm1, float hm1,
                                             float rw, float rh) {
#pragma clang fp contract(off)
    Proj p;
    p.X = d * r0;
    p.Y = d * r1;
    p.Z = d * r2;
    float c0 = md_dot4(m.P, p.X, p.Y, p.Z);
    float c1 = md_dot4(m.P + 4, p.X, p.Y, p.Z);
    float c2 = md_dot4(m.P + 8, p.X, p.Y, p.Z);
    p.zz = c2 + 1e-7f;
    p.u = c0 / p.zz;
    p.v = c1 / p.zz;
    const float qx = p.u * rw, qy = p.v * rh;
    p.gx = (fmaf(fmaf(-qx, wm1, p.u), rw, qx) - 0.5f) * 2.f;
    p.gy = (fmaf(fmaf(-qy, hm1, p.v), rh, qy) - 0.5f) * 2.f;
    p.ix = ((p.gx + 1.f) / 2.f) * wm1;
    p.iy = ((p.gy + 1.f) / 2.f) * hm1;
    return p;
}

// v_rcp_f32 (1 ulp) + one Newton step: within 1 ulp of the IEEE quotient at a fifth of its instruction count.
__device__ __forceinline__ float md_rcp_nr(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.f), r, r);
}

// Depth-range schedule (layers.py:261-279 / 375-393), split into its per-pixel constants and the per-bin
// evaluation so the cost-volume kernels can keep the constants in registers across the hypothesis loop.
// The standalone schedule kernel and the fused in-kernel schedule share this code: bit-identical hypotheses.
struct HypConst {
    float a, b;  // inverse: hyp = 1/(a + b*itv), a = 1/dmax, b = 1/dmin - 1/dmax;  linear/log: hyp = a + b*itv
};
__device__ __forceinline__ HypConst md_hyp_const(float c, float one_pf, int type) {
    const float dmin = c / one_pf, dmax = c * one_pf;
    HypConst h;
    if (type == MD_SCHED_INVERSE) { h.a = 1.f / dmax; h.b = 1.f / dmin - 1.f / dmax; }
    else { h.a = dmin; h.b = dmax - dmin; }
    return h;
}
// interval position of bin k: k/(D-1), or the reference's fp32 log spacing (layers.py:274-278)
__device__ __forceinline__ float md_hyp_itv(int k, int D, int type) {
    return (type == MD_SCHED_LOG) ? expf(logf(0.1f) + logf(1.f / 0.1f) * (float)k / (float)(D - 1))
                                  : (float)k / (float)(D - 1);
}
__device__ __forceinline__ float md_hyp_eval(const HypConst &h, float itv, int type) {
    // two roundings, as the reference's `a + b * itv` (layers.py:268-270), at every call site: the kernels that evaluate a
    // hypothesis twice (hot loop / window-miss loop) rely on bit-identical results
#pragma clang fp contract(off)
    const float v = h.a + h.b * itv;
    return type == MD_SCHED_INVERSE ? md_rcp_nr(v) : v;
}
__device__ __forceinline__ float md_hypothesis(float c, float one_pf, int k, int D, int type) {
    return md_hyp_eval(md_hyp_const(c, one_pf, type), md_hyp_itv(k, D, type), type);
}

// Hot-loop projection for the plane sweep: same operation order as md_project, with the two perspective
// divides and the two /(size-1) done as multiplications by reciprocals (rw = 1/(w-1), rh = 1/(h-1) are IEEE
// quotients computed once).  Deviates from md_project by <= ~2 ulp of the pixel coordinate.
__device__ __forceinline__ void md_project_fast(const CamMats &m, float r0, float r1, float r2, float d, float wm1,
                                                float hm1, float rw, float rh, float &ix, float &iy) {
    const float X = d * r0, Y = d * r1, Z = d * r2;
    const float c0 = m.P[0] * X + m.P[1] * Y + m.P[2] * Z + m.P[3];
    const float c1 = m.P[4] * X + m.P[5] * Y + m.P[6] * Z + m.P[7];
    const float c2 = m.P[8] * X + m.P[9] * Y + m.P[10] * Z + m.P[11];
    const float rz = md_rcp_nr(c2 + 1e-7f);
    const float gx = ((c0 * rz) * rw - 0.5f) * 2.f;
    const float gy = ((c1 * rz) * rh - 0.5f) * 2.f;
    ix = ((gx + 1.f) * 0.5f) * wm1;
    iy = ((gy + 1.f) * 0.5f) * hm1;
}

// Bilinear tap set.  x0,y0 = north-west tap; wx1, wy1 = weights of the east / south taps.
struct Tap {
    int x0, y0;
    float wx1, wy1;
};

__device__ __forceinline__ Tap md_make_tap(float ix, float iy, int w, int h) {
    Tap t;
    float fx = floorf(ix), fy = floorf(iy);
    t.wx1 = ix - fx;
    t.wy1 = iy - fy;
    // keep the int conversion defined for wild coordinates (those taps are out of range anyway)
    bool bad = !(ix == ix) || !(iy == iy);
    fx = fminf(fmaxf(fx, -2.f), (float)w);
    fy = fminf(fmaxf(fy, -2.f), (float)h);
    t.x0 = bad ? -2 : (int)fx;
    t.y0 = bad ? -2 : (int)fy;
    if (bad) { t.wx1 = 0.f; t.wy1 = 0.f; }
    return t;
}

// ---------------------------------------------------------------- reductions
// Sum over the 64 lanes with DPP adds only (no ds_bpermute: __shfl_xor goes through the LDS crossbar, and a kernel reducing 24
// values per wave that way became LDS-bound).  Butterflies inside each row of 16 lanes (quad_perm, row_half_mirror, row_mirror),
// then row_bcast15 / row_bcast31 carry the row totals up: lane 63 holds the sum, returned to every lane by a readlane.
__device__ __forceinline__ float md_wave_sum_dpp(float v) {
#define MD_DPP_ADD(ctrl, rmask)                                                                                          \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xF, false))
    MD_DPP_ADD(0xB1, 0xF);   // quad_perm [1,0,3,2]
    MD_DPP_ADD(0x4E, 0xF);   // quad_perm [2,3,0,1]
    MD_DPP_ADD(0x141, 0xF);  // row_half_mirror
    MD_DPP_ADD(0x140, 0xF);  // row_mirror: every lane of a row holds the row's sum
    MD_DPP_ADD(0x142, 0xA);  // row_bcast15 into rows 1 and 3
    MD_DPP_ADD(0x143, 0xC);  // row_bcast31 into rows 2 and 3
#undef MD_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Mixed: float butterflies inside each row of 16 lanes (neighbouring pixels: terms of one sign and size, 4 roundings), double
// from there on (two DPP moves + one v_add_f64 per step) -- for image-wide sums whose terms cancel across regions.
__device__ __forceinline__ double md_wave_sum_dpp_f16_d(float f) {
#define MD_DPP_ADDF(ctrl) f += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), ctrl, 0xF, 0xF, false))
    MD_DPP_ADDF(0xB1); MD_DPP_ADDF(0x4E); MD_DPP_ADDF(0x141); MD_DPP_ADDF(0x140);
#undef MD_DPP_ADDF
    double v = (double)f;
#define MD_DPP_ADD(ctrl, rmask)                                                                                          \
    v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rmask, 0xF, false),                    \
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rmask, 0xF, false))
    MD_DPP_ADD(0x142, 0xA);
    MD_DPP_ADD(0x143, 0xC);
#undef MD_DPP_ADD
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// all double: for sums whose terms cancel to 1e-3 of their magnitude already inside a wave
__device__ __forceinline__ double md_wave_sum_dpp(double v) {
#define MD_DPP_ADD(ctrl, rmask)                                                                                          \
    v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rmask, 0xF, false),                    \
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rmask, 0xF, false))
    MD_DPP_ADD(0xB1, 0xF);
    MD_DPP_ADD(0x4E, 0xF);
    MD_DPP_ADD(0x141, 0xF);
    MD_DPP_ADD(0x140, 0xF);
    MD_DPP_ADD(0x142, 0xA);
    MD_DPP_ADD(0x143, 0xC);
#undef MD_DPP_ADD
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

__device__ __forceinline__ float md_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// min / max over the 64 lanes with DPP moves only (the butterflies of md_wave_sum_dpp; `old` = the lane's own value, so a lane a
// row mask leaves out keeps it: min and max are idempotent).  All 64 lanes must be active.  The __shfl_xor versions below go
// through the LDS crossbar: 24 ds_bpermute round trips per window fit in the plane-sweep kernels' staging.
#define MD_DPP_RED(name, OP)                                                                                              \
    __device__ __forceinline__ int name(int v) {                                                                          \
        v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));                                              \
        v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));                                              \
        v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));                                             \
        v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));                                             \
        v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xF, false));                                             \
        v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xF, false));                                             \
        return __builtin_amdgcn_readlane(v, 63);                                                                          \
    }
MD_DPP_RED(md_wave_min_dpp, min)
MD_DPP_RED(md_wave_max_dpp, max)
#undef MD_DPP_RED

__device__ __forceinline__ int md_wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int md_wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
