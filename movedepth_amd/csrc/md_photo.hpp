// Device helpers shared by the photometric kernels (warp.hip, ssim.hip, photo.hip).
#pragma once
#include "md_common.hpp"

namespace mdp {

constexpr float kC1 = 0.01f * 0.01f;  // SSIM constants, layers.py:658-659
constexpr float kC2 = 0.03f * 0.03f;

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

struct Clip {
    float ix, iy, gmx, gmy;
};

// grid_sample 'border': clip to [0, size-1]; the borders themselves count as clipped (zero grid gradient)
__device__ __forceinline__ Clip clip_border(float ix, float iy, int W, int H) {
    Clip c;
    c.gmx = (float)(W - 1) / 2.f;
    c.gmy = (float)(H - 1) / 2.f;
    if (!(ix > 0.f)) { ix = 0.f; c.gmx = 0.f; }
    else if (ix >= (float)(W - 1)) { ix = (float)(W - 1); c.gmx = 0.f; }
    if (!(iy > 0.f)) { iy = 0.f; c.gmy = 0.f; }
    else if (iy >= (float)(H - 1)) { iy = (float)(H - 1); c.gmy = 0.f; }
    c.ix = ix; c.iy = iy;
    return c;
}

// x / y for a divisor whose correctly rounded reciprocal r = RN(1/y) is at hand (a constant, or wave-uniform and computed once):
// q = RN(x r), rem = x - q y (exact in an fma), RN(q + rem r) is the correctly rounded quotient (Markstein) -- the value an IEEE
// division returns, in 3 instructions instead of the ~12 of v_div_scale / v_rcp / 5 fma / v_div_fmas / v_div_fixup.  The window
// means (x / 9, five per SSIM value) were 40 % of the fused forward's VALU instructions as IEEE divisions.
__device__ __forceinline__ float div_by(float x, float y, float r) {
    const float q = x * r;
    return fmaf(fmaf(-q, y, x), r, q);
}
__device__ __forceinline__ float div9(float x) { return div_by(x, 9.f, 1.f / 9.f); }

// F.interpolate(bilinear, align_corners=False) source taps of output index o (trainer.py:512)
__device__ __forceinline__ void interp_idx(int o, int in, int out, int &i0, int &i1, float &l1) {
#pragma clang fp contract(off)
    const float scale = (float)in / (float)out;
    float s = scale * ((float)o + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    int a = (int)s;
    if (a > in - 1) a = in - 1;
    i0 = a;
    i1 = a < in - 1 ? a + 1 : a;
    l1 = s - (float)a;
}

// the same with scale = (float)in / (float)out hoisted by the caller (wave-uniform: one IEEE division per kernel, not two per pixel)
__device__ __forceinline__ void interp_idx_s(int o, int in, float scale, int &i0, int &i1, float &l1) {
#pragma clang fp contract(off)
    float s = scale * ((float)o + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    int a = (int)s;
    if (a > in - 1) a = in - 1;
    i0 = a;
    i1 = a < in - 1 ? a + 1 : a;
    l1 = s - (float)a;
}

// F.interpolate's four taps in the reference's order (the oracle's mdo_resize_bilinear_fwd; pinned by the depth maps of
// tests/golden/losses_mono.npz): the north-east product first, then nw, sw, se accumulated with fused multiply-adds
__device__ __forceinline__ float interp4(float nw, float ne, float sw, float se, float lx, float ly) {
#pragma clang fp contract(off)
    const float wy0 = 1.f - ly, wx0 = 1.f - lx;
    float o = (wy0 * lx) * ne;
    o = fmaf(wy0 * wx0, nw, o);
    o = fmaf(ly * wx0, sw, o);
    o = fmaf(ly * lx, se, o);
    return o;
}

// upsampled disparity at (y, x) of the full-resolution grid -> scaled disparity sd (depth = 1 / sd, layers.py:400-409)
// (sx = (float)w / (float)W, sy = (float)h / (float)H: IEEE quotients of launch constants, formed once on the host)
__device__ __forceinline__ float disp_up_sd_s(const float *__restrict__ s, int h, int w, float sy, float sx, int y, int x, float min_disp,
                                              float max_disp) {
#pragma clang fp contract(off)
    int x0, x1, y0, y1;
    float lx, ly;
    interp_idx_s(x, w, sx, x0, x1, lx);
    interp_idx_s(y, h, sy, y0, y1, ly);
    return min_disp + (max_disp - min_disp) * interp4(s[y0 * w + x0], s[y0 * w + x1], s[y1 * w + x0], s[y1 * w + x1], lx, ly);
}
__device__ __forceinline__ float disp_up_sd(const float *__restrict__ s, int h, int w, int H, int W, int y, int x,
                                            float min_disp, float max_disp) {
    // no contraction: the oracle's (mdo_resize_bilinear_fwd, mdo_disp_to_depth) operations one by one -- the depth feeds the
    // projection, whose sample positions are asserted bit-equal to the oracle's
#pragma clang fp contract(off)
    int x0, x1, y0, y1;
    float lx, ly;
    interp_idx_s(x, w, (float)w / (float)W, x0, x1, lx);   // the quotients are wave-uniform: hoisted out of any loop
    interp_idx_s(y, h, (float)h / (float)H, y0, y1, ly);
    return min_disp + (max_disp - min_disp) * interp4(s[y0 * w + x0], s[y0 * w + x1], s[y1 * w + x0], s[y1 * w + x1], lx, ly);
}

struct Moments {
    float mux, muy, ex2, ey2, exy;
};

// Two values per register pair.  v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 round each half exactly as the one-value
// instructions do, so arithmetic written on pairs is bit-identical to the same arithmetic written value by value -- at half the
// vector-ALU instructions, PROVIDED the pairs are born adjacent (the two halves of a 16-byte LDS record, two accumulators that
// are always updated together).  photo.hip is compiled with -fno-slp-vectorize: the compiler's own pairing of this code spent
// one v_mov per packed operation gathering operands (500 of the backward's 2600 vector instructions) and 20 more registers.
#ifndef MD_PHOTO_FAST_DIV
#define MD_PHOTO_FAST_DIV 1   // 0: every quotient of the fused kernels as a plain `/` (A/B builds)
#endif
using v2f = float __attribute__((ext_vector_type(2)));
using v4f = float __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v2f md_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float md_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ v2f div9(v2f x) {   // div_by per half
#pragma clang fp contract(off)
    const v2f q = x * (1.f / 9.f);
    return md_fma(md_fma(-q, (v2f)9.f, x), (v2f)(1.f / 9.f), q);
}

// IEEE quotient n / d as the compiler's own expansion computes it -- v_rcp_f32, one Newton step on the reciprocal, q = n r, two
// residual corrections (each an exact fma) -- WITHOUT that expansion's range scaling (two v_div_scale_f32, v_div_fmas_f32), which
// is the identity unless d is subnormal or >= 2^126, |n| < 2^-103, or the exponents of n and d differ by 96 or more (ISA:
// V_DIV_SCALE_F32): inside that range the bits are those of `n / d` (asserted against the per-operation kernels, which divide
// with `/`, in tests/test_photo_fused.py).  What the caller gains: the steps pack (two quotients per instruction), a divisor
// shared by several numerators is inverted once, and 11 instructions become 8.  md_div_fixup restores the IEEE results for zero /
// infinite / NaN operands where a caller can meet them.
__device__ __forceinline__ float md_rcp_newton(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.f), r, r);
}
__device__ __forceinline__ v2f md_rcp_newton(v2f d) {
    const v2f r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    return md_fma(md_fma(-d, r, (v2f)1.f), r, r);
}
template <class T>
__device__ __forceinline__ T md_div_core(T n, T d, T r /* md_rcp_newton(d) */) {
#pragma clang fp contract(off)
    const T q0 = n * r;
    const T q1 = md_fma(md_fma(-d, q0, n), r, q0);
    return md_fma(md_fma(-d, q1, n), r, q1);
}
__device__ __forceinline__ float md_div_fixup(float q, float d, float n) { return __builtin_amdgcn_div_fixupf(q, d, n); }

// md_project_r with the x and y rows of P = (K T)[:3] side by side: every operation of the two rows is the same instruction on a
// register pair, and the two quotients by zz share its reciprocal.  Operation for operation md_project_r (bit-equal sample grid;
// tests/test_photo_fused.py against md_warp_fwd and against the reference's fixtures).
struct CamPk {
    v2f Pxy[4];   // (P[0][j], P[1][j])
    float Pz[4];  // P[2][j]
};
struct ProjPk {
    v2f uv, g, i;   // (u, v) = cam.xy / zz; normalised grid coordinates; sample position in source pixels
    float zz;
};
__device__ __forceinline__ ProjPk md_project_pk(const CamPk &m, float r0, float r1, float r2, float d, v2f wh1 /* (W-1, H-1) */,
                                                v2f rwh /* RN(1 / wh1) */) {
#pragma clang fp contract(off)
    ProjPk p;
    const float X = d * r0, Y = d * r1, Z = d * r2;
    const v2f c01 = md_fma(m.Pxy[2], (v2f)Z, md_fma(m.Pxy[1], (v2f)Y, m.Pxy[0] * X)) + m.Pxy[3];
    const float c2 = fmaf(m.Pz[2], Z, fmaf(m.Pz[1], Y, m.Pz[0] * X)) + m.Pz[3];
    p.zz = c2 + 1e-7f;
#if MD_PHOTO_FAST_DIV
    // |zz| is 0 or >= 2^-47 (the spacing of floats at 1e-7) and |cam.xy| is far below 2^49 for any pose a frame can have: no
    // scaling case; zz == 0 (and infinities, NaN) through v_div_fixup_f32 as in the compiler's expansion
    const v2f q = md_div_core(c01, (v2f)p.zz, (v2f)md_rcp_newton(p.zz));
    p.uv = (v2f){md_div_fixup(q.x, p.zz, c01.x), md_div_fixup(q.y, p.zz, c01.y)};
#else
    p.uv = c01 / p.zz;
#endif
    const v2f qn = p.uv * rwh;
    p.g = (md_fma(md_fma(-qn, wh1, p.uv), rwh, qn) - 0.5f) * 2.f;   // div_by(uv, wh1, rwh)
    p.i = ((p.g + 1.f) / 2.f) * wh1;
    return p;
}

template <class T>
struct MomentsT {
    T mux, muy, ex2, ey2, exy;
};

// ssim_from on one value or on a pair: the same operations in the same order
template <class T>
__device__ __forceinline__ T ssim_from_t(const MomentsT<T> &m) {
#pragma clang fp contract(off)
    const T sx = m.ex2 - m.mux * m.mux, sy = m.ey2 - m.muy * m.muy, sxy = m.exy - m.mux * m.muy;
    const T n = (2.f * m.mux * m.muy + kC1) * (2.f * sxy + kC2);
    const T d = (m.mux * m.mux + m.muy * m.muy + kC1) * (sx + sy + kC2);
#if MD_PHOTO_FAST_DIV
    // d >= C1 (C2 - rounding noise) ~ 2^-24 and |n| / d < 2^30 for images of magnitude up to 2^20: no scaling case
    return (1.f - md_div_core(n, d, md_rcp_newton(d))) / 2.f;
#else
    return (1.f - n / d) / 2.f;
#endif
}

__device__ __forceinline__ float ssim_from(const Moments &m, float *n_out, float *d_out) {
    // as the reference's tensor expressions (layers.py:670-677): every product and difference rounded on its own
#pragma clang fp contract(off)
    const float sx = m.ex2 - m.mux * m.mux, sy = m.ey2 - m.muy * m.muy, sxy = m.exy - m.mux * m.muy;
    const float n = (2.f * m.mux * m.muy + kC1) * (2.f * sxy + kC2);
    const float d = (m.mux * m.mux + m.muy * m.muy + kC1) * (sx + sy + kC2);
    if (n_out) { *n_out = n; *d_out = d; }
    return (1.f - n / d) / 2.f;
}

// d SSIM-term / d(mu_x, E[x^2], E[xy]) at one window, times the upstream factor gs: the three coefficient maps of the
// backward (see ssim.hip).  Zero outside the clamp's pass band.
__device__ __forceinline__ void ssim_coeffs(const Moments &m, float gs, float &A, float &Bc, float &Cc) {
    float n, d;
    const float raw = ssim_from(m, &n, &d);
    A = Bc = Cc = 0.f;
    if (raw >= 0.f && raw <= 1.f) {  // clamp passes gradient only inside [0,1]
        const float sx = m.ex2 - m.mux * m.mux, sy = m.ey2 - m.muy * m.muy, sxy = m.exy - m.mux * m.muy;
        const float A1 = 2.f * m.mux * m.muy + kC1, A2 = 2.f * sxy + kC2;
        const float B1 = m.mux * m.mux + m.muy * m.muy + kC1, B2 = sx + sy + kC2;
        const float dn_dmux = 2.f * m.muy * A2 - 2.f * m.muy * A1;
        const float dd_dmux = 2.f * m.mux * B2 - 2.f * m.mux * B1;
        const float rd = 1.f / d, rd2 = rd * rd;   // one division instead of three (a gradient: 1e-4 parity, not bit parity)
        A = gs * (-0.5f * (dn_dmux * d - n * dd_dmux) * rd2);
        Bc = gs * (0.5f * n * B1 * rd2);
        Cc = gs * (-0.5f * (2.f * A1) * rd);
    }
}

// ssim_coeffs on one value or on a pair (the backward's coefficient maps; a gradient: 1e-4 parity, not bit parity -- but the
// clamp's pass band is decided on the forward's bits).  The reciprocal of d serves the forward quotient and the derivatives.
template <class T>
__device__ __forceinline__ void ssim_coeffs_t(const MomentsT<T> &m, float gs, T &A, T &Bc, T &Cc, T &raw) {
    T n, d, sx, sy, sxy, rd;
    {
#pragma clang fp contract(off)
        sx = m.ex2 - m.mux * m.mux; sy = m.ey2 - m.muy * m.muy; sxy = m.exy - m.mux * m.muy;
        n = (2.f * m.mux * m.muy + kC1) * (2.f * sxy + kC2);
        d = (m.mux * m.mux + m.muy * m.muy + kC1) * (sx + sy + kC2);
        rd = md_rcp_newton(d);
#if MD_PHOTO_FAST_DIV
        raw = (1.f - md_div_core(n, d, rd)) / 2.f;
#else
        raw = (1.f - n / d) / 2.f;
#endif
    }
    const T A1 = 2.f * m.mux * m.muy + kC1, A2 = 2.f * sxy + kC2;
    const T B1 = m.mux * m.mux + m.muy * m.muy + kC1, B2 = sx + sy + kC2;
    const T dn_dmux = 2.f * m.muy * A2 - 2.f * m.muy * A1;
    const T dd_dmux = 2.f * m.mux * B2 - 2.f * m.mux * B1;
    const T rd2 = rd * rd;
    A = gs * (-0.5f * (dn_dmux * d - n * dd_dmux) * rd2);
    Bc = gs * (0.5f * n * B1 * rd2);
    Cc = gs * (-0.5f * (2.f * A1) * rd);
}
__device__ __forceinline__ bool md_in01(float raw) { return raw >= 0.f && raw <= 1.f; }  // clamp passes gradient only inside [0,1]

}  // namespace mdp
