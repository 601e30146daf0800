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

}  // namespace mdp
