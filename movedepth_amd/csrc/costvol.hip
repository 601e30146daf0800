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
//   LDS       = the source-feature window the tile's epipolar segments can reach, staged ONCE per
//               workgroup channels-last ([pixel][CPW] as 16-byte quads, XOR-swizzled so a wave's
//               ds_read_b128 taps are bank-conflict free), zero-filled outside the image: that is
//               grid_sample's 'zeros' padding for free.  Taps are 4 x CPW/4 ds_read_b128 per hypothesis
//               instead of 4 x CPW scattered global loads.  The window origin comes from the tile's
//               bounding box of tap positions at the first and last hypothesis of the slice; taps that
//               still fall outside (wild poses) take a slow, correct global-memory path.
//   backward  = same walk; d_ref accumulates in registers, d_src is scattered with ds_add_f32 into a
//               planar LDS window and flushed once per workgroup with global float atomics.
#include <limits.h>

#include "md_common.hpp"

namespace {

struct CostvolArgs {
    const float *ref, *src, *K, *invK, *pose, *hyp, *prior, *ztrans;
    float scale_fac;
    int sched_type;
    int B, C, G, h, w, D;
    int dsplit, dper;
    int tiles_x;
    // forward: out; backward: gout (same addressing)
    float *out;
    const float *gout;
    long long sb, sd, sg;
    float *d_ref, *d_src;
};

template <int TW>
struct Tile {
    static constexpr int TH = 256 / TW;
    static constexpr int WW = TW + 16;  // window: tile + 16 columns / 8 rows of epipolar reach
    static constexpr int WH = TH + 8;
};

// swizzled quad index inside a pixel's QPP quads (see header comment)
template <int QPP>
__device__ __forceinline__ int swz(int wx, int q) {
    if (QPP == 1) return 0;
    return q ^ ((wx / (16 / QPP)) & (QPP - 1));
}

// Hypothesis for (pixel, k): from the hyp tensor or the fused schedule.
__device__ __forceinline__ float load_hyp(const CostvolArgs &a, int b, int k, int p, float prior_c, float one_pf) {
    if (a.hyp) return a.hyp[((size_t)b * a.D + k) * (size_t)(a.h * a.w) + p];
    return md_hypothesis(prior_c, one_pf, k, a.D, a.sched_type);
}

// channel held in LDS slot k of this workgroup: slot k = j*N + i  ->  channel i*G + gbase + j
template <int N>
__device__ __forceinline__ int slot_channel(int k, int G, int gbase) {
    return (k % N) * G + gbase + k / N;
}

// Tile set-up shared by forward and backward: camera constants, the thread's pixel/ray, the window origin,
// and the staged source window.
template <int GS, int N, int TW>
struct Setup {
    static constexpr int CPW = GS * N;
    static constexpr int QPP = CPW / 4;
    using T = Tile<TW>;

    CamMats cam;
    int b, ds, gbase, x, y, p;
    bool valid;
    float r0, r1, r2, prior_c, one_pf;
    int d0, d1, ox, oy;

    __device__ __forceinline__ void init(const CostvolArgs &a, float4 *win, int *bb) {
        const int tid = threadIdx.x;
        b = blockIdx.z / a.dsplit;
        ds = blockIdx.z % a.dsplit;
        gbase = blockIdx.y * GS;
        const int tx0 = (blockIdx.x % a.tiles_x) * TW, ty0 = (blockIdx.x / a.tiles_x) * T::TH;
        x = tx0 + tid % TW;
        y = ty0 + tid / TW;
        valid = x < a.w && y < a.h;
        p = y * a.w + x;
        d0 = ds * a.dper;
        d1 = min(a.D, d0 + a.dper);
        cam = md_load_cam(a.K + b * 16, a.invK + b * 16, a.pose + b * 16);
        md_ray(cam, (float)x, (float)y, r0, r1, r2);
        prior_c = 1.f;
        one_pf = 1.f;
        if (!a.hyp) {
            one_pf = 1.f + (a.ztrans ? a.scale_fac * a.ztrans[b] : a.scale_fac);
            if (valid) prior_c = a.prior[(size_t)b * a.h * a.w + p];
        }
        // bounding box of the north-west taps at both ends of the hypothesis slice (u(d), v(d) are
        // monotone in d between them unless c_z changes sign; stragglers take the global path)
        int mnx = INT_MAX, mny = INT_MAX, mxx = INT_MIN, mxy = INT_MIN;
        if (valid && d1 > d0) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int k = e ? d1 - 1 : d0;
                Proj pr = md_project(cam, r0, r1, r2, load_hyp(a, b, k, p, prior_c, one_pf), a.w, a.h);
                Tap t = md_make_tap(pr.ix, pr.iy, a.w, a.h);
                if (t.x0 >= -1 && t.x0 < a.w && t.y0 >= -1 && t.y0 < a.h) {
                    mnx = min(mnx, t.x0); mxx = max(mxx, t.x0 + 1);
                    mny = min(mny, t.y0); mxy = max(mxy, t.y0 + 1);
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
        int spanx = mxx - mnx + 1, spany = mxy - mny + 1;
        ox = spanx <= T::WW ? mnx : mnx + (spanx - T::WW) / 2;
        oy = spany <= T::WH ? mny : mny + (spany - T::WH) / 2;
        // stage the window: consecutive threads -> consecutive columns (coalesced per channel plane)
        const size_t hw = (size_t)a.h * a.w;
        const float *srcb = a.src + (size_t)b * a.C * hw;
        for (int idx = tid; idx < T::WW * T::WH * QPP; idx += 256) {
            int wx = idx % T::WW, rest = idx / T::WW;
            int wy = rest % T::WH, q = rest / T::WH;
            int sx = ox + wx, sy = oy + wy;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sx >= 0 && sx < a.w && sy >= 0 && sy < a.h) {
                size_t o = (size_t)sy * a.w + sx;
                v.x = srcb[(size_t)slot_channel<N>(q * 4 + 0, a.G, gbase) * hw + o];
                v.y = srcb[(size_t)slot_channel<N>(q * 4 + 1, a.G, gbase) * hw + o];
                v.z = srcb[(size_t)slot_channel<N>(q * 4 + 2, a.G, gbase) * hw + o];
                v.w = srcb[(size_t)slot_channel<N>(q * 4 + 3, a.G, gbase) * hw + o];
            }
            win[(wy * T::WW + wx) * QPP + swz<QPP>(wx, q)] = v;
        }
        __syncthreads();
    }

    // Bilinear samples S[k] of the CPW staged channels at tap t ('zeros' padding).
    __device__ __forceinline__ void sample(const CostvolArgs &a, const float4 *win, const Tap &t, float *S,
                                           bool &in_win) const {
        const float wx0 = 1.f - t.wx1, wy0 = 1.f - t.wy1;
        const float w00 = wy0 * wx0, w01 = wy0 * t.wx1, w10 = t.wy1 * wx0, w11 = t.wy1 * t.wx1;
        const int lx = t.x0 - ox, ly = t.y0 - oy;
        const bool dead = t.x0 < -1 || t.x0 >= a.w || t.y0 < -1 || t.y0 >= a.h;
        in_win = lx >= 0 && lx + 1 < T::WW && ly >= 0 && ly + 1 < T::WH;
        if (dead) {
#pragma unroll
            for (int k = 0; k < CPW; ++k) S[k] = 0.f;
            in_win = false;
        } else if (in_win) {
            const int base0 = (ly * T::WW + lx) * QPP, base1 = base0 + T::WW * QPP;
#pragma unroll
            for (int q = 0; q < QPP; ++q) {
                float4 a00 = win[base0 + swz<QPP>(lx, q)], a01 = win[base0 + QPP + swz<QPP>(lx + 1, q)];
                float4 a10 = win[base1 + swz<QPP>(lx, q)], a11 = win[base1 + QPP + swz<QPP>(lx + 1, q)];
                S[q * 4 + 0] = a00.x * w00 + a01.x * w01 + a10.x * w10 + a11.x * w11;
                S[q * 4 + 1] = a00.y * w00 + a01.y * w01 + a10.y * w10 + a11.y * w11;
                S[q * 4 + 2] = a00.z * w00 + a01.z * w01 + a10.z * w10 + a11.z * w11;
                S[q * 4 + 3] = a00.w * w00 + a01.w * w01 + a10.w * w10 + a11.w * w11;
            }
        } else {
            // slow path: tap inside the image but outside the staged window
            const size_t hw = (size_t)a.h * a.w;
            const float *srcb = a.src + (size_t)b * a.C * hw;
            const int x1 = t.x0 + 1, y1 = t.y0 + 1;
            const bool vx0 = t.x0 >= 0, vx1 = x1 < a.w, vy0 = t.y0 >= 0, vy1 = y1 < a.h;
#pragma unroll
            for (int k = 0; k < CPW; ++k) {
                const float *pl = srcb + (size_t)slot_channel<N>(k, a.G, gbase) * hw;
                float s = 0.f;
                if (vx0 && vy0) s += pl[t.y0 * a.w + t.x0] * w00;
                if (vx1 && vy0) s += pl[t.y0 * a.w + x1] * w01;
                if (vx0 && vy1) s += pl[y1 * a.w + t.x0] * w10;
                if (vx1 && vy1) s += pl[y1 * a.w + x1] * w11;
                S[k] = s;
            }
        }
    }
};

template <int GS, int N, int TW>
__global__ __launch_bounds__(256) void costvol_fwd_kernel(CostvolArgs a) {
    using SU = Setup<GS, N, TW>;
    using T = Tile<TW>;
    constexpr int CPW = SU::CPW, QPP = SU::QPP;
    __shared__ float4 win[T::WW * T::WH * QPP];
    __shared__ int bb[16];
    SU s;
    s.init(a, win, bb);
    if (!s.valid) return;

    const size_t hw = (size_t)a.h * a.w;
    float rf[CPW];
#pragma unroll
    for (int k = 0; k < CPW; ++k) rf[k] = a.ref[((size_t)s.b * a.C + slot_channel<N>(k, a.G, s.gbase)) * hw + s.p];

    float *outp = a.out + (size_t)s.b * a.sb + (size_t)s.gbase * a.sg + s.p;
    float dnext = load_hyp(a, s.b, s.d0, s.p, s.prior_c, s.one_pf);
    for (int d = s.d0; d < s.d1; ++d) {
        const float dep = dnext;
        if (d + 1 < s.d1) dnext = load_hyp(a, s.b, d + 1, s.p, s.prior_c, s.one_pf);
        Proj pr = md_project(s.cam, s.r0, s.r1, s.r2, dep, a.w, a.h);
        Tap t = md_make_tap(pr.ix, pr.iy, a.w, a.h);
        float S[CPW];
        bool in_win;
        s.sample(a, win, t, S, in_win);
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < N; ++i) acc += S[j * N + i] * rf[j * N + i];
            outp[(size_t)d * a.sd + (size_t)j * a.sg] = acc / (float)N;
        }
    }
}

template <int GS, int N, int TW>
__global__ __launch_bounds__(256) void costvol_bwd_kernel(CostvolArgs a) {
    using SU = Setup<GS, N, TW>;
    using T = Tile<TW>;
    constexpr int CPW = SU::CPW, QPP = SU::QPP;
    constexpr int WP = T::WW * T::WH;
    __shared__ float4 win[WP * QPP];
    __shared__ float gw[CPW * WP];  // planar d_src window: lanes on neighbouring columns hit neighbouring banks
    __shared__ int bb[16];
    const int tid = threadIdx.x;
    for (int i = tid; i < CPW * WP; i += 256) gw[i] = 0.f;
    SU s;
    s.init(a, win, bb);  // contains the barriers that also publish the zeroed gw

    const size_t hw = (size_t)a.h * a.w;
    if (s.valid) {
        float rf[CPW], dref[CPW];
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            rf[k] = a.ref[((size_t)s.b * a.C + slot_channel<N>(k, a.G, s.gbase)) * hw + s.p];
            dref[k] = 0.f;
        }
        const float *gp = a.gout + (size_t)s.b * a.sb + (size_t)s.gbase * a.sg + s.p;
        float *dsrcb = a.d_src + (size_t)s.b * a.C * hw;
        float dnext = load_hyp(a, s.b, s.d0, s.p, s.prior_c, s.one_pf);
        for (int d = s.d0; d < s.d1; ++d) {
            const float dep = dnext;
            if (d + 1 < s.d1) dnext = load_hyp(a, s.b, d + 1, s.p, s.prior_c, s.one_pf);
            float gq[GS];
#pragma unroll
            for (int j = 0; j < GS; ++j) gq[j] = gp[(size_t)d * a.sd + (size_t)j * a.sg] / (float)N;
            Proj pr = md_project(s.cam, s.r0, s.r1, s.r2, dep, a.w, a.h);
            Tap t = md_make_tap(pr.ix, pr.iy, a.w, a.h);
            float S[CPW];
            bool in_win;
            s.sample(a, win, t, S, in_win);
#pragma unroll
            for (int k = 0; k < CPW; ++k) dref[k] += gq[k / N] * S[k];
            const float wx0 = 1.f - t.wx1, wy0 = 1.f - t.wy1;
            const float w00 = wy0 * wx0, w01 = wy0 * t.wx1, w10 = t.wy1 * wx0, w11 = t.wy1 * t.wx1;
            if (in_win) {
                const int o = (t.y0 - s.oy) * T::WW + (t.x0 - s.ox);
#pragma unroll
                for (int k = 0; k < CPW; ++k) {
                    const float v = gq[k / N] * rf[k];
                    float *g = gw + k * WP + o;
                    atomicAdd(g, v * w00);
                    atomicAdd(g + 1, v * w01);
                    atomicAdd(g + T::WW, v * w10);
                    atomicAdd(g + T::WW + 1, v * w11);
                }
            } else {
                const int x1 = t.x0 + 1, y1 = t.y0 + 1;
                const bool vx0 = t.x0 >= 0 && t.x0 < a.w, vx1 = x1 >= 0 && x1 < a.w;
                const bool vy0 = t.y0 >= 0 && t.y0 < a.h, vy1 = y1 >= 0 && y1 < a.h;
                if ((vx0 || vx1) && (vy0 || vy1)) {
#pragma unroll
                    for (int k = 0; k < CPW; ++k) {
                        const float v = gq[k / N] * rf[k];
                        float *pl = dsrcb + (size_t)slot_channel<N>(k, a.G, s.gbase) * hw;
                        if (vx0 && vy0) unsafeAtomicAdd(pl + t.y0 * a.w + t.x0, v * w00);
                        if (vx1 && vy0) unsafeAtomicAdd(pl + t.y0 * a.w + x1, v * w01);
                        if (vx0 && vy1) unsafeAtomicAdd(pl + y1 * a.w + t.x0, v * w10);
                        if (vx1 && vy1) unsafeAtomicAdd(pl + y1 * a.w + x1, v * w11);
                    }
                }
            }
        }
        float *drp = a.d_ref + (size_t)s.b * a.C * hw + s.p;
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            float *o = drp + (size_t)slot_channel<N>(k, a.G, s.gbase) * hw;
            if (a.dsplit == 1) *o = dref[k];
            else unsafeAtomicAdd(o, dref[k]);
        }
    }
    __syncthreads();
    // flush the d_src window (cells outside the image are grid_sample's zero padding: dropped)
    float *dsrcb = a.d_src + (size_t)s.b * a.C * hw;
    for (int idx = tid; idx < CPW * WP; idx += 256) {
        const float v = gw[idx];
        if (v == 0.f) continue;
        const int k = idx / WP, cell = idx % WP;
        const int sx = s.ox + cell % T::WW, sy = s.oy + cell / T::WW;
        if (sx >= 0 && sx < a.w && sy >= 0 && sy < a.h)
            unsafeAtomicAdd(dsrcb + (size_t)slot_channel<N>(k, a.G, s.gbase) * hw + (size_t)sy * a.w + sx, v);
    }
}

int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

// Picks (GS, N, TW) and launches.  Supported: N = C/G in {1,2,4,8}, CPW = GS*N in {4,8,16}.
template <bool BWD>
int launch(CostvolArgs a, hipStream_t stream) {
    const int N = a.C / a.G;
    int GS = 0;
    if (N == 1) GS = (a.G % 8 == 0) ? 8 : ((a.G % 4 == 0) ? 4 : 0);
    else if (N == 2) GS = (a.G % 4 == 0) ? 4 : ((a.G % 2 == 0) ? 2 : 0);
    else if (N == 4) GS = (a.G % 2 == 0) ? 2 : 1;
    else if (N == 8) GS = 1;
    if (GS == 0) {
        md_set_error("costvol: unsupported channel grouping C=%d G=%d (need C/G in {1,2,4,8} and a group split)", a.C, a.G);
        return MD_EINVAL;
    }
    const int TW = (a.w % 64 == 0 && env_int("MD_COSTVOL_TW", 0) != 32) || env_int("MD_COSTVOL_TW", 0) == 64 ? 64 : 32;
    const int TH = 256 / TW;
    a.tiles_x = md_cdiv(a.w, TW);
    const int tiles = a.tiles_x * md_cdiv(a.h, TH);
    const int splits = a.G / GS;
    // enough workgroups to fill 256 CUs several times over; each D slice re-stages its window
    int dsplit = env_int("MD_COSTVOL_DSPLIT", 0);
    if (dsplit <= 0) {
        long long wgs = (long long)tiles * splits * a.B;
        dsplit = (int)((2048 + wgs - 1) / wgs);
        int cap = a.D / 16 > 0 ? a.D / 16 : 1;
        if (dsplit > cap) dsplit = cap;
    }
    if (dsplit > a.D) dsplit = a.D;
    if (dsplit < 1) dsplit = 1;
    a.dsplit = dsplit;
    a.dper = md_cdiv(a.D, dsplit);
    a.dsplit = md_cdiv(a.D, a.dper);
    dim3 grid(tiles, splits, a.B * a.dsplit), block(256);

#define MD_CV_LAUNCH(GS_, N_, TW_)                                                         \
    do {                                                                                   \
        if (BWD) hipLaunchKernelGGL((costvol_bwd_kernel<GS_, N_, TW_>), grid, block, 0, stream, a); \
        else hipLaunchKernelGGL((costvol_fwd_kernel<GS_, N_, TW_>), grid, block, 0, stream, a);     \
    } while (0)
#define MD_CV_TW(GS_, N_)                     \
    do {                                      \
        if (TW == 64) MD_CV_LAUNCH(GS_, N_, 64); \
        else MD_CV_LAUNCH(GS_, N_, 32);       \
    } while (0)

    if (N == 1 && GS == 8) MD_CV_TW(8, 1);
    else if (N == 1 && GS == 4) MD_CV_TW(4, 1);
    else if (N == 2 && GS == 4) MD_CV_TW(4, 2);
    else if (N == 2 && GS == 2) MD_CV_TW(2, 2);
    else if (N == 4 && GS == 2) MD_CV_TW(2, 4);
    else if (N == 4 && GS == 1) MD_CV_TW(1, 4);
    else MD_CV_TW(1, 8);
#undef MD_CV_TW
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
    MD_REQUIRE((long long)B * 65535 >= 1 && B <= 16384, "%s: batch too large", fn);
    return MD_OK;
}

}  // namespace

extern "C" int md_costvol_fwd(const float *ref, const float *src, const float *K, const float *invK,
                              const float *pose, const float *hyp, const float *prior, const float *ztrans,
                              float scale_fac, int sched_type, int B, int C, int G, int h, int w, int D, float *out,
                              long long out_sb, long long out_sd, long long out_sg, md_stream_t stream) {
    int rc = check_common("md_costvol_fwd", ref, src, K, invK, pose, hyp, prior, sched_type, B, C, G, h, w, D);
    if (rc) return rc;
    MD_REQUIRE(out, "md_costvol_fwd: null output");
    CostvolArgs a{};
    a.ref = ref; a.src = src; a.K = K; a.invK = invK; a.pose = pose; a.hyp = hyp; a.prior = prior; a.ztrans = ztrans;
    a.scale_fac = scale_fac; a.sched_type = sched_type;
    a.B = B; a.C = C; a.G = G; a.h = h; a.w = w; a.D = D;
    a.out = out; a.sb = out_sb; a.sd = out_sd; a.sg = out_sg;
    return launch<false>(a, (hipStream_t)stream);
}

extern "C" int md_costvol_bwd(const float *gout, long long g_sb, long long g_sd, long long g_sg, const float *ref,
                              const float *src, const float *K, const float *invK, const float *pose,
                              const float *hyp, const float *prior, const float *ztrans, float scale_fac,
                              int sched_type, int B, int C, int G, int h, int w, int D, float *d_ref, float *d_src,
                              md_stream_t stream) {
    int rc = check_common("md_costvol_bwd", ref, src, K, invK, pose, hyp, prior, sched_type, B, C, G, h, w, D);
    if (rc) return rc;
    MD_REQUIRE(gout && d_ref && d_src, "md_costvol_bwd: null gradient tensor");
    CostvolArgs a{};
    a.ref = ref; a.src = src; a.K = K; a.invK = invK; a.pose = pose; a.hyp = hyp; a.prior = prior; a.ztrans = ztrans;
    a.scale_fac = scale_fac; a.sched_type = sched_type;
    a.B = B; a.C = C; a.G = G; a.h = h; a.w = w; a.D = D;
    a.gout = gout; a.sb = g_sb; a.sd = g_sd; a.sg = g_sg;
    a.d_ref = d_ref; a.d_src = d_src;
    const size_t bytes = sizeof(float) * (size_t)B * C * h * w;
    MD_CHECK_HIP(hipMemsetAsync(d_src, 0, bytes, (hipStream_t)stream));
    MD_CHECK_HIP(hipMemsetAsync(d_ref, 0, bytes, (hipStream_t)stream));
    return launch<true>(a, (hipStream_t)stream);
}
