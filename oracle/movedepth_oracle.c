/*
 * movedepth_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C (fp32) CPU restatement of the MOVEDepth training hot path: the algorithm
 * the reference expresses as compositions of PyTorch ops.  It is the checker the
 * HIP kernels are compared against in tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py.  Nothing in movedepth_amd/ may import, link or
 * call it; the product path fails loudly when the HIP library is missing.
 *
 * Parity pin: every function below is checked against golden vectors produced by
 * the reference's own PyTorch-CPU code (tools/gen_golden.py -> tests/golden/ npz files,
 * tests/test_oracle_golden.py).  The reference ships no tests of its own
 * (SURVEY.md section 4), so those fixtures are the only pin.
 *
 * Citations "layers.py:N" / "trainer.py:N" are into /root/reference/movedepth/.
 * Third-party arithmetic restated from PyTorch's published semantics (reference
 * pins torch==1.7.1, environment.yml:14; fixtures made with torch 2.10 CPU):
 *   F.grid_sample(bilinear, align_corners=True, zeros|border), AvgPool2d(3,1),
 *   ReflectionPad2d(1), F.interpolate(bilinear, align_corners=False), softmax.
 *
 * All arrays are contiguous row-major float32.  Optional OpenMP (-fopenmp).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int mdo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void mdo_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------ geometry */

/* P = (K @ T)[:3, :]   -- layers.py:608 */
static void kt_rows(const float *K, const float *T, float P[12]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s += K[i * 4 + k] * T[k * 4 + j];
            P[i * 4 + j] = s;
        }
}

/* Operation order of the three matrix products on the path.  The reference writes them as torch.matmul (layers.py:582, 608,
 * 610); what arithmetic that is, is decided by the BLAS behind it, and the committed fixtures -- the reference's own outputs --
 * are the pin: tools/diag/op_order_search.py evaluates every combination of {each product and sum rounded, fused multiply-add
 * chain ascending, descending} against the 20,720 pixel coordinates held by tests/golden/{warp_small, warp_border, geometry,
 * losses_mono}.npz; exactly ONE combination reproduces them, all of them, bit for bit:
 *   P = K @ T (4x4 by 4x4: the small-matrix path)        -- every product and sum rounded on its own (kt_rows above);
 *   inv_K[:3,:3] @ (x, y, 1) and P @ (X, Y, Z, 1) (sgemm)  -- acc = a0 b0, then acc = fma(a_k, b_k, acc), k ascending.
 * This file is compiled with -ffp-contract=off, so the only fused operations are the fmaf calls written here. */
static inline float dot3_fma(const float *a, float x, float y) { return fmaf(a[1], y, a[0] * x) + a[2]; }
static inline float dot4_fma(const float *a, float X, float Y, float Z) { return fmaf(a[2], Z, fmaf(a[1], Y, a[0] * X)) + a[3]; }

/* One pixel through BackprojectDepth (layers.py:581-586) and Project3D (layers.py:608-620).
 * Returns the normalised grid coordinates (gx, gy); optionally the camera point (X,Y,Z). */
static inline void project_pixel(const float *invK, const float *P, float x, float y, float d, float eps,
                                 int w, int h, float *gx, float *gy, float *X3) {
    /* inv_K[:3,:3] @ (x, y, 1) */
    float r0 = dot3_fma(invK, x, y);
    float r1 = dot3_fma(invK + 4, x, y);
    float r2 = dot3_fma(invK + 8, x, y);
    float X = d * r0, Y = d * r1, Z = d * r2; /* depth * cam_points, layers.py:583 */
    if (X3) { X3[0] = X; X3[1] = Y; X3[2] = Z; }
    float c0 = dot4_fma(P, X, Y, Z);
    float c1 = dot4_fma(P + 4, X, Y, Z);
    float c2 = dot4_fma(P + 8, X, Y, Z);
    float zz = c2 + eps;                       /* layers.py:612 */
    float u = c0 / zz, v = c1 / zz;
    u = u / (float)(w - 1);                    /* layers.py:618 */
    v = v / (float)(h - 1);                    /* layers.py:619 */
    *gx = (u - 0.5f) * 2.f;                    /* layers.py:620 */
    *gy = (v - 0.5f) * 2.f;
}

/* grid_sample's un-normalise for align_corners=True */
static inline float unnorm(float g, int size) { return ((g + 1.f) / 2.f) * (float)(size - 1); }

/* BackprojectDepth + Project3D over a batch.  depth [Bs,h*w]; invK/K/T [nk,16] with nk==1 (shared, the
 * cost-volume use: Bs = D hypotheses of one sample) or nk==Bs (photometric use).
 * cam_points [Bs,4,h*w] (may be NULL), pix [Bs,h,w,2]. */
void mdo_backproject_project(const float *depth, const float *invK, const float *K, const float *T, int Bs, int nk,
                             int h, int w, float eps, float *cam_points, float *pix) {
    int hw = h * w;
    for (int b = 0; b < Bs; ++b) {
        int kb = nk == 1 ? 0 : b;
        float P[12];
        kt_rows(K + kb * 16, T + kb * 16, P);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                int p = y * w + x;
                float gx, gy, X3[3];
                project_pixel(invK + kb * 16, P, (float)x, (float)y, depth[b * hw + p], eps, w, h, &gx, &gy, X3);
                if (cam_points) {
                    float *cp = cam_points + (size_t)b * 4 * hw;
                    cp[p] = X3[0]; cp[hw + p] = X3[1]; cp[2 * hw + p] = X3[2]; cp[3 * hw + p] = 1.f;
                }
                pix[((size_t)b * hw + p) * 2 + 0] = gx;
                pix[((size_t)b * hw + p) * 2 + 1] = gy;
            }
    }
}

/* disp_to_depth, layers.py:400-409 */
void mdo_disp_to_depth(const float *disp, int n, float min_depth, float max_depth, float *scaled, float *depth) {
    float min_disp = 1.f / max_depth, max_disp = 1.f / min_depth;
    for (int i = 0; i < n; ++i) {
        float s = min_disp + (max_disp - min_disp) * disp[i];
        if (scaled) scaled[i] = s;
        depth[i] = 1.f / s;
    }
}

/* transformation_from_parameters, layers.py:412-429 (+ rot_from_axisangle 479-518, get_translation_matrix 464-477).
 * axisangle, translation [B,3]; out [B,16]. */
void mdo_transformation_from_parameters(const float *aa, const float *tr, int B, int invert, float *out) {
    for (int b = 0; b < B; ++b) {
        const float *v = aa + b * 3;
        float angle = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        float x = v[0] / (angle + 1e-7f), y = v[1] / (angle + 1e-7f), z = v[2] / (angle + 1e-7f);
        float ca = cosf(angle), sa = sinf(angle), C = 1.f - ca;
        float xs = x * sa, ys = y * sa, zs = z * sa, xC = x * C, yC = y * C, zC = z * C;
        float xyC = x * yC, yzC = y * zC, zxC = z * xC;
        float R[16] = {x * xC + ca, xyC - zs, zxC + ys, 0, xyC + zs, y * yC + ca, yzC - xs, 0,
                       zxC - ys, yzC + xs, z * zC + ca, 0, 0, 0, 0, 1};
        float t[3] = {tr[b * 3], tr[b * 3 + 1], tr[b * 3 + 2]};
        float Rm[16], Tm[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        memcpy(Rm, R, sizeof(R));
        if (invert) { /* R^T, -t */
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) Rm[i * 4 + j] = R[j * 4 + i];
            t[0] = -t[0]; t[1] = -t[1]; t[2] = -t[2];
        }
        Tm[3] = t[0]; Tm[7] = t[1]; Tm[11] = t[2];
        const float *A = invert ? Rm : Tm, *Bm = invert ? Tm : Rm; /* M = R@T if invert else T@R */
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float s = 0.f;
                for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * Bm[k * 4 + j];
                out[b * 16 + i * 4 + j] = s;
            }
    }
}

/* ------------------------------------------------------------------ depth-range schedule */

/* schedule_depth_rangev2 (layers.py:256-284) when ztrans==NULL, schedule_depth_range_zv2 (layers.py:370-398)
 * otherwise (ztrans[b] = z_scale * T[b,2,3], trainer.py:340).  type: 0 inverse, 1 linear, 2 log.
 * prior [B,hw] -> out [B,D,hw].  Index 0 is the FARTHEST hypothesis for 'inverse'. */
void mdo_schedule(const float *prior, const float *ztrans, int B, int hw, int D, float scale_fac, int type,
                  float *out) {
    for (int b = 0; b < B; ++b) {
        float f = ztrans ? scale_fac * ztrans[b] : scale_fac;
        float one_pf = 1.f + f;
        for (int k = 0; k < D; ++k) {
            float itv;
            if (type == 2) { /* layers.py:274-278: exp(log(0.1) + log(1/0.1) * K/(D-1)) in fp32 */
                itv = expf(logf(0.1f) + logf(1.f / 0.1f) * (float)k / (float)(D - 1));
            } else {
                itv = (float)k / (float)(D - 1);
            }
            for (int p = 0; p < hw; ++p) {
                float c = prior[b * hw + p];
                float dmin = c / one_pf, dmax = c * one_pf;
                float v;
                if (type == 0) {
                    float inv = 1.f / dmax + (1.f / dmin - 1.f / dmax) * itv;
                    v = 1.f / inv;
                } else {
                    v = dmin + (dmax - dmin) * itv;
                }
                out[((size_t)b * D + k) * hw + p] = v;
            }
        }
    }
}

/* ------------------------------------------------------------------ bilinear taps */

typedef struct {
    int x0, y0;          /* north-west tap */
    float wx1, wy1;      /* distance to the west / north tap (weight of the east / south tap) */
    float gmx, gmy;      /* d(ix)/d(grid) multipliers incl. border clipping (0 where clipped) */
} tap_t;

/* padding: 0 zeros, 1 border (clip to [0,size-1], zero grid-gradient where clipped) */
static inline tap_t make_tap(float gx, float gy, int w, int h, int border) {
    tap_t t;
    float ix = unnorm(gx, w), iy = unnorm(gy, h);
    t.gmx = (float)(w - 1) / 2.f;
    t.gmy = (float)(h - 1) / 2.f;
    if (border) {
        /* clip_coordinates_set_grad: the borders themselves count as clipped (zero grid gradient) */
        if (!(ix > 0.f)) { ix = 0.f; t.gmx = 0.f; }
        else if (ix >= (float)(w - 1)) { ix = (float)(w - 1); t.gmx = 0.f; }
        if (!(iy > 0.f)) { iy = 0.f; t.gmy = 0.f; }
        else if (iy >= (float)(h - 1)) { iy = (float)(h - 1); t.gmy = 0.f; }
    }
    float fx = floorf(ix), fy = floorf(iy);
    t.wx1 = ix - fx;
    t.wy1 = iy - fy;
    /* guard the int conversion for wild coordinates; such taps are out of range anyway */
    if (!(fx > -2.f)) fx = -2.f; if (fx > (float)w) fx = (float)w;
    if (!(fy > -2.f)) fy = -2.f; if (fy > (float)h) fy = (float)h;
    if (ix != ix || iy != iy) { fx = -2.f; fy = -2.f; t.wx1 = 0.f; t.wy1 = 0.f; }
    t.x0 = (int)fx;
    t.y0 = (int)fy;
    return t;
}

static inline float tap_sample(const float *img, int w, int h, const tap_t *t) {
    float wx0 = 1.f - t->wx1, wy0 = 1.f - t->wy1;
    int x0 = t->x0, y0 = t->y0, x1 = x0 + 1, y1 = y0 + 1;
    int vx0 = x0 >= 0 && x0 < w, vx1 = x1 >= 0 && x1 < w, vy0 = y0 >= 0 && y0 < h, vy1 = y1 >= 0 && y1 < h;
    /* grid_sample's interpolation as the reference's build evaluates it -- nw_val * nw, then fused multiply-adds of the ne,
     * sw, se taps in that order, out-of-range taps valued 0 -- found the same way as the matrix products' order above
     * (tools/diag/op_order_search.py: the only one of five candidate orders that reproduces the 30,720 warped values of
     * tests/golden/{warp_small, warp_border, losses_mono}.npz bit for bit) */
    float nw = (vx0 && vy0) ? img[y0 * w + x0] : 0.f, ne = (vx1 && vy0) ? img[y0 * w + x1] : 0.f;
    float sw = (vx0 && vy1) ? img[y1 * w + x0] : 0.f, se = (vx1 && vy1) ? img[y1 * w + x1] : 0.f;
    float o = nw * (wy0 * wx0);
    o = fmaf(ne, wy0 * t->wx1, o);
    o = fmaf(sw, t->wy1 * wx0, o);
    o = fmaf(se, t->wy1 * t->wx1, o);
    return o;
}

static inline void tap_scatter(float *img, int w, int h, const tap_t *t, float g) {
    float wx0 = 1.f - t->wx1, wy0 = 1.f - t->wy1;
    int x0 = t->x0, y0 = t->y0, x1 = x0 + 1, y1 = y0 + 1;
    int vx0 = x0 >= 0 && x0 < w, vx1 = x1 >= 0 && x1 < w, vy0 = y0 >= 0 && y0 < h, vy1 = y1 >= 0 && y1 < h;
    if (vx0 && vy0) img[y0 * w + x0] += g * (wy0 * wx0);
    if (vx1 && vy0) img[y0 * w + x1] += g * (wy0 * t->wx1);
    if (vx0 && vy1) img[y1 * w + x0] += g * (t->wy1 * wx0);
    if (vx1 && vy1) img[y1 * w + x1] += g * (t->wy1 * t->wx1);
}

static inline void tap_scatter_atomic(float *img, int w, int h, const tap_t *t, float g) {
    float wx0 = 1.f - t->wx1, wy0 = 1.f - t->wy1;
    int x0 = t->x0, y0 = t->y0, x1 = x0 + 1, y1 = y0 + 1;
    int vx0 = x0 >= 0 && x0 < w, vx1 = x1 >= 0 && x1 < w, vy0 = y0 >= 0 && y0 < h, vy1 = y1 >= 0 && y1 < h;
    if (vx0 && vy0) {
#pragma omp atomic
        img[y0 * w + x0] += g * (wy0 * wx0);
    }
    if (vx1 && vy0) {
#pragma omp atomic
        img[y0 * w + x1] += g * (wy0 * t->wx1);
    }
    if (vx0 && vy1) {
#pragma omp atomic
        img[y1 * w + x0] += g * (t->wy1 * wx0);
    }
    if (vx1 && vy1) {
#pragma omp atomic
        img[y1 * w + x1] += g * (t->wy1 * t->wx1);
    }
}

/* ------------------------------------------------------------------ plane-sweep cost volume */

/* generate_costvol, layers.py:778-794.  ref, src [B,C,h,w]; K, invK [B,16]; hyp [B,D,h,w]; pose [B,16]
 * (= pose[:,0] of the (B,1,4,4) argument).  out_full [B,D,C,h,w]. */
void mdo_costvol_fwd(const float *ref, const float *src, const float *K, const float *invK, const float *hyp,
                     const float *pose, int B, int C, int h, int w, int D, float *out_full) {
    int hw = h * w;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d) {
            float P[12];
            kt_rows(K + b * 16, pose + b * 16, P);
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    int p = y * w + x;
                    float gx, gy;
                    project_pixel(invK + b * 16, P, (float)x, (float)y, hyp[((size_t)b * D + d) * hw + p], 1e-7f, w,
                                  h, &gx, &gy, NULL);
                    tap_t t = make_tap(gx, gy, w, h, 0); /* padding_mode='zeros', layers.py:791 */
                    for (int c = 0; c < C; ++c) {
                        float s = tap_sample(src + ((size_t)b * C + c) * hw, w, h, &t);
                        out_full[(((size_t)b * D + d) * C + c) * hw + p] = s * ref[((size_t)b * C + c) * hw + p];
                    }
                }
        }
}

/* generate_costvol + reshape(B,D,C/G,G,h,w).mean(2) (trainer.py:359): group g = channels {g, g+G, ...}.
 * out [B,D,G,h,w]. */
void mdo_costvol_grouped_fwd(const float *ref, const float *src, const float *K, const float *invK,
                             const float *hyp, const float *pose, int B, int C, int G, int h, int w, int D,
                             float *out) {
    int hw = h * w, n = C / G;
    /* every output element is independent: threads over (sample, plane, row), so that ONE sample keeps all cores busy */
#pragma omp parallel for collapse(3) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d)
            for (int y = 0; y < h; ++y) {
                float P[12];
                kt_rows(K + b * 16, pose + b * 16, P);
                for (int x = 0; x < w; ++x) {
                    int p = y * w + x;
                    float gx, gy;
                    project_pixel(invK + b * 16, P, (float)x, (float)y, hyp[((size_t)b * D + d) * hw + p], 1e-7f, w,
                                  h, &gx, &gy, NULL);
                    tap_t t = make_tap(gx, gy, w, h, 0);
                    for (int g = 0; g < G; ++g) {
                        float acc = 0.f;
                        for (int i = 0; i < n; ++i) {
                            int c = i * G + g;
                            acc += tap_sample(src + ((size_t)b * C + c) * hw, w, h, &t) *
                                   ref[((size_t)b * C + c) * hw + p];
                        }
                        out[(((size_t)b * D + d) * G + g) * hw + p] = acc / (float)n;
                    }
                }
        }
}

/* Autograd of the above w.r.t. ref and src (the grid is under no_grad, layers.py:784).
 * gout [B,D,G,h,w]; d_ref, d_src [B,C,h,w] (overwritten). */
void mdo_costvol_grouped_bwd(const float *gout, const float *ref, const float *src, const float *K,
                             const float *invK, const float *hyp, const float *pose, int B, int C, int G, int h,
                             int w, int D, float *d_ref, float *d_src) {
    int hw = h * w, n = C / G;
    memset(d_ref, 0, sizeof(float) * (size_t)B * C * hw);
    memset(d_src, 0, sizeof(float) * (size_t)B * C * hw);
    /* Threads over (row, column segment) of one sample at a time: a pixel -- hence its d_ref entries -- belongs to one thread;
     * the d_src scatter goes into a thread-private copy of the sample's gradient (no atomics), and the copies are added up in
     * thread order afterwards (deterministic for a given thread count).  XS column segments per row so that 48 rows feed
     * hundreds of threads. */
    int T = mdo_num_threads();
    const int XS = 8;
    float *priv = T > 1 ? (float *)calloc((size_t)T * C * hw, sizeof(float)) : NULL;
    for (int b = 0; b < B; ++b) {
        float P[12];
        kt_rows(K + b * 16, pose + b * 16, P);
#pragma omp parallel for collapse(2) schedule(static)
        for (int y = 0; y < h; ++y)
            for (int xs = 0; xs < XS; ++xs) {
#ifdef _OPENMP
                float *ds = priv ? priv + (size_t)omp_get_thread_num() * C * hw : d_src + (size_t)b * C * hw;
#else
                float *ds = d_src + (size_t)b * C * hw;
#endif
                int xa = (int)((long long)w * xs / XS), xb = (int)((long long)w * (xs + 1) / XS);
                for (int d = 0; d < D; ++d)
                    for (int x = xa; x < xb; ++x) {
                        int p = y * w + x;
                        float gx, gy;
                        project_pixel(invK + b * 16, P, (float)x, (float)y, hyp[((size_t)b * D + d) * hw + p], 1e-7f, w,
                                      h, &gx, &gy, NULL);
                        tap_t t = make_tap(gx, gy, w, h, 0);
                        for (int c = 0; c < C; ++c) {
                            float g = gout[(((size_t)b * D + d) * G + (c % G)) * hw + p] / (float)n;
                            size_t o = ((size_t)b * C + c) * hw;
                            d_ref[o + p] += g * tap_sample(src + o, w, h, &t); /* pixel p is owned by this thread */
                            tap_scatter(ds + (size_t)c * hw, w, h, &t, g * ref[o + p]);
                        }
                    }
            }
        if (priv) {
#pragma omp parallel for schedule(static)
            for (int i = 0; i < C * hw; ++i) {
                float acc = 0.f;
                for (int t = 0; t < T; ++t) { acc += priv[(size_t)t * C * hw + i]; priv[(size_t)t * C * hw + i] = 0.f; }
                d_src[(size_t)b * C * hw + i] = acc;
            }
        }
    }
    free(priv);
}

/* Confidence-weighted frame fusion, trainer.py:349-363.
 * vols: N pointers to [B,D,G,hw]; out [B,D,G,hw]; weights [N,B,hw] (may be NULL). */
void mdo_fuse_fwd(const float *const *vols, int N, int B, int D, int G, int hw, float *out, float *weights) {
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        float *wf = (float *)malloc(sizeof(float) * N);
        float *m = (float *)malloc(sizeof(float) * G);
        for (int p = 0; p < hw; ++p) {
            float wsum = 1e-8f;
            for (int f = 0; f < N; ++f) {
                const float *v = vols[f] + (size_t)b * D * G * hw + p;
                float mx = -INFINITY;
                for (int g = 0; g < G; ++g) {
                    float s = 0.f;
                    for (int d = 0; d < D; ++d) s += v[((size_t)d * G + g) * hw];
                    m[g] = s / (float)D; /* .mean(1) over D */
                    if (m[g] > mx) mx = m[g];
                }
                float den = 0.f;
                for (int g = 0; g < G; ++g) den += expf(m[g] - mx);
                wf[f] = 1.f / den; /* softmax(dim=G).max(): exp(0)/sum */
                wsum += wf[f];
                if (weights) weights[((size_t)f * B + b) * hw + p] = wf[f];
            }
            for (int d = 0; d < D; ++d)
                for (int g = 0; g < G; ++g) {
                    float acc = 0.f;
                    for (int f = 0; f < N; ++f) acc += wf[f] * vols[f][(((size_t)b * D + d) * G + g) * hw + p];
                    out[(((size_t)b * D + d) * G + g) * hw + p] = acc / wsum;
                }
        }
        free(wf); free(m);
    }
}

/* The evaluation script's fusion, evaluate_depth.py:225-243: cor_weight = softmax(cost_vols.mean(2), dim=1).max(1)[0] on a
 * (B,D,G,h,w) volume, i.e. mean over G, soft-max over D (the training code, above, takes the mean over D and the soft-max
 * over G: SURVEY App. B-7).  vols: N pointers [B,D,G,hw]; out [B,D,G,hw]; weights [N,B,hw] (may be NULL). */
void mdo_fuse_eval_fwd(const float *const *vols, int N, int B, int D, int G, int hw, float *out, float *weights) {
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        float *wf = (float *)malloc(sizeof(float) * N);
        float *m = (float *)malloc(sizeof(float) * D);
        for (int p = 0; p < hw; ++p) {
            float wsum = 1e-8f;
            for (int f = 0; f < N; ++f) {
                const float *v = vols[f] + (size_t)b * D * G * hw + p;
                float mx = -INFINITY;
                for (int d = 0; d < D; ++d) {
                    float s = 0.f;
                    for (int g = 0; g < G; ++g) s += v[((size_t)d * G + g) * hw];
                    m[d] = s / (float)G; /* .mean(2) over G */
                    if (m[d] > mx) mx = m[d];
                }
                float den = 0.f;
                for (int d = 0; d < D; ++d) den += expf(m[d] - mx);
                wf[f] = 1.f / den; /* softmax(dim=D).max(1) */
                wsum += wf[f];
                if (weights) weights[((size_t)f * B + b) * hw + p] = wf[f];
            }
            for (int d = 0; d < D; ++d)
                for (int g = 0; g < G; ++g) {
                    float acc = 0.f;
                    for (int f = 0; f < N; ++f) acc += wf[f] * vols[f][(((size_t)b * D + d) * G + g) * hw + p];
                    out[(((size_t)b * D + d) * G + g) * hw + p] = acc / wsum;
                }
        }
        free(wf); free(m);
    }
}

/* Autograd of mdo_fuse_fwd (the weights are NOT detached in the reference). d_vols: N pointers [B,D,G,hw]. */
void mdo_fuse_bwd(const float *gout, const float *const *vols, int N, int B, int D, int G, int hw,
                  float *const *d_vols) {
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        float *wf = (float *)malloc(sizeof(float) * N);
        float *pr = (float *)malloc(sizeof(float) * N * G);
        int *am = (int *)malloc(sizeof(int) * N);
        for (int p = 0; p < hw; ++p) {
            float wsum = 1e-8f;
            for (int f = 0; f < N; ++f) {
                const float *v = vols[f] + (size_t)b * D * G * hw + p;
                float mx = -INFINITY;
                float *m = pr + f * G;
                am[f] = 0;
                for (int g = 0; g < G; ++g) {
                    float s = 0.f;
                    for (int d = 0; d < D; ++d) s += v[((size_t)d * G + g) * hw];
                    m[g] = s / (float)D;
                    if (m[g] > mx) { mx = m[g]; am[f] = g; }
                }
                float den = 0.f;
                for (int g = 0; g < G; ++g) { m[g] = expf(m[g] - mx); den += m[g]; }
                for (int g = 0; g < G; ++g) m[g] /= den; /* softmax probabilities */
                wf[f] = m[am[f]];
                wsum += wf[f];
            }
            for (int f = 0; f < N; ++f) {
                /* dL/dw_f = sum_{d,g} gout * (vol_f - cor) / wsum */
                float dw = 0.f;
                for (int d = 0; d < D; ++d)
                    for (int g = 0; g < G; ++g) {
                        size_t o = (((size_t)b * D + d) * G + g) * hw + p;
                        float acc = 0.f;
                        for (int f2 = 0; f2 < N; ++f2) acc += wf[f2] * vols[f2][o];
                        dw += gout[o] * (vols[f][o] - acc / wsum) / wsum;
                    }
                const float *prf = pr + f * G;
                for (int d = 0; d < D; ++d)
                    for (int g = 0; g < G; ++g) {
                        size_t o = (((size_t)b * D + d) * G + g) * hw + p;
                        float dm = prf[am[f]] * ((g == am[f] ? 1.f : 0.f) - prf[g]); /* d max-softmax / d m_g */
                        d_vols[f][o] = gout[o] * wf[f] / wsum + dw * dm / (float)D;
                    }
            }
        }
        free(wf); free(pr); free(am);
    }
}

/* ------------------------------------------------------------------ photometric warp (border) */

/* generate_images_pred's warp: backproject (scale 0) -> project -> grid_sample(border, align_corners=True),
 * trainer.py:501-507, 519-529, 575-580.  img [B,Ci,H,W]; depth [B,H,W]; K,invK,T [B,16].
 * pix [B,H,W,2] (may be NULL); out [B,Ci,H,W]. */
void mdo_warp_fwd(const float *img, const float *depth, const float *K, const float *invK, const float *T, int B,
                  int Ci, int H, int W, float *pix, float *out) {
    int HW = H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y) {
            float P[12];
            kt_rows(K + b * 16, T + b * 16, P);
            for (int x = 0; x < W; ++x) {
                int p = y * W + x;
                float gx, gy;
                project_pixel(invK + b * 16, P, (float)x, (float)y, depth[(size_t)b * HW + p], 1e-7f, W, H, &gx, &gy,
                              NULL);
                if (pix) { pix[((size_t)b * HW + p) * 2] = gx; pix[((size_t)b * HW + p) * 2 + 1] = gy; }
                tap_t t = make_tap(gx, gy, W, H, 1);
                for (int c = 0; c < Ci; ++c)
                    out[((size_t)b * Ci + c) * HW + p] = tap_sample(img + ((size_t)b * Ci + c) * HW, W, H, &t);
            }
        }
}

/* Autograd of mdo_warp_fwd w.r.t. depth and T (the image is an input, no grad).  SURVEY App. A.2.
 * gout [B,Ci,H,W]; d_depth [B,H,W]; d_T [B,16]. */
void mdo_warp_bwd(const float *gout, const float *img, const float *depth, const float *K, const float *invK,
                  const float *T, int B, int Ci, int H, int W, float *d_depth, float *d_T) {
    int HW = H * W;
    for (int b = 0; b < B; ++b) {
        float P[12];
        double dP[12];
        const float *iK = invK + b * 16;
        kt_rows(K + b * 16, T + b * 16, P);
        for (int i = 0; i < 12; ++i) dP[i] = 0.0;
        /* rows over threads; the 12 sums of dL/dP are double-precision reductions (their order depends on the thread count:
         * ~1e-16 relative) */
#pragma omp parallel for schedule(static) reduction(+ : dP[:12])
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int p = y * W + x;
                float d = depth[(size_t)b * HW + p];
                float r0 = dot3_fma(iK, (float)x, (float)y), r1 = dot3_fma(iK + 4, (float)x, (float)y),
                      r2 = dot3_fma(iK + 8, (float)x, (float)y);          /* as project_pixel */
                float X = d * r0, Y = d * r1, Z = d * r2;
                float c0 = dot4_fma(P, X, Y, Z), c1 = dot4_fma(P + 4, X, Y, Z), c2 = dot4_fma(P + 8, X, Y, Z);
                float zz = c2 + 1e-7f;
                float u = c0 / zz, v = c1 / zz;
                float gx = (u / (float)(W - 1) - 0.5f) * 2.f, gy = (v / (float)(H - 1) - 0.5f) * 2.f;
                tap_t t = make_tap(gx, gy, W, H, 1);
                /* d(out)/d(ix), d(out)/d(iy): grid_sampler_2d_backward, bilinear */
                int x0 = t.x0, y0 = t.y0, x1 = x0 + 1, y1 = y0 + 1;
                int vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
                float wx0 = 1.f - t.wx1, wy0 = 1.f - t.wy1;
                float gix = 0.f, giy = 0.f;
                for (int c = 0; c < Ci; ++c) {
                    const float *im = img + ((size_t)b * Ci + c) * HW;
                    float g = gout[((size_t)b * Ci + c) * HW + p];
                    float nw = (vx0 && vy0) ? im[y0 * W + x0] : 0.f, ne = (vx1 && vy0) ? im[y0 * W + x1] : 0.f,
                          sw = (vx0 && vy1) ? im[y1 * W + x0] : 0.f, se = (vx1 && vy1) ? im[y1 * W + x1] : 0.f;
                    gix += g * ((ne - nw) * wy0 + (se - sw) * t.wy1);
                    giy += g * ((sw - nw) * wx0 + (se - ne) * t.wx1);
                }
                /* grid -> pix: ix = ((gx+1)/2)(W-1), gx = (u/(W-1) - .5)*2 */
                float du = gix * t.gmx * (2.f / (float)(W - 1));
                float dv = giy * t.gmy * (2.f / (float)(H - 1));
                float dc0 = du / zz, dc1 = dv / zz, dc2 = -(du * u + dv * v) / zz;
                float a0 = P[0] * r0 + P[1] * r1 + P[2] * r2, a1 = P[4] * r0 + P[5] * r1 + P[6] * r2,
                      a2 = P[8] * r0 + P[9] * r1 + P[10] * r2;
                d_depth[(size_t)b * HW + p] = dc0 * a0 + dc1 * a1 + dc2 * a2;
                float Xh[4] = {X, Y, Z, 1.f}, dc[3] = {dc0, dc1, dc2};
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 4; ++j) dP[i * 4 + j] += (double)(dc[i] * Xh[j]);
            }
        /* P = (K@T)[:3]  =>  dT[k][j] = sum_{i<3} K[i][k] dP[i][j] */
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 4; ++j) {
                double s = 0.0;
                for (int i = 0; i < 3; ++i) s += (double)K[b * 16 + i * 4 + k] * dP[i * 4 + j];
                d_T[b * 16 + k * 4 + j] = (float)s;
            }
    }
}

/* Bilinear resize, F.interpolate(mode='bilinear', align_corners=False), trainer.py:512. in [N,h,w] -> out [N,H,W] */
static inline void interp_idx(int o, int in, int out, int *i0, int *i1, float *l1) {
    float scale = (float)in / (float)out;
    float s = scale * ((float)o + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    int a = (int)s;
    if (a > in - 1) a = in - 1;
    *i0 = a;
    *i1 = a < in - 1 ? a + 1 : a;
    *l1 = s - (float)a;
}
void mdo_resize_bilinear_fwd(const float *in, int N, int h, int w, int H, int W, float *out) {
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < H; ++y) {
            int y0, y1; float ly;
            interp_idx(y, h, H, &y0, &y1, &ly);
            for (int x = 0; x < W; ++x) {
                int x0, x1; float lx;
                interp_idx(x, w, W, &x0, &x1, &lx);
                const float *s = in + (size_t)n * h * w;
                /* the four taps in the order the reference's build evaluates them (found like the matrix products' order,
                 * tools/diag/op_order_search.py: of the 24 fused-multiply-add chains over the taps only this one reproduces
                 * F.interpolate and the depth_0_{1,2,3} maps of tests/golden/losses_mono.npz bit for bit): the north-east
                 * product first, then nw, sw, se accumulated with fused multiply-adds.  The weights are products of the two
                 * axes' weights (exact for the power-of-two pyramid ratios the trainer uses). */
                float wy0 = 1.f - ly, wx0 = 1.f - lx;
                float o = (wy0 * lx) * s[y0 * w + x1];
                o = fmaf(wy0 * wx0, s[y0 * w + x0], o);
                o = fmaf(ly * wx0, s[y1 * w + x0], o);
                o = fmaf(ly * lx, s[y1 * w + x1], o);
                out[((size_t)n * H + y) * W + x] = o;
            }
        }
}
void mdo_resize_bilinear_bwd(const float *gout, int N, int h, int w, int H, int W, float *gin) {
    memset(gin, 0, sizeof(float) * (size_t)N * h * w);
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < H; ++y) {
            int y0, y1; float ly;
            interp_idx(y, h, H, &y0, &y1, &ly);
            for (int x = 0; x < W; ++x) {
                int x0, x1; float lx;
                interp_idx(x, w, W, &x0, &x1, &lx);
                float g = gout[((size_t)n * H + y) * W + x];
                float *s = gin + (size_t)n * h * w;
                s[y0 * w + x0] += g * (1.f - ly) * (1.f - lx);
                s[y0 * w + x1] += g * (1.f - ly) * lx;
                s[y1 * w + x0] += g * ly * (1.f - lx);
                s[y1 * w + x1] += g * ly * lx;
            }
        }
}

/* ------------------------------------------------------------------ SSIM + L1 reprojection loss */

static inline int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

typedef struct { float mux, muy, ex2, ey2, exy; } ssim_stats_t;

static inline ssim_stats_t ssim_stats(const float *x, const float *y, int H, int W, int py, int px) {
    ssim_stats_t s = {0, 0, 0, 0, 0};
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            int yy = reflect1(py + dy, H), xx = reflect1(px + dx, W); /* ReflectionPad2d(1), layers.py:658 */
            float a = x[yy * W + xx], b = y[yy * W + xx];
            s.mux += a; s.muy += b; s.ex2 += a * a; s.ey2 += b * b; s.exy += a * b;
        }
    s.mux /= 9.f; s.muy /= 9.f; s.ex2 /= 9.f; s.ey2 /= 9.f; s.exy /= 9.f; /* AvgPool2d(3,1) */
    return s;
}

#define SSIM_C1 (0.01f * 0.01f)
#define SSIM_C2 (0.03f * 0.03f)

static inline float ssim_value(const ssim_stats_t *s, float *n_out, float *d_out) {
    float sx = s->ex2 - s->mux * s->mux, sy = s->ey2 - s->muy * s->muy, sxy = s->exy - s->mux * s->muy;
    float n = (2.f * s->mux * s->muy + SSIM_C1) * (2.f * sxy + SSIM_C2);
    float d = (s->mux * s->mux + s->muy * s->muy + SSIM_C1) * (sx + sy + SSIM_C2);
    if (n_out) { *n_out = n; *d_out = d; }
    float v = (1.f - n / d) / 2.f;
    return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); /* torch.clamp(.,0,1), layers.py:677 */
}

/* SSIM.forward, layers.py:663-677.  x,y,out [N,H,W] (N = B*C planes) */
void mdo_ssim(const float *x, const float *y, int N, int H, int W, float *out) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (int py = 0; py < H; ++py)
            for (int px = 0; px < W; ++px) {
                ssim_stats_t s = ssim_stats(x + (size_t)n * H * W, y + (size_t)n * H * W, H, W, py, px);
                out[((size_t)n * H + py) * W + px] = ssim_value(&s, NULL, NULL);
            }
}

/* compute_reprojection_loss, trainer.py:535-550: ssim_w*mean_c SSIM(pred,target) + (1-ssim_w)*mean_c|target-pred|.
 * no_ssim -> L1 only.  pred,target [B,C,H,W]; out [B,H,W]. */
void mdo_reproj_loss_fwd(const float *pred, const float *target, int B, int C, int H, int W, float ssim_w,
                         int no_ssim, float *out) {
    int HW = H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int py = 0; py < H; ++py)
            for (int px = 0; px < W; ++px) {
                int p = py * W + px;
                float l1 = 0.f, ss = 0.f;
                for (int c = 0; c < C; ++c) {
                    const float *xp = pred + ((size_t)b * C + c) * HW, *yp = target + ((size_t)b * C + c) * HW;
                    l1 += fabsf(yp[p] - xp[p]);
                    if (!no_ssim) {
                        ssim_stats_t s = ssim_stats(xp, yp, H, W, py, px);
                        ss += ssim_value(&s, NULL, NULL);
                    }
                }
                l1 /= (float)C;
                ss /= (float)C;
                out[(size_t)b * HW + p] = no_ssim ? l1 : ssim_w * ss + (1.f - ssim_w) * l1;
            }
}

/* Autograd of mdo_reproj_loss_fwd w.r.t. pred.  gout [B,H,W]; d_pred [B,C,H,W].
 * Two passes, both over (plane, row) so that one sample keeps all cores busy and no atomics are needed: (1) per window p the
 * three coefficients of d SSIM-term / d(mu_x, E[x^2], E[xy]) times the upstream factor; (2) per pixel q the sum over the
 * windows that contain q -- with ReflectionPad2d(1) a border window reads some pixels more than once: multiplicity
 * refl_mult() per axis -- plus the L1 term. */
static inline int refl_mult(int p, int q, int n) {
    int m = 0;
    for (int d = -1; d <= 1; ++d) m += reflect1(p + d, n) == q;
    return m;
}
void mdo_reproj_loss_bwd(const float *gout, const float *pred, const float *target, int B, int C, int H, int W,
                         float ssim_w, int no_ssim, float *d_pred) {
    int HW = H * W;
    int use_ssim = !(no_ssim || ssim_w == 0.f);
    float *co = use_ssim ? (float *)malloc(sizeof(float) * 3 * (size_t)B * C * HW) : NULL;
    if (use_ssim) {
#pragma omp parallel for collapse(3) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < C; ++c)
                for (int py = 0; py < H; ++py) {
                    const float *xp = pred + ((size_t)b * C + c) * HW, *yp = target + ((size_t)b * C + c) * HW;
                    float *ca_ = co + 3 * ((size_t)b * C + c) * HW, *cb_ = ca_ + HW, *cc_ = cb_ + HW;
                    for (int px = 0; px < W; ++px) {
                        int p = py * W + px;
                        ca_[p] = cb_[p] = cc_[p] = 0.f;
                        ssim_stats_t s = ssim_stats(xp, yp, H, W, py, px);
                        float n, d;
                        ssim_value(&s, &n, &d);
                        float raw = (1.f - n / d) / 2.f;
                        if (raw < 0.f || raw > 1.f) continue; /* clamp: zero gradient outside [0,1] */
                        float gs = gout[(size_t)b * HW + p] * ssim_w / (float)C;
                        float sx = s.ex2 - s.mux * s.mux, sy = s.ey2 - s.muy * s.muy, sxy = s.exy - s.mux * s.muy;
                        float A1 = 2.f * s.mux * s.muy + SSIM_C1, A2 = 2.f * sxy + SSIM_C2;
                        float B1 = s.mux * s.mux + s.muy * s.muy + SSIM_C1, B2 = sx + sy + SSIM_C2;
                        /* s = (1 - n/d)/2 ; ds = -(dn*d - n*dd)/(2 d^2) */
                        float dn_dmux = 2.f * s.muy * A2 - 2.f * s.muy * A1;
                        float dd_dmux = 2.f * s.mux * B2 - 2.f * s.mux * B1;
                        ca_[p] = gs * (-0.5f * (dn_dmux * d - n * dd_dmux) / (d * d)); /* d s / d mu_x   */
                        cb_[p] = gs * (0.5f * n * B1 / (d * d));                        /* d s / d E[x^2] */
                        cc_[p] = gs * (-0.5f * (2.f * A1) / d);                         /* d s / d E[xy]  */
                    }
                }
    }
#pragma omp parallel for collapse(3) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int qy = 0; qy < H; ++qy) {
                const float *xp = pred + ((size_t)b * C + c) * HW, *yp = target + ((size_t)b * C + c) * HW;
                float *dx = d_pred + ((size_t)b * C + c) * HW;
                const float *ca_ = use_ssim ? co + 3 * ((size_t)b * C + c) * HW : NULL, *cb_ = ca_ + HW, *cc_ = cb_ + HW;
                float wl1 = no_ssim ? 1.f : (1.f - ssim_w);
                for (int qx = 0; qx < W; ++qx) {
                    int q = qy * W + qx;
                    float diff = yp[q] - xp[q];
                    float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
                    float acc = -sg * gout[(size_t)b * HW + q] * wl1 / (float)C;
                    if (use_ssim) {
                        for (int py = qy - 2; py <= qy + 2; ++py) {
                            if (py < 0 || py >= H) continue;
                            int my = refl_mult(py, qy, H);
                            if (!my) continue;
                            for (int px = qx - 2; px <= qx + 2; ++px) {
                                if (px < 0 || px >= W) continue;
                                int mx = refl_mult(px, qx, W);
                                if (!mx) continue;
                                int p = py * W + px;
                                acc += (float)(my * mx) * ((ca_[p] + 2.f * cb_[p] * xp[q] + cc_[p] * yp[q]) / 9.f);
                            }
                        }
                    }
                    dx[q] = acc;
                }
            }
    free(co);
}

/* ------------------------------------------------------------------ min / automask / masked mean */

/* trainer.py:687-709 (mono) and 630-662 (MVS).  reproj [B,N,HW]; ident [B,N,HW] or NULL (no automask);
 * noise [B,HW] or NULL (the randn*1e-5 tie-break, trainer.py:698 -- passed in already scaled);
 * ext_mask [B,HW] or NULL multiplies the mask (MVS conf/dist masks, trainer.py:650-656).
 * mvs_mode!=0: the automask is computed and then replaced by ones (trainer.py:642->647).
 * Outputs: min_reproj [B,HW], mask [B,HW], loss[0] = sum(min*mask)/(sum(mask)+1e-7). */
void mdo_masked_min_fwd(const float *reproj, const float *ident, const float *noise, const float *ext_mask, int B,
                        int N, int HW, int mvs_mode, float *min_reproj, float *mask, float *loss) {
    double num = 0.0, den = 0.0;
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < HW; ++p) {
            float r = reproj[((size_t)b * N) * HW + p];
            for (int f = 1; f < N; ++f) { float v = reproj[((size_t)b * N + f) * HW + p]; if (v < r) r = v; }
            float m = 1.f;
            if (ident && !mvs_mode) {
                float id = ident[((size_t)b * N) * HW + p];
                for (int f = 1; f < N; ++f) { float v = ident[((size_t)b * N + f) * HW + p]; if (v < id) id = v; }
                if (noise) id += noise[(size_t)b * HW + p];
                m = (r <= id) ? 1.f : 0.f; /* argmin over [reproj, identity] == 0 (first index wins ties) */
            }
            if (ext_mask) m *= ext_mask[(size_t)b * HW + p];
            min_reproj[(size_t)b * HW + p] = r;
            mask[(size_t)b * HW + p] = m;
            num += (double)(r * m);
            den += (double)m;
        }
    loss[0] = (float)(num / (den + 1e-7));
}

/* d loss / d reproj: routed to the arg-min frame (first index on ties). gloss scalar. */
void mdo_masked_min_bwd(float gloss, const float *reproj, const float *mask, int B, int N, int HW,
                        float *d_reproj) {
    double den = 0.0;
    for (size_t i = 0; i < (size_t)B * HW; ++i) den += (double)mask[i];
    float scale = gloss / (float)(den + 1e-7);
    memset(d_reproj, 0, sizeof(float) * (size_t)B * N * HW);
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < HW; ++p) {
            int am = 0;
            float r = reproj[((size_t)b * N) * HW + p];
            for (int f = 1; f < N; ++f) { float v = reproj[((size_t)b * N + f) * HW + p]; if (v < r) { r = v; am = f; } }
            d_reproj[((size_t)b * N + am) * HW + p] = scale * mask[(size_t)b * HW + p];
        }
}

/* ------------------------------------------------------------------ edge-aware smoothness */

/* get_smooth_loss (layers.py:630-643) on the mean-normalised disparity (trainer.py:712-714, or 666-669 for
 * the MVS depth).  disp [B,h,w], img [B,3,h,w]; normalize!=0 divides by (per-sample mean + 1e-7) first.
 * Returns the scalar. */
float mdo_smooth_fwd(const float *disp, const float *img, int B, int Ci, int h, int w, int normalize) {
    int hw = h * w;
    double sx = 0.0, sy = 0.0;
    for (int b = 0; b < B; ++b) {
        const float *d = disp + (size_t)b * hw;
        float dn = 1.f;
        if (normalize) {
            double m = 0.0;
            for (int p = 0; p < hw; ++p) m += d[p];
            dn = (float)(m / hw) + 1e-7f;
        }
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                int p = y * w + x;
                float a = d[p] / dn;
                if (x + 1 < w) {
                    float gi = 0.f;
                    for (int c = 0; c < Ci; ++c) {
                        const float *im = img + ((size_t)b * Ci + c) * hw;
                        gi += fabsf(im[p] - im[p + 1]);
                    }
                    sx += (double)(fabsf(a - d[p + 1] / dn) * expf(-gi / (float)Ci));
                }
                if (y + 1 < h) {
                    float gi = 0.f;
                    for (int c = 0; c < Ci; ++c) {
                        const float *im = img + ((size_t)b * Ci + c) * hw;
                        gi += fabsf(im[p] - im[p + w]);
                    }
                    sy += (double)(fabsf(a - d[p + w] / dn) * expf(-gi / (float)Ci));
                }
            }
    }
    return (float)(sx / ((double)B * h * (w - 1)) + sy / ((double)B * (h - 1) * w));
}

/* Autograd of mdo_smooth_fwd w.r.t. disp. gloss scalar; d_disp [B,h,w]. */
void mdo_smooth_bwd(float gloss, const float *disp, const float *img, int B, int Ci, int h, int w, int normalize,
                    float *d_disp) {
    int hw = h * w;
    float cx = gloss / (float)((double)B * h * (w - 1)), cy = gloss / (float)((double)B * (h - 1) * w);
    float *gn = (float *)malloc(sizeof(float) * hw);
    for (int b = 0; b < B; ++b) {
        const float *d = disp + (size_t)b * hw;
        float dn = 1.f;
        if (normalize) {
            double m = 0.0;
            for (int p = 0; p < hw; ++p) m += d[p];
            dn = (float)(m / hw) + 1e-7f;
        }
        memset(gn, 0, sizeof(float) * hw);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                int p = y * w + x;
                float a = d[p] / dn;
                if (x + 1 < w) {
                    float gi = 0.f;
                    for (int c = 0; c < Ci; ++c) {
                        const float *im = img + ((size_t)b * Ci + c) * hw;
                        gi += fabsf(im[p] - im[p + 1]);
                    }
                    float df = a - d[p + 1] / dn;
                    float s = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * expf(-gi / (float)Ci) * cx;
                    gn[p] += s; gn[p + 1] -= s;
                }
                if (y + 1 < h) {
                    float gi = 0.f;
                    for (int c = 0; c < Ci; ++c) {
                        const float *im = img + ((size_t)b * Ci + c) * hw;
                        gi += fabsf(im[p] - im[p + w]);
                    }
                    float df = a - d[p + w] / dn;
                    float s = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * expf(-gi / (float)Ci) * cy;
                    gn[p] += s; gn[p + w] -= s;
                }
            }
        /* nd = d / (mean(d) + 1e-7): d_d[q] = gn[q]/dn - (sum_p gn[p] d[p]) / dn^2 / hw */
        double dot = 0.0;
        if (normalize)
            for (int p = 0; p < hw; ++p) dot += (double)gn[p] * d[p];
        float corr = normalize ? (float)(dot / ((double)dn * dn) / hw) : 0.f;
        for (int p = 0; p < hw; ++p) d_disp[(size_t)b * hw + p] = gn[p] / dn - corr;
    }
    free(gn);
}

/* ------------------------------------------------------------------ post-volume ops ("next" rows, forward) */

/* localmax, layers.py:796-812.  prob [B,D,hw]; min_inv, max_inv [B,hw] (the caller passes 1/hyp[:,-1] and
 * 1/hyp[:,0], trainer.py:371 -- index d decodes to hypothesis D-1-d).  out [B,hw]. */
void mdo_localmax(const float *prob, int B, int D, int hw, int radius, const float *min_inv, const float *max_inv,
                  float *out) {
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < hw; ++p) {
            const float *pr = prob + (size_t)b * D * hw + p;
            int am = 0;
            float mx = pr[0];
            for (int d = 1; d < D; ++d) if (pr[(size_t)d * hw] > mx) { mx = pr[(size_t)d * hw]; am = d; }
            float num = 0.f, den = 1e-6f;
            for (int i = -radius; i <= radius; ++i) {
                int idx = am + i;
                idx = idx < 0 ? 0 : (idx > D - 1 ? D - 1 : idx); /* clamp: border bins are counted twice */
                float v = pr[(size_t)idx * hw];
                num += (float)idx * v;
                den += v;
            }
            float nrm = (num / den) / (float)(D - 1);
            float a = min_inv[(size_t)b * hw + p], bb = max_inv[(size_t)b * hw + p];
            out[(size_t)b * hw + p] = 1.f / (a + nrm * (bb - a));
        }
}

/* entropy, layers.py:862-863: sum_d -p * log(clamp(p, 1e-9, 1)).  prob [B,D,hw] -> out [B,hw] */
void mdo_entropy(const float *prob, int B, int D, int hw, float *out) {
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < hw; ++p) {
            float s = 0.f;
            for (int d = 0; d < D; ++d) {
                float v = prob[((size_t)b * D + d) * hw + p];
                float c = v < 1e-9f ? 1e-9f : (v > 1.f ? 1.f : v);
                s += -v * logf(c);
            }
            out[(size_t)b * hw + p] = s;
        }
}

/* softmax over D, F.softmax(cost, 1), trainer.py:367.  in/out [B,D,hw] */
void mdo_softmax_d(const float *in, int B, int D, int hw, float *out) {
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < hw; ++p) {
            float mx = -INFINITY, den = 0.f;
            for (int d = 0; d < D; ++d) { float v = in[((size_t)b * D + d) * hw + p]; if (v > mx) mx = v; }
            for (int d = 0; d < D; ++d) den += expf(in[((size_t)b * D + d) * hw + p] - mx);
            for (int d = 0; d < D; ++d)
                out[((size_t)b * D + d) * hw + p] = expf(in[((size_t)b * D + d) * hw + p] - mx) / den;
        }
}

/* convex_upsample, layers.py:200-214.  depth [B,h,w]; mask [B,9*s*s,h,w] with s = 2**scale;
 * out [B, s*h, s*w]: softmax over the 9 taps, zero-padded 3x3 unfold. */
void mdo_convex_upsample(const float *depth, const float *mask, int B, int h, int w, int scale, float *out) {
    int s = 1 << scale, hw = h * w;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
                for (int i = 0; i < s; ++i)
                    for (int j = 0; j < s; ++j) {
                        float m[9], mx = -INFINITY, den = 0.f, acc = 0.f;
                        for (int k = 0; k < 9; ++k) {
                            /* mask.view(B, 9, s, s, H, W): channel = (k*s + i)*s + j */
                            m[k] = mask[(((size_t)b * 9 * s * s) + ((size_t)k * s + i) * s + j) * hw + y * w + x];
                            if (m[k] > mx) mx = m[k];
                        }
                        for (int k = 0; k < 9; ++k) { m[k] = expf(m[k] - mx); den += m[k]; }
                        for (int k = 0; k < 9; ++k) {
                            int yy = y + k / 3 - 1, xx = x + k % 3 - 1; /* F.unfold([3,3], padding=1) */
                            float v = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? depth[(size_t)b * hw + yy * w + xx] : 0.f;
                            acc += (m[k] / den) * v;
                        }
                        out[((size_t)b * s * h + (y * s + i)) * ((size_t)s * w) + x * s + j] = acc;
                    }
}

/* reg3d's last layer `prob`: nn.Conv3d(base_channels, 1, 3, stride=1, padding=1, bias=False)
 * (networks/resnet_encoder.py:254, applied :277).  Plain NCDHW indexing as the reference sees it:
 * x [B,C,D,H,W]; wt [1,C,3,3,3]; y [B,1,D,H,W].  fp64 accumulation (this is the checker, not the product). */
void mdo_conv3d_c1_fwd(const float *x, const float *wt, int B, int C, int D, int H, int W, float *y) {
    size_t hw = (size_t)H * W, dhw = (size_t)D * hw;
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d)
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < W; ++w) {
                    double acc = 0.0;
                    for (int c = 0; c < C; ++c)
                        for (int kd = 0; kd < 3; ++kd) {
                            int dd = d + kd - 1;
                            if (dd < 0 || dd >= D) continue;
                            for (int kh = 0; kh < 3; ++kh) {
                                int hh = h + kh - 1;
                                if (hh < 0 || hh >= H) continue;
                                for (int kw = 0; kw < 3; ++kw) {
                                    int ww = w + kw - 1;
                                    if (ww < 0 || ww >= W) continue;
                                    acc += (double)x[((size_t)b * C + c) * dhw + dd * hw + (size_t)hh * W + ww] *
                                           (double)wt[c * 27 + (kd * 3 + kh) * 3 + kw];
                                }
                            }
                        }
                    y[(size_t)b * dhw + d * hw + (size_t)h * W + w] = (float)acc;
                }
}

/* Adjoint of the above: gy [B,1,D,H,W] -> dx [B,C,D,H,W] and dwt [1,C,3,3,3] (either may be NULL). */
void mdo_conv3d_c1_bwd(const float *gy, const float *x, const float *wt, int B, int C, int D, int H, int W, float *dx,
                       float *dwt) {
    size_t hw = (size_t)H * W, dhw = (size_t)D * hw;
    if (dx) {
#pragma omp parallel for collapse(2)
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < C; ++c)
                for (int d = 0; d < D; ++d)
                    for (int h = 0; h < H; ++h)
                        for (int w = 0; w < W; ++w) {
                            double acc = 0.0;
                            for (int kd = 0; kd < 3; ++kd) {
                                int dd = d - kd + 1; /* output voxel that read x[d] through tap kd */
                                if (dd < 0 || dd >= D) continue;
                                for (int kh = 0; kh < 3; ++kh) {
                                    int hh = h - kh + 1;
                                    if (hh < 0 || hh >= H) continue;
                                    for (int kw = 0; kw < 3; ++kw) {
                                        int ww = w - kw + 1;
                                        if (ww < 0 || ww >= W) continue;
                                        acc += (double)gy[(size_t)b * dhw + dd * hw + (size_t)hh * W + ww] *
                                               (double)wt[c * 27 + (kd * 3 + kh) * 3 + kw];
                                    }
                                }
                            }
                            dx[((size_t)b * C + c) * dhw + d * hw + (size_t)h * W + w] = (float)acc;
                        }
    }
    if (dwt) {
#pragma omp parallel for
        for (int o = 0; o < C * 27; ++o) {
            int c = o / 27, k = o % 27, kd = k / 9, kh = (k / 3) % 3, kw = k % 3;
            double acc = 0.0;
            for (int b = 0; b < B; ++b)
                for (int d = 0; d < D; ++d) {
                    int dd = d + kd - 1;
                    if (dd < 0 || dd >= D) continue;
                    for (int h = 0; h < H; ++h) {
                        int hh = h + kh - 1;
                        if (hh < 0 || hh >= H) continue;
                        for (int w = 0; w < W; ++w) {
                            int ww = w + kw - 1;
                            if (ww < 0 || ww >= W) continue;
                            acc += (double)x[((size_t)b * C + c) * dhw + dd * hw + (size_t)hh * W + ww] *
                                   (double)gy[(size_t)b * dhw + d * hw + (size_t)h * W + w];
                        }
                    }
                }
            dwt[o] = (float)acc;
        }
    }
}

/* ConvBnReLU3D.conv of reg3d.conv0: nn.Conv3d(Ci, Co, 3, stride=1, padding=1, bias=False)
 * (networks/resnet_encoder.py:231 via module.py's ConvBnReLU3D, applied :258).  NCDHW as the reference sees it:
 * x [B,Ci,D,H,W]; wt [Co,Ci,3,3,3]; y, gy [B,Co,D,H,W].  fp64 accumulation.  Any of y / dx / dwt may be NULL. */
void mdo_conv3d(const float *x, const float *wt, const float *gy, int B, int Ci, int Co, int D, int H, int W, float *y,
                float *dx, float *dwt) {
    size_t hw = (size_t)H * W, dhw = (size_t)D * hw;
    if (y) {
#pragma omp parallel for collapse(3)
        for (int b = 0; b < B; ++b)
            for (int co = 0; co < Co; ++co)
                for (int d = 0; d < D; ++d)
                    for (int h = 0; h < H; ++h)
                        for (int w = 0; w < W; ++w) {
                            double acc = 0.0;
                            for (int ci = 0; ci < Ci; ++ci)
                                for (int k = 0; k < 27; ++k) {
                                    int dd = d + k / 9 - 1, hh = h + (k / 3) % 3 - 1, ww = w + k % 3 - 1;
                                    if (dd < 0 || dd >= D || hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
                                    acc += (double)x[((size_t)b * Ci + ci) * dhw + dd * hw + (size_t)hh * W + ww] *
                                           (double)wt[((size_t)co * Ci + ci) * 27 + k];
                                }
                            y[((size_t)b * Co + co) * dhw + d * hw + (size_t)h * W + w] = (float)acc;
                        }
    }
    if (dx) {
#pragma omp parallel for collapse(3)
        for (int b = 0; b < B; ++b)
            for (int ci = 0; ci < Ci; ++ci)
                for (int d = 0; d < D; ++d)
                    for (int h = 0; h < H; ++h)
                        for (int w = 0; w < W; ++w) {
                            double acc = 0.0;
                            for (int co = 0; co < Co; ++co)
                                for (int k = 0; k < 27; ++k) {
                                    int dd = d - k / 9 + 1, hh = h - (k / 3) % 3 + 1, ww = w - k % 3 + 1;
                                    if (dd < 0 || dd >= D || hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
                                    acc += (double)gy[((size_t)b * Co + co) * dhw + dd * hw + (size_t)hh * W + ww] *
                                           (double)wt[((size_t)co * Ci + ci) * 27 + k];
                                }
                            dx[((size_t)b * Ci + ci) * dhw + d * hw + (size_t)h * W + w] = (float)acc;
                        }
    }
    if (dwt) {
#pragma omp parallel for
        for (int o = 0; o < Co * Ci * 27; ++o) {
            int k = o % 27, ci = (o / 27) % Ci, co = o / (27 * Ci);
            int kd = k / 9, kh = (k / 3) % 3, kw = k % 3;
            double acc = 0.0;
            for (int b = 0; b < B; ++b)
                for (int d = 0; d < D; ++d) {
                    int dd = d + kd - 1;
                    if (dd < 0 || dd >= D) continue;
                    for (int h = 0; h < H; ++h) {
                        int hh = h + kh - 1;
                        if (hh < 0 || hh >= H) continue;
                        for (int w = 0; w < W; ++w) {
                            int ww = w + kw - 1;
                            if (ww < 0 || ww >= W) continue;
                            acc += (double)x[((size_t)b * Ci + ci) * dhw + dd * hw + (size_t)hh * W + ww] *
                                   (double)gy[((size_t)b * Co + co) * dhw + d * hw + (size_t)h * W + w];
                        }
                    }
                }
            dwt[o] = (float)acc;
        }
    }
}

/* Adjoint of mdo_transformation_from_parameters (what torch autograd computes for layers.py:412-429): gT [B,16] ->
 * d_axisangle, d_translation [B,3].  fp64 arithmetic.  d|v|/dv is taken as 0 at v = 0 (torch's norm backward). */
void mdo_transformation_from_parameters_bwd(const float *gT, const float *aa, const float *tr, int B, int invert,
                                            float *d_aa, float *d_tr) {
    for (int b = 0; b < B; ++b) {
        double v[3] = {aa[b * 3], aa[b * 3 + 1], aa[b * 3 + 2]}, t[3] = {tr[b * 3], tr[b * 3 + 1], tr[b * 3 + 2]};
        const float *g = gT + b * 16;
        double angle = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), s = angle + 1e-7;
        double x = v[0] / s, y = v[1] / s, z = v[2] / s, ca = cos(angle), sa = sin(angle), C = 1.0 - ca;
        double R[9] = {x * x * C + ca, x * y * C - z * sa, z * x * C + y * sa, x * y * C + z * sa, y * y * C + ca,
                       y * z * C - x * sa, z * x * C - y * sa, y * z * C + x * sa, z * z * C + ca};
        double G[9], gt[3] = {0, 0, 0};
        if (invert) { /* M[i][j] = R[j][i], M[i][3] = -sum_j R[j][i] t_j */
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    G[j * 3 + i] = (double)g[i * 4 + j] - (double)g[i * 4 + 3] * t[j];
                    gt[j] -= (double)g[i * 4 + 3] * R[j * 3 + i];
                }
        } else {
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) G[i * 3 + j] = g[i * 4 + j];
                gt[i] = g[i * 4 + 3];
            }
        }
        double s01 = G[1] + G[3], s02 = G[2] + G[6], s12 = G[5] + G[7];
        double gx = G[0] * 2 * x * C + s01 * y * C + s02 * z * C + (G[7] - G[5]) * sa;
        double gy = G[4] * 2 * y * C + s01 * x * C + s12 * z * C + (G[2] - G[6]) * sa;
        double gz = G[8] * 2 * z * C + s02 * x * C + s12 * y * C + (G[3] - G[1]) * sa;
        double gC = G[0] * x * x + G[4] * y * y + G[8] * z * z + s01 * x * y + s02 * z * x + s12 * y * z;
        double gca = G[0] + G[4] + G[8] - gC;
        double gsa = (G[3] - G[1]) * z + (G[2] - G[6]) * y + (G[7] - G[5]) * x;
        double gth = -gca * sa + gsa * ca;
        double dotgv = gx * v[0] + gy * v[1] + gz * v[2];
        double k = angle > 0 ? (gth - dotgv / (s * s)) / angle : 0.0;
        d_aa[b * 3] = (float)(gx / s + k * v[0]);
        d_aa[b * 3 + 1] = (float)(gy / s + k * v[1]);
        d_aa[b * 3 + 2] = (float)(gz / s + k * v[2]);
        for (int i = 0; i < 3; ++i) d_tr[b * 3 + i] = (float)gt[i];
    }
}
