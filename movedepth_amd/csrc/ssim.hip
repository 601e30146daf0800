// SSIM + L1 reprojection loss (reference layers.py:646-677 SSIM, trainer.py:535-550 compute_reprojection_loss).
//
// The reference runs 2 reflection pads, 5 AvgPool2d(3,1) and ~20 elementwise kernels (about 30 passes over the
// (B,3,H,W) images) per call, 20 calls per step.  Here: one pass.
//   forward : a wave owns a 62-column x RY-row strip (lanes 0 and 63 are the halo columns).  Rows roll through
//             registers (each image row is loaded once per strip, not three times); the horizontal 3-sum of the
//             five window moments comes from neighbouring lanes with wave shuffles -- no LDS, no padded copies.
//   backward: d/dpred.  Per pixel p the SSIM derivative collapses to three coefficients (w.r.t. mu_x, E[x^2],
//             E[xy]); the gradient at q is then a reflection-adjoint 3x3 box sum of the coefficient maps:
//             d_pred[q] = (boxT(A) + 2 x_q boxT(B) + y_q boxT(C))[q] / 9 + L1 term.  One 32x8 tile per block,
//             coefficients staged in LDS (halo 1), inputs staged with halo 2.
#include "md_photo.hpp"

namespace {

using namespace mdp;
constexpr int RY = 8;  // rows per wave strip

__device__ __forceinline__ float hsum3(float v) { return __shfl_up(v, 1, 64) + v + __shfl_down(v, 1, 64); }

// MODE 0: SSIM map per channel (out [B,C,H,W]);  MODE 1: reprojection loss (out [B,1,H,W]).
template <int MODE, int C>
__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float *__restrict__ X, const float *__restrict__ Y, int H,
                                                       int W, float ssim_w, int no_ssim, float *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z;
    const int xcol = blockIdx.x * 62 - 1 + lane;
    const int xr = clampi(reflect1(xcol, W), 0, W - 1);
    const int row0 = (blockIdx.y * 4 + wave) * RY;
    if (row0 >= H) return;
    const size_t HW = (size_t)H * W;
    const float *xb = X + (size_t)b * C * HW, *yb = Y + (size_t)b * C * HW;
    // rows roll through registers together with their left / right neighbours (wave shuffles), so that the nine taps of a window
    // are added ONE BY ONE in the reference's order -- AvgPool2d walks the window row-major and adds already-rounded products
    // (layers.py:663-671).  sigma = E[x^2] - mu^2 cancels to 1e-3 of its terms on smooth images: with the columns summed first
    // (the first version of this kernel) the loss of a 192 x 640 step sat 4e-4 from the reference's.  No contraction below.
    float xv[3][C], yv[3][C], xl[3][C], xg[3][C], yl[3][C], yg[3][C];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int rr = clampi(reflect1(row0 - 1 + s, H), 0, H - 1);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            xv[s][c] = xb[c * HW + (size_t)rr * W + xr];
            yv[s][c] = yb[c * HW + (size_t)rr * W + xr];
            xl[s][c] = __shfl_up(xv[s][c], 1, 64); xg[s][c] = __shfl_down(xv[s][c], 1, 64);
            yl[s][c] = __shfl_up(yv[s][c], 1, 64); yg[s][c] = __shfl_down(yv[s][c], 1, 64);
        }
    }
    const bool writer = lane >= 1 && lane <= 62 && xcol < W;
#pragma unroll 1
    for (int i = 0; i < RY; ++i) {
        const int r = row0 + i;
        if (r >= H) break;
        const int rr = clampi(reflect1(r + 1, H), 0, H - 1);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            xv[2][c] = xb[c * HW + (size_t)rr * W + xr];
            yv[2][c] = yb[c * HW + (size_t)rr * W + xr];
            xl[2][c] = __shfl_up(xv[2][c], 1, 64); xg[2][c] = __shfl_down(xv[2][c], 1, 64);
            yl[2][c] = __shfl_up(yv[2][c], 1, 64); yg[2][c] = __shfl_down(yv[2][c], 1, 64);
        }
        float l1 = 0.f, ss = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            Moments m = {0.f, 0.f, 0.f, 0.f, 0.f};
            {
#pragma clang fp contract(off)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float xa[3] = {xl[dy][c], xv[dy][c], xg[dy][c]}, ya[3] = {yl[dy][c], yv[dy][c], yg[dy][c]};
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float xx = xa[dx] * xa[dx], yy = ya[dx] * ya[dx], xy = xa[dx] * ya[dx];
                        m.mux += xa[dx]; m.muy += ya[dx]; m.ex2 += xx; m.ey2 += yy; m.exy += xy;
                    }
                }
                m.mux /= 9.f; m.muy /= 9.f; m.ex2 /= 9.f; m.ey2 /= 9.f; m.exy /= 9.f;
            }
            float s = ssim_from(m, nullptr, nullptr);
            s = fminf(fmaxf(s, 0.f), 1.f);  // torch.clamp(., 0, 1)
            if (MODE == 0) {
                if (writer) out[((size_t)b * C + c) * HW + (size_t)r * W + xcol] = s;
            } else {
                ss += s;
                l1 += fabsf(yv[1][c] - xv[1][c]);
            }
        }
        if (MODE == 1 && writer) {
            l1 /= (float)C;
            ss /= (float)C;
            out[(size_t)b * HW + (size_t)r * W + xcol] = no_ssim ? l1 : ssim_w * ss + (1.f - ssim_w) * l1;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            xv[0][c] = xv[1][c]; xv[1][c] = xv[2][c]; xl[0][c] = xl[1][c]; xl[1][c] = xl[2][c]; xg[0][c] = xg[1][c]; xg[1][c] = xg[2][c];
            yv[0][c] = yv[1][c]; yv[1][c] = yv[2][c]; yl[0][c] = yl[1][c]; yl[1][c] = yl[2][c]; yg[0][c] = yg[1][c]; yg[1][c] = yg[2][c];
        }
    }
}

constexpr int BT_W = 32, BT_H = 8;  // backward tile

__global__ __launch_bounds__(256) void reproj_bwd_kernel(const float *__restrict__ gout, const float *__restrict__ X,
                                                         const float *__restrict__ Y, int C, int H, int W, float ssim_w,
                                                         int no_ssim, float *__restrict__ d_pred) {
    constexpr int IW = BT_W + 4, IH = BT_H + 4;  // inputs, halo 2
    constexpr int CW = BT_W + 2, CH = BT_H + 2;  // coefficients, halo 1
    __shared__ float xs[IH * IW], ys[IH * IW];
    __shared__ float cA[CH * CW], cB[CH * CW], cC[CH * CW];
    // one channel per workgroup (grid z = B*C): the channels are independent in the backward, and a serial loop over them
    // with three barriers each left the 2880 workgroups latency-bound (36 us for 10 MB)
    const int b = blockIdx.z / C, tid = threadIdx.x;
    const int tx0 = blockIdx.x * BT_W, ty0 = blockIdx.y * BT_H;
    const size_t HW = (size_t)H * W;
    const int qx = tx0 + tid % BT_W, qy = ty0 + tid / BT_W;
    const bool qvalid = qx < W && qy < H;
    const bool use_ssim = !no_ssim && ssim_w != 0.f;
    const float wl1 = no_ssim ? 1.f : (1.f - ssim_w);
    {
        const int c = blockIdx.z % C;
        const float *xp = X + ((size_t)b * C + c) * HW, *yp = Y + ((size_t)b * C + c) * HW;
        float gA = 0.f, gB = 0.f, gC = 0.f;
        if (use_ssim) {
            for (int i = tid; i < IH * IW; i += 256) {
                const int yy = clampi(reflect1(ty0 - 2 + i / IW, H), 0, H - 1);
                const int xx = clampi(reflect1(tx0 - 2 + i % IW, W), 0, W - 1);
                xs[i] = xp[(size_t)yy * W + xx];
                ys[i] = yp[(size_t)yy * W + xx];
            }
            __syncthreads();
            for (int i = tid; i < CH * CW; i += 256) {
                const int cy = i / CW, cx = i % CW;
                const int py = ty0 - 1 + cy, px = tx0 - 1 + cx;
                float A = 0.f, Bc = 0.f, Cc = 0.f;
                if (py >= 0 && py < H && px >= 0 && px < W) {
                    Moments m = {0.f, 0.f, 0.f, 0.f, 0.f};
                    {
#pragma clang fp contract(off)   // the forward's (= the reference's) window sums: rounded products added one by one
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx) {
                                const float a = xs[(cy + dy) * IW + cx + dx], bb = ys[(cy + dy) * IW + cx + dx];
                                const float aa = a * a, b2 = bb * bb, ab = a * bb;
                                m.mux += a; m.muy += bb; m.ex2 += aa; m.ey2 += b2; m.exy += ab;
                            }
                        m.mux /= 9.f; m.muy /= 9.f; m.ex2 /= 9.f; m.ey2 /= 9.f; m.exy /= 9.f;
                    }
                    float n, d;
                    const float raw = ssim_from(m, &n, &d);
                    if (raw >= 0.f && raw <= 1.f) {  // clamp passes gradient only inside [0,1]
                        const float gs = gout[(size_t)b * HW + (size_t)py * W + px] * ssim_w / (float)C;
                        const float sx = m.ex2 - m.mux * m.mux, sy = m.ey2 - m.muy * m.muy, sxy = m.exy - m.mux * m.muy;
                        const float A1 = 2.f * m.mux * m.muy + kC1, A2 = 2.f * sxy + kC2;
                        const float B1 = m.mux * m.mux + m.muy * m.muy + kC1, B2 = sx + sy + kC2;
                        const float dn_dmux = 2.f * m.muy * A2 - 2.f * m.muy * A1;
                        const float dd_dmux = 2.f * m.mux * B2 - 2.f * m.mux * B1;
                        A = gs * (-0.5f * (dn_dmux * d - n * dd_dmux) / (d * d));
                        Bc = gs * (0.5f * n * B1 / (d * d));
                        Cc = gs * (-0.5f * (2.f * A1) / d);
                    }
                }
                cA[i] = A; cB[i] = Bc; cC[i] = Cc;
            }
            __syncthreads();
            if (qvalid) {
                const int cy = tid / BT_W + 1, cx = tid % BT_W + 1;
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy) {
                    const int py = qy + dy;
                    if (py < 0 || py >= H) continue;
                    // ReflectionPad2d adjoint: row 1 is also pad row -1 (seen by window row 0), row H-2 also pad row H
                    const float wy = 1.f + ((qy == 1 && py == 0) ? 1.f : 0.f) + ((qy == H - 2 && py == H - 1) ? 1.f : 0.f);
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int px = qx + dx;
                        if (px < 0 || px >= W) continue;
                        const float wx = 1.f + ((qx == 1 && px == 0) ? 1.f : 0.f) + ((qx == W - 2 && px == W - 1) ? 1.f : 0.f);
                        const int o = (cy + dy) * CW + cx + dx;
                        gA += wy * wx * cA[o]; gB += wy * wx * cB[o]; gC += wy * wx * cC[o];
                    }
                }
            }
        }
        if (qvalid) {
            const size_t q = (size_t)qy * W + qx;
            const float xq = xp[q], yq = yp[q];
            const float diff = yq - xq;
            const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
            float g = -sg * gout[(size_t)b * HW + q] * wl1 / (float)C;
            if (use_ssim) g += (gA + 2.f * gB * xq + gC * yq) / 9.f;
            d_pred[((size_t)b * C + c) * HW + q] = g;
        }
    }
}

int check_dims(const char *fn, int B, int C, int H, int W) {
    MD_REQUIRE(B > 0 && B <= 65535 && C > 0 && H >= 3 && W >= 3, "%s: bad dims B=%d C=%d H=%d W=%d (H,W >= 3)", fn, B, C,
               H, W);
    return MD_OK;
}

}  // namespace

extern "C" int md_ssim(const float *x, const float *y, int B, int C, int H, int W, float *out, md_stream_t stream) {
    int rc = check_dims("md_ssim", B, C, H, W);
    if (rc) return rc;
    MD_REQUIRE(x && y && out, "md_ssim: null tensor");
    // channel-agnostic: treat every plane as a one-channel sample
    const long long planes = (long long)B * C;
    MD_REQUIRE(planes <= 65535, "md_ssim: too many planes");
    dim3 grid(md_cdiv(W, 62), md_cdiv(H, 4 * RY), (unsigned)planes);
    hipLaunchKernelGGL((ssim_fwd_kernel<0, 1>), grid, dim3(256), 0, (hipStream_t)stream, x, y, H, W, 0.f, 0, out);
    MD_CHECK_LAUNCH("md_ssim");
    return MD_OK;
}

extern "C" int md_reproj_loss_fwd(const float *pred, const float *target, int B, int C, int H, int W, float ssim_w,
                                  int no_ssim, float *out, md_stream_t stream) {
    int rc = check_dims("md_reproj_loss_fwd", B, C, H, W);
    if (rc) return rc;
    MD_REQUIRE(pred && target && out, "md_reproj_loss_fwd: null tensor");
    MD_REQUIRE(C == 3 || C == 1, "md_reproj_loss_fwd: C=%d unsupported (images are 3-channel; 1 also built)", C);
    dim3 grid(md_cdiv(W, 62), md_cdiv(H, 4 * RY), B);
    if (C == 3)
        hipLaunchKernelGGL((ssim_fwd_kernel<1, 3>), grid, dim3(256), 0, (hipStream_t)stream, pred, target, H, W, ssim_w,
                           no_ssim, out);
    else
        hipLaunchKernelGGL((ssim_fwd_kernel<1, 1>), grid, dim3(256), 0, (hipStream_t)stream, pred, target, H, W, ssim_w,
                           no_ssim, out);
    MD_CHECK_LAUNCH("md_reproj_loss_fwd");
    return MD_OK;
}

extern "C" int md_reproj_loss_bwd(const float *gout, const float *pred, const float *target, int B, int C, int H, int W,
                                  float ssim_w, int no_ssim, float *d_pred, md_stream_t stream) {
    int rc = check_dims("md_reproj_loss_bwd", B, C, H, W);
    if (rc) return rc;
    MD_REQUIRE(gout && pred && target && d_pred, "md_reproj_loss_bwd: null tensor");
    MD_REQUIRE((long long)B * C <= 65535, "md_reproj_loss_bwd: B*C too large");
    dim3 grid(md_cdiv(W, BT_W), md_cdiv(H, BT_H), B * C);
    hipLaunchKernelGGL(reproj_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, gout, pred, target, C, H, W, ssim_w,
                       no_ssim, d_pred);
    MD_CHECK_LAUNCH("md_reproj_loss_bwd");
    return MD_OK;
}
