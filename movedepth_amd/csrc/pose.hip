// transformation_from_parameters (reference layers.py:412-429, with rot_from_axisangle :479-518 and
// get_translation_matrix :464-477): 6 numbers per sample -> the 4x4 pose the warp and plane-sweep kernels consume,
// forward and backward in one launch each.  As torch ops this is ~75 launches forward and as many backward per call
// (two calls per training step), each a few microseconds for 6 samples.
//   angle = |v|, axis = v / (angle + 1e-7), R = Rodrigues(axis, angle)   (the reference's +1e-7 is kept)
//   invert:  M = R^T . T(-t)        else:  M = T(t) . R
#include "md_common.hpp"

namespace {

struct Rod {
    float x, y, z, ca, sa, C, angle, s;
    float R[9];
};

__device__ __forceinline__ Rod rodrigues(const float *v) {
    Rod r;
    r.angle = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    r.s = r.angle + 1e-7f;
    r.x = v[0] / r.s; r.y = v[1] / r.s; r.z = v[2] / r.s;
    r.ca = cosf(r.angle); r.sa = sinf(r.angle); r.C = 1.f - r.ca;
    const float xs = r.x * r.sa, ys = r.y * r.sa, zs = r.z * r.sa;
    const float xC = r.x * r.C, yC = r.y * r.C, zC = r.z * r.C;
    const float xyC = r.x * yC, yzC = r.y * zC, zxC = r.z * xC;  // same products, same order as the reference
    r.R[0] = r.x * xC + r.ca; r.R[1] = xyC - zs;          r.R[2] = zxC + ys;
    r.R[3] = xyC + zs;        r.R[4] = r.y * yC + r.ca;   r.R[5] = yzC - xs;
    r.R[6] = zxC - ys;        r.R[7] = yzC + xs;          r.R[8] = r.z * zC + r.ca;
    return r;
}

__global__ void pose_matrix_fwd_kernel(const float *__restrict__ aa, const float *__restrict__ tr, int B, int invert,
                                       float *__restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const Rod r = rodrigues(aa + b * 3);
    const float t0 = tr[b * 3], t1 = tr[b * 3 + 1], t2 = tr[b * 3 + 2];
    float *M = out + b * 16;
    if (invert) {
        // R^T . T(-t): rotation block R^T, last column sum_j R^T[i][j] * (-t_j)
        const float n0 = -t0, n1 = -t1, n2 = -t2;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float a = r.R[0 * 3 + i], bb = r.R[1 * 3 + i], c = r.R[2 * 3 + i];  // row i of R^T
            M[i * 4 + 0] = a; M[i * 4 + 1] = bb; M[i * 4 + 2] = c;
            M[i * 4 + 3] = a * n0 + bb * n1 + c * n2;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            M[i * 4 + 0] = r.R[i * 3]; M[i * 4 + 1] = r.R[i * 3 + 1]; M[i * 4 + 2] = r.R[i * 3 + 2];
        }
        M[3] = t0; M[7] = t1; M[11] = t2;
    }
    M[12] = 0.f; M[13] = 0.f; M[14] = 0.f; M[15] = 1.f;
}

__global__ void pose_matrix_bwd_kernel(const float *__restrict__ gM, const float *__restrict__ aa,
                                       const float *__restrict__ tr, int B, int invert, float *__restrict__ d_aa,
                                       float *__restrict__ d_tr) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *v = aa + b * 3, *g = gM + b * 16;
    const Rod r = rodrigues(v);
    float G[9];  // dL/dR
    if (invert) {
        const float t[3] = {tr[b * 3], tr[b * 3 + 1], tr[b * 3 + 2]};
        // M[i][j] = R[j][i]; M[i][3] = -sum_j R[j][i] t_j
        float gt[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                G[j * 3 + i] = g[i * 4 + j] - g[i * 4 + 3] * t[j];
                gt[j] -= g[i * 4 + 3] * r.R[j * 3 + i];
            }
        d_tr[b * 3] = gt[0]; d_tr[b * 3 + 1] = gt[1]; d_tr[b * 3 + 2] = gt[2];
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) G[i * 3 + j] = g[i * 4 + j];
        d_tr[b * 3] = g[3]; d_tr[b * 3 + 1] = g[7]; d_tr[b * 3 + 2] = g[11];
    }
    const float x = r.x, y = r.y, z = r.z, C = r.C, sa = r.sa, ca = r.ca;
    const float s01 = G[1] + G[3], s02 = G[2] + G[6], s12 = G[5] + G[7];
    const float gx = G[0] * 2.f * x * C + s01 * y * C + s02 * z * C + (G[7] - G[5]) * sa;
    const float gy = G[4] * 2.f * y * C + s01 * x * C + s12 * z * C + (G[2] - G[6]) * sa;
    const float gz = G[8] * 2.f * z * C + s02 * x * C + s12 * y * C + (G[3] - G[1]) * sa;
    const float gC = G[0] * x * x + G[4] * y * y + G[8] * z * z + s01 * x * y + s02 * z * x + s12 * y * z;
    const float gca = G[0] + G[4] + G[8] - gC;
    const float gsa = (G[3] - G[1]) * z + (G[2] - G[6]) * y + (G[7] - G[5]) * x;
    const float gth = -gca * sa + gsa * ca;  // through cos / sin of the angle
    // axis = v / s, s = angle + 1e-7; d angle / d v = v / angle (0 at the origin, as torch's norm backward)
    const float dotgv = gx * v[0] + gy * v[1] + gz * v[2];
    const float k = r.angle > 0.f ? (gth - dotgv / (r.s * r.s)) / r.angle : 0.f;
    d_aa[b * 3] = gx / r.s + k * v[0];
    d_aa[b * 3 + 1] = gy / r.s + k * v[1];
    d_aa[b * 3 + 2] = gz / r.s + k * v[2];
}

}  // namespace

extern "C" {

int md_pose_matrix_fwd(const float *axisangle, const float *translation, int B, int invert, float *T, md_stream_t stream) {
    MD_REQUIRE(axisangle && translation && T, "md_pose_matrix_fwd: null tensor argument");
    MD_REQUIRE(B > 0, "md_pose_matrix_fwd: bad batch size %d", B);
    hipLaunchKernelGGL(pose_matrix_fwd_kernel, dim3(md_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, axisangle, translation, B,
                       invert, T);
    MD_CHECK_LAUNCH("md_pose_matrix_fwd");
    return MD_OK;
}

int md_pose_matrix_bwd(const float *gT, const float *axisangle, const float *translation, int B, int invert,
                       float *d_axisangle, float *d_translation, md_stream_t stream) {
    MD_REQUIRE(gT && axisangle && translation && d_axisangle && d_translation, "md_pose_matrix_bwd: null tensor argument");
    MD_REQUIRE(B > 0, "md_pose_matrix_bwd: bad batch size %d", B);
    hipLaunchKernelGGL(pose_matrix_bwd_kernel, dim3(md_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, gT, axisangle, translation,
                       B, invert, d_axisangle, d_translation);
    MD_CHECK_LAUNCH("md_pose_matrix_bwd");
    return MD_OK;
}

}  // extern "C"
