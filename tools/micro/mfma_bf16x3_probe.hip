// Probe for DESIGN 8 item 4: issue cost of the fp32 MFMA the 16->16 convolution kernels use against the bf16 MFMA a three-piece
// split (bf16x3) would use, per K = 32 slice of a 16 x 16 output tile: 8 x v_mfma_f32_16x16x4_f32 against 6 x
// v_mfma_f32_16x16x32_bf16 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid), plus the accuracy of the split on random data.
//   hipcc -O3 --offload-arch=gfx950 mfma_bf16x3_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
static float bf_round(float v) { unsigned u; memcpy(&u, &v, 4); u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u; float r; memcpy(&r, &u, 4); return r; }
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void rate(float *out, int iters) {
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const float a = 1.f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 2e-3f;
    bf8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(a + i); bh[i] = (__bf16)(b - i); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + k, b - k, acc[k & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[k & 3], 0, 0, 0);
        }
    }
    f4 t = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * 256 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
}

int main() {
    float *out;
    (void)hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) for (int wg = 1; wg <= 2; ++wg) {
        const int iters = 20000, grid = 256 * wg;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) rate<0><<<grid, 256>>>(out, iters); else rate<1><<<grid, 256>>>(out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s, %d waves per SIMD: %.1f cycles per K = 32 slice per SIMD (at 2.4 GHz)\n", mode == 0 ? "8 x mfma_f32_16x16x4_f32      " : "6 x mfma_f32_16x16x32_bf16 (x3)",
               wg, ms * 1e-3 * 2.4e9 / iters / wg);
    }
    // accuracy of the split: x = hi + mid + lo in bf16 pieces (round to nearest each), products hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid
    double worst = 0, sum = 0;
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) * (1.0 / 16777216.0) * 2.0 - 1.0); };
    auto bf = [](float v) { return bf_round(v); };
    for (int n = 0; n < 100000; ++n) {
        const float x = rnd(), y = rnd();
        const float xh = bf(x), xm = bf(x - xh), xl = bf(x - xh - xm), yh = bf(y), ym = bf(y - yh), yl = bf(y - yh - ym);
        const double got = (double)xh * yh + ((double)xh * ym + (double)xm * yh) + ((double)xh * yl + (double)xl * yh + (double)xm * ym);
        const double err = fabs(got - (double)x * y) / fmax(fabs((double)x * y), 1e-30);
        if (fabs(x * y) > 1e-3) { worst = fmax(worst, err); sum += err; }
    }
    printf("bf16x3 product: worst relative error %.2e (fp32 product rounding: 6e-8)\n", worst);
    return 0;
}
