// Probe: v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products per wave instruction) -- operand / result layout and issue
// cost, for the plane sweep's per-step sum out_j(d) = sum_t w_t(d) * C_t[j] (rows = 4 steps of a pixel's lane quad, columns = the
// quad's 4 lanes).   hipcc -O3 --offload-arch=gfx950 mfma4x4_probe.hip -o mfma4x4_probe && ./mfma4x4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void layout(const float *a, const float *b, float *d) {
    const int l = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

// MODE 0: 16 MFMAs (4 accumulators x 4 taps) per batch; 1: the same sums as 32 v_pk_fma_f32 (8 per step) + 16 DPP broadcasts
template <int S>
__device__ __forceinline__ float bc(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), S * 0x55, 0xF, 0xF, true)); }
template <int S>
__device__ __forceinline__ void step(const float (&w)[4], const float (&c)[16], f4 &tot) {
    float ws[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) ws[t] = bc<S>(w[t]);
    f2 o0 = {0.f, 0.f}, o1 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        o0 = __builtin_elementwise_fma((f2){c[t], c[4 + t]}, (f2){ws[t], ws[t]}, o0);
        o1 = __builtin_elementwise_fma((f2){c[8 + t], c[12 + t]}, (f2){ws[t], ws[t]}, o1);
    }
    tot[0] += o0[0]; tot[1] += o0[1]; tot[2] += o1[0]; tot[3] += o1[1];
}
template <int MODE>
__global__ __launch_bounds__(256) void rate(float *out, int iters) {
    const int l = threadIdx.x;
    float w[4], c[16];
    for (int i = 0; i < 4; ++i) w[i] = 0.25f + 1e-3f * (l + i);
    for (int i = 0; i < 16; ++i) c[i] = 1.f + 1e-3f * (l * 16 + i);
    f4 tot = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            f4 acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[0], c[q * 4 + 0], (f4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int t = 1; t < 4; ++t) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[t], c[q * 4 + t], acc[q], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) tot += acc[q];
        } else {
            step<0>(w, c, tot); step<1>(w, c, tot); step<2>(w, c, tot); step<3>(w, c, tot);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] += 1e-6f;  // keep the loop from being hoisted
    }
    out[blockIdx.x * 256 + l] = tot[0] + tot[1] + tot[2] + tot[3];
}

int main() {
    float ha[64], hb[64], hd[256], *a, *b, *d;
    for (int i = 0; i < 64; ++i) { ha[i] = 1 + i; hb[i] = 100 * (1 + i); }
    (void)hipMalloc(&a, 256); (void)hipMalloc(&b, 256); (void)hipMalloc(&d, 1024);
    (void)hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    layout<<<1, 64>>>(a, b, d);
    (void)hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    // hypothesis: lane l = (block l/4, column l%4); register r = row r: D = A[block*4 + r] * B[l]
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hd[l * 4 + r] != ha[(l / 4) * 4 + r] * hb[l]) ok = 0;
    printf("layout D[lane l][reg r] == A[4*(l/4)+r] * B[l]: %s\n", ok ? "yes" : "NO");
    if (!ok) for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3]);
    float *out; (void)hipMalloc(&out, 1024 * 256 * 4 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) for (int wg = 1; wg <= 4; wg *= 2) {
        const int iters = 20000, grid = 256 * wg;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) rate<0><<<grid, 256>>>(out, iters); else rate<1><<<grid, 256>>>(out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: wg workgroups/CU x 4 waves / 4 SIMDs = wg waves; cycles per batch per wave at 2.4 GHz
        printf("%s, %d waves per SIMD: %.1f cycles per batch of 4 steps per wave (all waves of a SIMD together: %.1f)\n", mode == 0 ? "16 x mfma_4x4x1" : "32 x pk_fma + 16 dpp",
               wg, ms * 1e-3 * 2.4e9 / iters, ms * 1e-3 * 2.4e9 / iters / wg);
    }
    return 0;
}
