// Microbenchmark: the cost volume's two output patterns as pure stores (no compute), same grid / loop shape.
// hipcc -O3 --offload-arch=gfx950 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
constexpr int B = 6, D = 96, G = 16, H = 48, W = 160, TW = 32, TH = 8;
// mode 0: planar (B,G,D,h,w), lane = pixel, 16 dword stores per step
// mode 1: ndhwc (B,D,h,w,G), coalesced dwordx4: lane l of store k -> piece k*64+l of the wave's 64px x 64B block
// mode 2: ndhwc, lane = pixel writes its own 64 B as 4 dwordx4 (strided)
// mode 4: ndhwc, two waves share a pixel row piece: each writes 32-byte halves of the 64-byte pixel records (lane l -> pixel l/2, chunk l%2)
// mode 5: ndhwc via 4-lane quads: each store writes whole 64-byte records at a 256-byte stride
// mode 3: planar but each wave-step writes dwordx4 along x (4 px per lane, 16 lanes per row piece) - idealised wide planar
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int dsplit) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = W / TW;
    const int tx0 = (blockIdx.x % tiles_x) * TW, ty0 = (blockIdx.x / tiles_x) * TH;
    const int b = blockIdx.z / dsplit, ds = blockIdx.z % dsplit, dper = D / dsplit;
    const int x = tx0 + tid % TW, y = ty0 + tid / TW;
    const float v = (float)tid;
    for (int d = ds * dper; d < (ds + 1) * dper; ++d) {
        if (MODE == 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) out[(((size_t)b * G + g) * D + d) * H * W + y * W + x] = v + g;
        } else if (MODE == 1) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int L = kk * 64 + lane, pw = L / 4, ch = L % 4;
                const int px = tx0 + pw % TW, py = ty0 + wave * 2 + pw / TW;
                float4 *dst = reinterpret_cast<float4 *>(out + ((((size_t)b * D + d) * H + py) * W + px) * G + ch * 4);
                *dst = make_float4(v, v + 1, v + 2, v + kk);
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float4 *dst = reinterpret_cast<float4 *>(out + ((((size_t)b * D + d) * H + y) * W + x) * G + c * 4);
                *dst = make_float4(v, v + 1, v + 2, v + c);
            }
        } else if (MODE == 4) {
            // 128 px per WG (32x4): waves 0,1 -> half 0, waves 2,3 -> half 1; wave covers 64 px (2 rows)
            const int half = wave >> 1, wrow = (wave & 1) * 2;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int L = kk * 64 + lane, pw = L / 2, ch = L % 2;
                const int px = tx0 + pw % TW, py = (ty0 / 2) + wrow + pw / TW;   // tile height 4 -> ty0/2
                float4 *dst = reinterpret_cast<float4 *>(out + ((((size_t)b * D + d) * H + py) * W + px) * G + half * 8 + ch * 4);
                *dst = make_float4(v, v + 1, v + 2, v + kk);
            }
        } else if (MODE == 5) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int pw = 4 * (lane / 4) + kk, ch = lane % 4;
                const int px = tx0 + pw % TW, py = ty0 + wave * 2 + pw / TW;
                float4 *dst = reinterpret_cast<float4 *>(out + ((((size_t)b * D + d) * H + py) * W + px) * G + ch * 4);
                *dst = make_float4(v, v + 1, v + 2, v + kk);
            }
        } else {
            // 16 groups x 64 px per wave-step = 1024 floats = 256 float4: lane handles 4 float4: (g = kk*4 + lane/16, 4 px)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int g = kk * 4 + lane / 16, q = lane % 16;  // q: float4 index within the wave's 64 px (2 rows x 8)
                const int py = ty0 + wave * 2 + q / 8, px = tx0 + (q % 8) * 4;
                float4 *dst = reinterpret_cast<float4 *>(out + (((size_t)b * G + g) * D + d) * H * W + py * W + px);
                *dst = make_float4(v, v + 1, v + 2, v + kk);
            }
        }
    }
}
template <int MODE>
float run(float *out, int dsplit, int iters) {
    dim3 grid((W / TW) * (H / TH) * (MODE == 4 ? 2 : 1), 1, B * dsplit);
    hipEvent_t a, b_;
    hipEventCreate(&a); hipEventCreate(&b_);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, out, dsplit);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, out, dsplit);
    hipEventRecord(b_);
    hipEventSynchronize(b_);
    float ms; hipEventElapsedTime(&ms, a, b_);
    return ms * 1e3f / iters;
}
int main() {
    float *out; size_t n = (size_t)B * D * G * H * W;
    hipMalloc(&out, n * 4);
    for (int ds : {1, 2, 3, 4, 6, 12}) {
        float t0 = run<0>(out, ds, 30), t1 = run<1>(out, ds, 30), t2 = run<4>(out, ds, 30), t3 = run<5>(out, ds, 30);
        printf("dsplit %2d (%4d WGs): planar-dword %.1f us (%.0f GB/s) | ndhwc-coalesced %.1f us (%.0f GB/s) | ndhwc-halfrecords %.1f us (%.0f GB/s) | ndhwc-quadrecords %.1f us (%.0f GB/s)\n",
               ds, 30 * 6 * ds, t0, n * 4 / t0 / 1e3, t1, n * 4 / t1 / 1e3, t2, n * 4 / t2 / 1e3, t3, n * 4 / t3 / 1e3);
    }
    return 0;
}
