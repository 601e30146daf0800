// Microbenchmark 2: channels-last (B,D,h,w,16) volume written by 128-pixel workgroups (32x4 tile, 256 threads) whose
// wave pairs each own one 32-byte half of every pixel's 64-byte record ("two threads per pixel" decomposition).
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int B = 6, D = 96, G = 16, H = 48, W = 160, TW = 32, TH = 4;
// MODE 0: each wave stores its own 32-byte halves (lane l -> pixel l/2, 16-byte chunk l%2), 2 stores per step
// MODE 1: reference: full 64-byte records, 256-pixel workgroups (the current kernel's pattern)
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int dsplit) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z / dsplit, ds = blockIdx.z % dsplit, dper = D / dsplit;
    const float v = (float)tid;
    if (MODE == 0) {
        const int tiles_x = W / TW;
        const int tx0 = (blockIdx.x % tiles_x) * TW, ty0 = (blockIdx.x / tiles_x) * TH;
        const int half = wave >> 1, wrow = (wave & 1) * 2;
        for (int d = ds * dper; d < (ds + 1) * dper; ++d)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int L = kk * 64 + lane, pw = L / 2, ch = L % 2;
                const int px = tx0 + pw % TW, py = ty0 + wrow + pw / TW;
                *reinterpret_cast<float4 *>(out + ((((size_t)b * D + d) * H + py) * W + px) * G + half * 8 + ch * 4) = make_float4(v, v + 1, v + 2, v + kk);
            }
    } else {
        const int tiles_x = W / TW;
        const int tx0 = (blockIdx.x % tiles_x) * TW, ty0 = (blockIdx.x / tiles_x) * 8;
        for (int d = ds * dper; d < (ds + 1) * dper; ++d)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int L = kk * 64 + lane, pw = L / 4, ch = L % 4;
                const int px = tx0 + pw % TW, py = ty0 + wave * 2 + pw / TW;
                *reinterpret_cast<float4 *>(out + ((((size_t)b * D + d) * H + py) * W + px) * G + ch * 4) = make_float4(v, v + 1, v + 2, v + kk);
            }
    }
}
template <int MODE>
float run(float *out, int dsplit, int iters) {
    dim3 grid((W / TW) * (H / (MODE == 0 ? TH : 8)), 1, B * dsplit);
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
    for (int rep = 0; rep < 2; ++rep)
        for (int ds : {1, 2, 3, 4, 6}) {
            float t0 = run<0>(out, ds, 30), t1 = run<1>(out, ds, 30);
            printf("dsplit %d: half-records by wave pairs (%4d WGs of 128 px) %.1f us (%.0f GB/s) | full records (%4d WGs of 256 px) %.1f us (%.0f GB/s)\n",
                   ds, 60 * 6 * ds, t0, n * 4 / t0 / 1e3, 30 * 6 * ds, t1, n * 4 / t1 / 1e3);
        }
    return 0;
}
