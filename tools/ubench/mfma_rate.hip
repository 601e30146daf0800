// Issue rate of v_mfma_f32_16x16x32_bf16 on gfx950 in isolation: NACC independent accumulators, back to back, one wave per SIMD or several.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_rate mfma_rate.hip ; run on the GPU box.  Prints cycles per instruction per wave
// (s_memtime) and the chip-wide rate from the wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC, bool LDSFEED>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, int iters) {
    __shared__ bf16x8 lds[256 * 2];
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    lds[threadIdx.x] = a; lds[256 + threadIdx.x] = b;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (LDSFEED) { a = lds[(threadIdx.x + it) & 255]; b = lds[256 + ((threadIdx.x + 2 * it) & 255)]; }
        // inline assembly with destination == source C: the builtin form let the register allocator rotate the accumulators through
        // each other (a[8:11] <- a[6:9] ...), which chains "independent" instructions
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, bool LDSFEED>
void run(int wgs_per_cu, const char *tag) {
    const int iters = 4096, grid = 256 * wgs_per_cu;
    float *out; long long *cyc;
    hipMalloc(&out, sizeof(float) * grid * 256); hipMalloc(&cyc, sizeof(long long) * grid);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, LDSFEED><<<grid, 256>>>(out, cyc, 16);
    hipEventRecord(e0);
    k<NACC, LDSFEED><<<grid, 256>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid); hipMemcpy(h.data(), cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= grid;
    const double n_inst = (double)iters * NACC;                       // per wave
    const double flops = n_inst * 16.0 * 16 * 32 * 2 * 4 * grid;      // 4 waves per workgroup
    printf("%-34s acc %2d  waves/SIMD %d: %6.1f shader cycles per MFMA per wave (s_memtime)%.0s; wall %7.3f ms = %7.1f TFLOP/s chip-wide, %5.2f ns per MFMA per SIMD\n",
           tag, NACC, wgs_per_cu, mean / n_inst, "", ms, flops / ms / 1e9, ms * 1e6 / (n_inst * wgs_per_cu));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1, false>(1, "dependent chain");
    run<1, false>(2, "dependent chain");
    run<1, false>(4, "dependent chain");
    run<3, false>(1, "3 independent");
    run<3, false>(2, "3 independent");
    run<6, false>(1, "6 independent");
    run<6, false>(2, "6 independent");
    run<4, false>(2, "4 independent");
    run<4, false>(3, "4 independent");
    run<2, false>(1, "2 independent");
    run<4, false>(1, "4 independent");
    run<8, false>(1, "8 independent");
    run<8, false>(2, "8 independent");
    run<16, false>(1, "16 independent");
    run<16, false>(2, "16 independent");
    run<8, true>(1, "8 independent, operands from LDS");
    run<8, true>(2, "8 independent, operands from LDS");
    run<27, false>(1, "27 independent (the wgrad's count)");
    return 0;
}
