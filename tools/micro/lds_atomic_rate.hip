// LDS atomic throughput on gfx950: float vs integer, 32- vs 64-bit, by number of active lanes and address pattern.
//   hipcc -O3 --offload-arch=gfx950 lds_atomic_rate.hip -o lds_atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int ITERS = 256, NPER = 16;

// KIND 0: ds_add_f32, 1: ds_add_u32, 2: ds_add_u64, 3: ds_add_f64, 4: plain read-add-write (no atomic), 5: ds_max_f32
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int active_mod, int stride, long long *cyc) {
    __shared__ double buf[4096];  // 32 KB
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) buf[i] = 0.0;
    __syncthreads();
    const bool active = (lane % active_mod) == 0;
    float *f = reinterpret_cast<float *>(buf);
    unsigned *u = reinterpret_cast<unsigned *>(buf);
    unsigned long long *u64 = reinterpret_cast<unsigned long long *>(buf);
    const int base = (tid * stride) & 2047;
    const long long t0 = clock64();
    if (active) {
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int j = 0; j < NPER; ++j) {
                const int idx = (base + j * 256 + it) & 4095;
                if (KIND == 0) atomicAdd(f + idx, 1.0f);
                else if (KIND == 1) atomicAdd(u + idx, 1u);
                else if (KIND == 2) atomicAdd(u64 + (idx & 2047), 1ull);
                else if (KIND == 3) atomicAdd(buf + (idx & 2047), 1.0);
                else if (KIND == 4) f[idx] += 1.0f;
                else atomicMax(reinterpret_cast<int *>(u) + idx, it);
            }
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    if (tid < 64) out[blockIdx.x * 64 + tid] = f[tid] + (float)buf[tid + 64];
}

template <int KIND>
void run(const char *name, float *out, long long *cyc) {
    for (int wgs_per_cu : {1, 2})
        for (int active_mod : {1, 4, 16})
            for (int stride : {1, 33}) {
                hipEvent_t a, b;
                hipEventCreate(&a); hipEventCreate(&b);
                const int grid = 256 * wgs_per_cu;
                hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, active_mod, stride, cyc);
                hipEventRecord(a);
                hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, active_mod, stride, cyc);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
                const double wave_instr = 4.0 * wgs_per_cu * ITERS * NPER;      // per CU
                const double lane_ops = wave_instr * (64 / active_mod);
                printf("%-12s %d WG/CU active 1/%-2d stride %2d: %7.1f us  %6.1f cyc per wave-instr per CU  %5.2f lane-ops/clk/CU\n", name,
                       wgs_per_cu, active_mod, stride, ms * 1e3, (double)c / (wave_instr / wgs_per_cu / 1.0) * 1.0, lane_ops / (double)c / 1.0);
            }
}
int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 512 * 64 * 4); hipMalloc(&cyc, 512 * 8);
    run<0>("ds_add_f32", out, cyc);
    run<1>("ds_add_u32", out, cyc);
    run<2>("ds_add_u64", out, cyc);
    run<3>("ds_add_f64", out, cyc);
    run<4>("plain rmw", out, cyc);
    run<5>("ds_max_i32", out, cyc);
    return 0;
}
