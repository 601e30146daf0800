// Probe: what bounds the channels-last plane-sweep forward -- its store pattern, its occupancy, or the overlap of its
// per-step work with its stores?  Same launch geometry as costvol_fwd_nhwc_kernel (256-thread workgroups, 32x8 pixel tile,
// a linear split of items x D steps over the grid, each wave transposing 64 px x 16 groups through LDS and writing 4 x 1 KB),
// cold output (8 x 283 MB blocks in turn).  Knobs: volume layout, workgroups per CU (dynamic LDS padding), grid size,
// dummy VALU work per step, dummy LDS reads per step.
//   hipcc -O3 --offload-arch=gfx950 cv_store_probe.hip -o cv_store_probe && ./cv_store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
constexpr int B = 6, D = 96, G = 16, H = 48, W = 160, TW = 32, TH = 8;
constexpr int TILES_X = W / TW, TILES = TILES_X * (H / TH), ITEMS = B * TILES;

// LAYOUT 0: (B,D,h,w,G) channels-last;  1: tile-blocked (B,tile,D,256 px,G): a workgroup's steps are one contiguous stream
template <int LAYOUT, bool NOLDS = false, int WAITN = -1>
__global__ __launch_bounds__(256) void probe(float *__restrict__ out, const float *__restrict__ src, int nfma, int nlds, int kslices = 0) {
    extern __shared__ float4 smem[];  // [0, 1024): transpose tiles (4 waves x 64 px x 4 chunks); the rest: dummy window
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float4 *my_stage = smem + wave * 256;
    float4 *win = smem + 1024;
    for (int i = tid; i < 2048; i += 256) win[i] = make_float4(src[i], src[i + 1], src[i + 2], src[i + 3]);
    __syncthreads();
    const long long total = (long long)ITEMS * D;
    long long lo = total * blockIdx.x / gridDim.x;
    long long hi = total * (blockIdx.x + 1) / gridDim.x;
    if (kslices > 0) {
        // item-aligned: workgroup = (slice, item), consecutive workgroups = consecutive tiles at the same d range, so the
        // whole grid sweeps d in lockstep (kslices moving windows of 6 planes each)
        const int item = blockIdx.x % ITEMS, sl = blockIdx.x / ITEMS;
        lo = (long long)item * D + (long long)D * sl / kslices;
        hi = (long long)item * D + (long long)D * (sl + 1) / kslices;
    }
    float og[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) og[j] = kslices < 0 ? 1.f : (float)(tid + j);
    const float a = src[tid & 15], c = src[16 + (tid & 15)];
    while (lo < hi) {
        const int item = (int)(lo / D), d0 = (int)(lo % D);
        const int d1 = (hi - lo) + d0 < D ? d0 + (int)(hi - lo) : D;
        const int b = item / TILES, tile = item % TILES;
        const int tx0 = (tile % TILES_X) * TW, ty0 = (tile / TILES_X) * TH + wave * 2;
        long long soff[4];
        int sidx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int L = k * 64 + lane, pw = L / 4, ch = L % 4;
            sidx[k] = pw * 4 + (ch ^ ((pw >> 1) & 3));
            if (LAYOUT == 0) {
                const int px = tx0 + pw % TW, py = ty0 + pw / TW;
                soff[k] = (((long long)b * D + d0) * H * W + (long long)py * W + px) * G + ch * 4;
            } else {
                soff[k] = (((long long)item * D + d0) * 256 + wave * 64 + pw) * G + ch * 4;
            }
        }
        const long long sd = LAYOUT == 0 ? (long long)H * W * G : 256LL * G;
        for (int d = d0; d < d1; ++d) {
            // dummy per-step work: nlds ds_read_b128 from a lane-dependent window position, nfma FMAs over 16 accumulators
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int base = (lane * 5 + d) & 1023;
            for (int i = 0; i < nlds; ++i) {
                const float4 v = win[(base + i * 37) & 2047];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            for (int i = 0; i < nfma; i += 16) {
#pragma unroll
                for (int j = 0; j < 16; ++j) og[j] = fmaf(og[j], a, c);
            }
            og[0] += acc.x; og[1] += acc.y; og[2] += acc.z; og[3] += acc.w;
            if (NOLDS) {
                // same addresses, values straight from registers (wrong values: this only prices the LDS transpose)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<float4 *>(out + soff[k] + (long long)(d - d0) * sd) = make_float4(og[4 * k], og[4 * k + 1], og[4 * k + 2], og[4 * k + 3]);
            } else {
#pragma unroll
            for (int cidx = 0; cidx < 4; ++cidx)
                my_stage[lane * 4 + (cidx ^ ((lane >> 1) & 3))] = make_float4(og[4 * cidx], og[4 * cidx + 1], og[4 * cidx + 2], og[4 * cidx + 3]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 v = my_stage[sidx[k]];
                *reinterpret_cast<float4 *>(out + soff[k] + (long long)(d - d0) * sd) = v;
            }
            __builtin_amdgcn_wave_barrier();
            }
            // flow control: let at most WAITN of this wave's stores stay in flight before it goes on
            if (WAITN == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (WAITN == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if (WAITN == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (WAITN == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (WAITN == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
        lo += d1 - d0;
    }
}

// every block writes one contiguous chunk of `per` float4 per thread-row (what an elementwise library kernel does)
__global__ __launch_bounds__(256) void fill_chunked(float4 *out, size_t n4, int per, int varied = 0) {
    size_t base = (size_t)blockIdx.x * 256 * per + threadIdx.x;
    float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    if (varied) {  // pseudo-random bit patterns per element (hash of the index), as real data would be
        unsigned h = (unsigned)base * 2654435761u;
        v = make_float4(__uint_as_float((h & 0x007fffffu) | 0x3f000000u), __uint_as_float(((h * 31u) & 0x007fffffu) | 0x3f000000u),
                        __uint_as_float(((h * 131u) & 0x007fffffu) | 0x3f000000u), __uint_as_float(((h * 1031u) & 0x007fffffu) | 0x3f000000u));
    }
    for (int i = 0; i < per; ++i)
        if (base + (size_t)i * 256 < n4) out[base + (size_t)i * 256] = v;
}

__global__ void fill_linear(float4 *out, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    const size_t n = (size_t)B * D * G * H * W;
    constexpr int NBUF = 8;
    std::vector<float *> bufs(NBUF);
    for (auto &p : bufs) hipMalloc(&p, n * 4);
    float *src;
    hipMalloc(&src, 16384);
    hipMemset(src, 0, 16384);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](auto launch, int iters) {
        for (int i = 0; i < 4; ++i) launch(bufs[i % NBUF]);
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) launch(bufs[i % NBUF]);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        return ms * 1e3f / iters;
    };
    {
        float t = timeit([&](float *o) { hipLaunchKernelGGL(fill_linear, dim3(2048), dim3(256), 0, 0, (float4 *)o, n / 4); }, 40);
        printf("linear fill (2048 x 256 threads, grid-stride float4): %.1f us  %.0f GB/s\n", t, n * 4 / t / 1e3);
    }
    for (int per : {1, 2, 4, 8, 16, 64}) {
        const size_t n4 = n / 4;
        const unsigned grid = (unsigned)((n4 + 256ull * per - 1) / (256ull * per));
        float t = timeit([&](float *o) { hipLaunchKernelGGL(fill_chunked, dim3(grid), dim3(256), 0, 0, (float4 *)o, n4, per); }, 40);
        printf("chunked fill (%u blocks, %d float4 per thread): %.1f us  %.0f GB/s\n", grid, per, t, n * 4 / t / 1e3);
    }
    for (int layout = 0; layout < 2; ++layout)
        for (int oi : {0, 2, 4}) {
            const int lds_kb_[] = {80, 53, 40, 26, 20}, occ_[] = {2, 3, 4, 6, 8};
            const int lds = lds_kb_[oi] * 1024, nwg = 256 * occ_[oi];
            if (layout == 0) hipFuncSetAttribute((const void *)probe<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            else hipFuncSetAttribute((const void *)probe<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            auto launch = [&](float *o) {
                if (layout == 0) hipLaunchKernelGGL((probe<0, true>), dim3(nwg), dim3(256), lds, 0, o, src, 0, 0);
                else hipLaunchKernelGGL((probe<1, true>), dim3(nwg), dim3(256), lds, 0, o, src, 0, 0);
            };
            float t = timeit(launch, 24);
            printf("NO LDS TRANSPOSE layout %d, %d WG/CU: %.1f us  %.0f GB/s\n", layout, occ_[oi], t, n * 4 / t / 1e3);
        }
    for (int varied = 0; varied < 2; ++varied)
        for (int per : {1, 4}) {
            const size_t n4 = n / 4;
            const unsigned grid = (unsigned)((n4 + 256ull * per - 1) / (256ull * per));
            float t = timeit([&](float *o) { hipLaunchKernelGGL(fill_chunked, dim3(grid), dim3(256), 0, 0, (float4 *)o, n4, per, varied); }, 40);
            printf("DATA chunked fill per %d, %s data: %.1f us  %.0f GB/s\n", per, varied ? "hashed" : "constant", t, n * 4 / t / 1e3);
        }
    {
        // the probe with all-equal data: src is zero, nfma=0 keeps og = tid + j; use kslices=-1 as "constant data" switch
        const int lds = 80 * 1024, nwg = 512;
        hipFuncSetAttribute((const void *)probe<0, false, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        auto launch = [&](float *o) { hipLaunchKernelGGL((probe<0, false, -1>), dim3(nwg), dim3(256), lds, 0, o, src, 0, 0, -1); };
        float t = timeit(launch, 24);
        printf("DATA probe stores only, constant data: %.1f us  %.0f GB/s\n", t, n * 4 / t / 1e3);
        auto launch2 = [&](float *o) { hipLaunchKernelGGL((probe<0, false, -1>), dim3(nwg), dim3(256), lds, 0, o, src, 0, 0, 0); };
        t = timeit(launch2, 24);
        printf("DATA probe stores only, per-lane data: %.1f us  %.0f GB/s\n", t, n * 4 / t / 1e3);
        return 0;
    }
    {
        const int lds_kb_[] = {80, 53, 40, 26, 20}, occ_[] = {2, 3, 4, 6, 8};
        for (int occi : {0, 2, 4}) {
            const int lds = lds_kb_[occi] * 1024, nwg = 256 * occ_[occi];
            for (int work = 0; work < 3; ++work) {
                const int nf = work == 0 ? 0 : (work == 1 ? 128 : 256), nl = work ? 32 : 0;
#define RUNW(WN)                                                                                                              \
                {                                                                                                             \
                    hipFuncSetAttribute((const void *)probe<0, false, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);  \
                    auto launch = [&](float *o) { hipLaunchKernelGGL((probe<0, false, WN>), dim3(nwg), dim3(256), lds, 0, o, src, nf, nl, 0); }; \
                    float t = timeit(launch, 24);                                                                             \
                    printf("FLOWCTL vmcnt(%2d) %d WG/CU nfma %3d nlds %2d: %.1f us  %.0f GB/s\n", WN, occ_[occi], nf, nl, t, n * 4 / t / 1e3); \
                }
                RUNW(-1) RUNW(0) RUNW(2) RUNW(4) RUNW(8) RUNW(16)
            }
        }
        return 0;
    }
    for (int occi : {0, 4})
        for (int ks : {2, 3, 4, 6, 8, 12, 24}) {
            const int lds_kb_[] = {80, 53, 40, 26, 20}, occ_[] = {2, 3, 4, 6, 8};
            const int lds = lds_kb_[occi] * 1024, nwg = ITEMS * ks;
            hipFuncSetAttribute((const void *)probe<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            for (int work = 0; work < 2; ++work) {
                auto launch = [&](float *o) { hipLaunchKernelGGL((probe<0, false>), dim3(nwg), dim3(256), lds, 0, o, src, work ? 256 : 0, work ? 32 : 0, ks); };
                float t = timeit(launch, 24);
                printf("ITEM-ALIGNED lockstep: %2d slices (%4d WGs, %d WG/CU by LDS) %s: %.1f us  %.0f GB/s\n", ks, nwg, occ_[occi],
                       work ? "nfma 256 nlds 32" : "stores only     ", t, n * 4 / t / 1e3);
            }
        }
    return 0;
    // LDS per workgroup -> workgroups per CU: 80 KB -> 2, 53 KB -> 3, 40 KB -> 4, 26 KB -> 6, 20 KB -> 8
    const int lds_kb[] = {80, 53, 40, 26, 20};
    const int occ[] = {2, 3, 4, 6, 8};
    for (int layout = 0; layout < 2; ++layout)
        for (int oi : {0, 2})
            for (int nfma : {0, 256})
                for (int nlds : {0, 32}) {
                    
                    const int lds = lds_kb[oi] * 1024, nwg = 256 * occ[oi];
                    auto launch = [&](float *o) {
                        if (layout == 0) hipLaunchKernelGGL(probe<0>, dim3(nwg), dim3(256), lds, 0, o, src, nfma, nlds);
                        else hipLaunchKernelGGL(probe<1>, dim3(nwg), dim3(256), lds, 0, o, src, nfma, nlds);
                    };
                    if (layout == 0) hipFuncSetAttribute((const void *)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    else hipFuncSetAttribute((const void *)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    float t = timeit(launch, 24);
                    printf("layout %s  %d WG/CU (%4d WGs, %2d KB LDS)  nfma %3d  nlds %2d : %6.1f us  %.0f GB/s (%.1f%% of 8 TB/s)\n",
                           layout ? "tile-blocked" : "ndhwc       ", occ[oi], nwg, lds_kb[oi], nfma, nlds, t, n * 4 / t / 1e3,
                           n * 4 / t / 1e3 / 80);
                }
    // the real grid (512 WGs at 2 per CU) but over-subscribed grids at the same LDS: does a longer queue of workgroups help?
    for (int nwg : {512, 1024, 2048, 4096})
        for (int layout = 0; layout < 2; ++layout) {
            const int lds = 80 * 1024;
            auto launch = [&](float *o) {
                if (layout == 0) hipLaunchKernelGGL(probe<0>, dim3(nwg), dim3(256), lds, 0, o, src, 256, 32);
                else hipLaunchKernelGGL(probe<1>, dim3(nwg), dim3(256), lds, 0, o, src, 256, 32);
            };
            float t = timeit(launch, 24);
            printf("2 WG/CU resident, grid %4d, layout %d, nfma 256 nlds 32: %6.1f us\n", nwg, layout, t);
        }
    return 0;
}
