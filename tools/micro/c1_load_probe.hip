// Read (and write) ceilings of the access patterns of the 16 -> 1 convolution kernels (csrc/conv3d_c1.hip) on a
// channels-last volume 6 x 96 x 48 x 160 x 16 fp32 (283 MB): persistent workgroups that own an (h, w) tile of one sample and
// march over a slice of D one plane per step, against contiguous chunks and a plain grid-stride stream.
//   hipcc -O3 --offload-arch=gfx950 c1_load_probe.hip -o c1_load_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

constexpr int B = 6, D = 96, H = 48, W = 160, C = 16;
constexpr size_t PLANE = (size_t)H * W;          // voxels per plane
constexpr size_t NVOX = (size_t)B * D * PLANE;

// PAT 0: 8x32 tile, dword loads in MFMA-operand order (16 per lane per plane: 4 voxels x 16 channels = 256 B per wave-instr)
// PAT 1: 8x32 tile, float4 loads (4 per lane per plane, 1 KB per wave-instr)
// PAT 2: 16 KB contiguous chunk per workgroup and plane, float4 loads
// PAT 3: 16 KB contiguous chunk, dword loads
// PAT 5: 16x32 tile + 1 halo (18x34 cells), float4 loads (the forward's staging), slices re-read 2 halo planes
// AHEAD: planes in flight ahead of the consumer (1 = load plane d+1 while consuming d)
template <int PAT, int AHEAD, bool STORE>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ x, float *__restrict__ out, float *__restrict__ big, int planes, int dslices) {
    extern __shared__ float dummy[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int TH = PAT == 5 ? 16 : 8, TW = 32;
    constexpr int tiles_x = W / TW, tiles = tiles_x * (H / TH);
    const int item = blockIdx.x, sl = item % dslices, t = (item / dslices) % tiles, b = item / (dslices * tiles);
    const int tx0 = (t % tiles_x) * TW, ty0 = (t / tiles_x) * TH;
    int d0 = sl * planes, d1 = min(d0 + planes, D);
    if (PAT == 5) { d0 = max(d0 - 1, 0); d1 = min(d1 + 1, D); }
    constexpr int N = PAT == 0 || PAT == 3 ? 16 : PAT == 5 ? 10 : 4;
    int ofs[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (PAT == 0) {
            const int yy = ty0 + 2 * wave + (i >> 3), xx = tx0 + (i & 7) * 4 + (lane >> 4);
            ofs[i] = (yy * W + xx) * 16 + (lane & 15);
        } else if (PAT == 1) {
            const int idx = tid + i * 256, v = idx >> 2, q = idx & 3;
            ofs[i] = ((ty0 + v / TW) * W + tx0 + v % TW) * 4 + q;
        } else if (PAT == 2) {
            ofs[i] = t * 1024 + tid + i * 256;  // float4 units; 30 chunks of 16 KB per plane
        } else if (PAT == 3) {
            ofs[i] = t * 4096 + tid + i * 256;
        } else {
            const int idx = tid + i * 256, cell = idx >> 2, q = idx & 3;
            const int yy = ty0 - 1 + cell / 34, xx = tx0 - 1 + cell % 34;
            ofs[i] = (idx < 612 * 4 && yy >= 0 && yy < H && xx >= 0 && xx < W) ? (yy * W + xx) * 4 + q : -1;
        }
    }
    constexpr bool V4 = !(PAT == 0 || PAT == 3);
    const float *xb = x + (size_t)b * D * PLANE * C;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 buf[AHEAD + 1][V4 ? N : N / 4];
    auto load = [&](int p, int slot) {
        if (V4) {
            const float4 *x4 = reinterpret_cast<const float4 *>(xb) + (size_t)p * PLANE * 4;
#pragma unroll
            for (int i = 0; i < N; ++i) buf[slot][i] = (PAT != 5 || ofs[i] >= 0) ? x4[ofs[i]] : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const float *x1 = xb + (size_t)p * PLANE * 16;
            float *bf = reinterpret_cast<float *>(buf[slot]);
#pragma unroll
            for (int i = 0; i < N; ++i) bf[i] = x1[ofs[i]];
        }
    };
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
        if (d0 + a < d1) load(d0 + a, a);
    int slot = 0;
    for (int d = d0; d < d1; ++d) {
        // static slot indices: rotate by copying (register moves only)
        if (d + AHEAD < d1) load(d + AHEAD, AHEAD);
#pragma unroll
        for (int i = 0; i < (V4 ? N : N / 4); ++i) {
            acc.x += buf[0][i].x; acc.y += buf[0][i].y; acc.z += buf[0][i].z; acc.w += buf[0][i].w;
        }
        if (STORE && PAT == 1) {
            float4 *o4 = reinterpret_cast<float4 *>(big + (size_t)b * D * PLANE * C) + (size_t)d * PLANE * 4;
#pragma unroll
            for (int i = 0; i < N; ++i) o4[ofs[i]] = acc;
        }
#pragma unroll
        for (int a = 0; a < AHEAD; ++a)
#pragma unroll
            for (int i = 0; i < (V4 ? N : N / 4); ++i) buf[a][i] = buf[a + 1][i];
        (void)slot;
    }
    out[(size_t)blockIdx.x * 256 + tid] = acc.x + acc.y + acc.z + acc.w;
}

// Store-only: the data gradient's pattern.  A wave writes its two tile rows as 4 x 1 KB per plane (16 voxels x 64 B each).
// ORDER 0: lane l writes piece l of the 1 KB (voxel l >> 2, quad l & 3) -- memory order;
// ORDER 1: lane l writes voxel l & 15, quad l >> 4 -- the MFMA D layout (same 1 KB per instruction, lanes transposed).
template <int ORDER>
__global__ __launch_bounds__(256) void store_probe(float *__restrict__ big, int planes, int dslices, float val) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int TH = 8, TW = 32, tiles_x = W / TW, tiles = tiles_x * (H / TH);
    const int item = blockIdx.x, sl = item % dslices, t = (item / dslices) % tiles, b = item / (dslices * tiles);
    const int tx0 = (t % tiles_x) * TW, ty0 = (t / tiles_x) * TH;
    const int d0 = sl * planes, d1 = min(d0 + planes, D);
    const int vox = ORDER == 0 ? lane >> 2 : lane & 15, q = ORDER == 0 ? lane & 3 : lane >> 4;
    int ofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ofs[i] = ((ty0 + 2 * wave + (i >> 1)) * W + tx0 + (i & 1) * 16 + vox) * 4 + q;
    float4 *o4 = reinterpret_cast<float4 *>(big + (size_t)b * D * PLANE * C);
    float4 v = make_float4(val, val + 1.f, val + 2.f, val + 3.f);
    for (int d = d0; d < d1; ++d) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o4[(size_t)d * PLANE * 4 + ofs[i]] = v;
        v.x += 1.f;
    }
}

// The plane-sweep backward's gradient read: workgroup = 16 x 4 pixel tile (wave = one row of 16 pixels = 1 KB per plane),
// walks ALL 96 planes; BATCH planes are requested together, AHEAD batches before they are consumed.
template <int BATCH, int AHEAD>
__global__ __launch_bounds__(256) void sweep_probe(const float *__restrict__ x, float *__restrict__ out, int nitems, int work) {
    extern __shared__ float dummy2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = W / 16, tiles = tiles_x * (H / 4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int t = item % tiles, b = item / tiles;
        const int tx0 = (t % tiles_x) * 16, ty0 = (t / tiles_x) * 4;
        const float4 *p = reinterpret_cast<const float4 *>(x + (size_t)b * D * PLANE * C) + ((ty0 + wave) * W + tx0) * 4 + lane;
        float4 q[AHEAD + 1][BATCH];
#pragma unroll
        for (int a = 0; a < AHEAD; ++a)
#pragma unroll
            for (int j = 0; j < BATCH; ++j) q[a][j] = p[(size_t)(a * BATCH + j) * PLANE * 4];
        for (int d = 0; d < D; d += BATCH) {
#pragma unroll
            for (int j = 0; j < BATCH; ++j) q[AHEAD][j] = p[(size_t)min(d + AHEAD * BATCH + j, D - 1) * PLANE * 4];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) { acc.x += q[0][j].x; acc.y += q[0][j].y; acc.z += q[0][j].z; acc.w += q[0][j].w; }
            for (int k = 0; k < work; ++k) {  // stand-in for the batch's arithmetic: a dependent chain of 4 FMAs per round
                acc.x = fmaf(acc.x, 1.0001f, acc.y); acc.y = fmaf(acc.y, 0.9999f, acc.z);
                acc.z = fmaf(acc.z, 1.0001f, acc.w); acc.w = fmaf(acc.w, 0.9999f, acc.x);
            }
#pragma unroll
            for (int a = 0; a < AHEAD; ++a)
#pragma unroll
                for (int j = 0; j < BATCH; ++j) q[a][j] = q[a + 1][j];
        }
    }
    out[(size_t)blockIdx.x * 256 + tid] = acc.x + acc.y + acc.z + acc.w;
}

// plain stream: grid-stride float4 reads of the whole volume
__global__ __launch_bounds__(256) void stream(const float4 *__restrict__ x, float *__restrict__ out, size_t n4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = x[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <typename F>
double time_us(F launch, int iters = 20) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch(i);
    hipDeviceSynchronize();
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
        hipEventRecord(a);
        launch(i);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    const size_t bytes = NVOX * C * 4;
    constexpr int ROT = 4;  // rotate over 4 input volumes (1.1 GB) so that the 256 MB cache holds nothing of the next launch
    float *x[ROT], *out, *big;
    for (int r = 0; r < ROT; ++r) { hipMalloc(&x[r], bytes); hipMemset(x[r], 0, bytes); }
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&big, bytes);
    hipDeviceSynchronize();
    auto rep = [&](const char *name, double us, double mb) { printf("%-64s %7.1f us  %6.0f GB/s (%4.1f%% of 8 TB/s)\n", name, us, mb / us * 1e3, mb / us / 8 * 100); fflush(stdout); };
    const double mb = bytes / 1e6;
    for (int grid : {1024, 2048, 4096}) {
        char nm[96]; snprintf(nm, 96, "stream float4, %d workgroups", grid);
        rep(nm, time_us([&](int i) { hipLaunchKernelGGL(stream, dim3(grid), dim3(256), 0, 0, (const float4 *)x[i % ROT], out, bytes / 16); }), mb);
    }
#define RUN(PAT, AHEAD, STORE, planes, lds, label)                                                                          \
    {                                                                                                                       \
        const int tiles = (PAT == 5 ? 15 : 30), ds = (D + planes - 1) / planes, grid = B * tiles * ds;                       \
        char nm[128]; snprintf(nm, 128, "%s, %d planes/WG, %d WGs, ahead %d, lds %d KB", label, planes, grid, AHEAD, lds);   \
        rep(nm, time_us([&](int i) { hipLaunchKernelGGL((probe<PAT, AHEAD, STORE>), dim3(grid), dim3(256), lds * 1024, 0, x[i % ROT], out, big, planes, ds); }), mb * (STORE ? 2 : 1)); \
    }
    RUN(0, 1, false, 12, 0, "8x32 tile, dword loads (MFMA order)");
    RUN(0, 2, false, 12, 0, "8x32 tile, dword loads (MFMA order)");
    RUN(0, 1, false, 24, 0, "8x32 tile, dword loads (MFMA order)");
    RUN(0, 1, false, 6, 0, "8x32 tile, dword loads (MFMA order)");
    RUN(0, 1, false, 12, 20, "8x32 tile, dword loads (MFMA order)");
    RUN(1, 1, false, 12, 0, "8x32 tile, float4 loads");
    RUN(1, 2, false, 12, 0, "8x32 tile, float4 loads");
    RUN(1, 1, false, 6, 0, "8x32 tile, float4 loads");
    RUN(2, 1, false, 12, 0, "16 KB contiguous chunk, float4 loads");
    RUN(2, 2, false, 12, 0, "16 KB contiguous chunk, float4 loads");
    RUN(3, 1, false, 12, 0, "16 KB contiguous chunk, dword loads");
    RUN(5, 1, false, 12, 39, "16x32 tile + halo (fwd staging), float4");
    RUN(5, 1, false, 24, 39, "16x32 tile + halo (fwd staging), float4");
    RUN(5, 1, false, 12, 0, "16x32 tile + halo (fwd staging), float4");
    RUN(1, 1, true, 12, 0, "8x32 tile, float4 load + float4 store (copy)");
#define SWEEP(BATCH, AHEAD, grid, lds, work)                                                                                        \
    {                                                                                                                               \
        char nm[160]; snprintf(nm, 160, "plane-sweep gradient read: batch %d, %d ahead, %d WGs, %d KB LDS/WG, %d FMA rounds per batch", BATCH, AHEAD, grid, lds, work); \
        rep(nm, time_us([&](int i) { hipLaunchKernelGGL((sweep_probe<BATCH, AHEAD>), dim3(grid), dim3(256), lds * 1024, 0, x[i % ROT], out, B * (W / 16) * (H / 4), work); }), mb); \
    }
    SWEEP(4, 0, 720, 0, 0);
    SWEEP(4, 1, 720, 0, 0);
    SWEEP(4, 3, 720, 0, 0);
    SWEEP(4, 1, 512, 0, 0);
    SWEEP(4, 3, 512, 0, 0);
    // the kernel's residency: 2 workgroups per CU (64 KB of LDS each), 720 workgroups = 1.4 rounds
    SWEEP(4, 0, 720, 64, 0);
    SWEEP(4, 1, 720, 64, 0);
    SWEEP(4, 2, 720, 64, 0);
    SWEEP(4, 0, 720, 64, 100);
    SWEEP(4, 1, 720, 64, 100);
    SWEEP(4, 2, 720, 64, 100);
    SWEEP(4, 0, 720, 64, 300);
    SWEEP(4, 1, 720, 64, 300);
    SWEEP(4, 2, 720, 64, 300);
    float *bigs[ROT];
    bigs[0] = big;
    for (int r = 1; r < ROT; ++r) bigs[r] = x[r];
    rep("store only, 8x32 tile, 1 KB per wave, lanes in memory order, 1440 WGs", time_us([&](int i) { hipLaunchKernelGGL(store_probe<0>, dim3(1440), dim3(256), 0, 0, bigs[i % ROT], 12, 8, 1.f); }), mb);
    rep("store only, 8x32 tile, 1 KB per wave, lanes in MFMA-D order, 1440 WGs", time_us([&](int i) { hipLaunchKernelGGL(store_probe<1>, dim3(1440), dim3(256), 0, 0, bigs[i % ROT], 12, 8, 1.f); }), mb);
    rep("store only, memory order, 2880 WGs x 6 planes", time_us([&](int i) { hipLaunchKernelGGL(store_probe<0>, dim3(2880), dim3(256), 0, 0, bigs[i % ROT], 6, 16, 1.f); }), mb);
    rep("store only, MFMA-D order, 2880 WGs x 6 planes", time_us([&](int i) { hipLaunchKernelGGL(store_probe<1>, dim3(2880), dim3(256), 0, 0, bigs[i % ROT], 6, 16, 1.f); }), mb);
    return 0;
}
