// Probe (round 6): what the STORE PATTERN of the channels-last plane-sweep forward costs on its own, per element size.
// Volume (B,D,h,w,G) with G = 16 elements of EB bytes per pixel; an 8-wave workgroup owns a tile of PPW x 8 pixels (one tile row per
// wave) and walks `steps` hypothesis planes, every lane storing VB bytes per step -- nothing else (optionally `nfma` dependent FMAs
// per step to pace it like the real walk).  Cases:
//   fp32  VB = 16: 4 lanes per pixel, 16 pixels per wave, 1 KB contiguous per wave and step   (the fp32 kernel)
//   fp16  VB =  8: 4 lanes per pixel, 16 pixels per wave, 512 B per wave and step              (the 2-byte kernels, rounds 2-5)
//   fp16  VB = 16: 2 lanes per pixel, 32 pixels per wave, 1 KB per wave and step               (a lane owning 8 groups)
//   hipcc -O3 --offload-arch=gfx950 cv_store_probe2.hip -o cv_store_probe2 && ./cv_store_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
constexpr int B = 6, D = 96, G = 16, H = 48, W = 160;

template <int EB, int VB>
__global__ __launch_bounds__(512, 1) void probe(char *__restrict__ out, int k, int nfma, float seed) {
    constexpr int LPP = G * EB / VB, PPW = 64 / LPP, TH = 8;
    constexpr int TX = W / PPW, TILES = TX * (H / TH);
    extern __shared__ float pad[];   // sized by the host so that two workgroups fit a CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane % LPP;
    const int item = blockIdx.x / k, part = blockIdx.x % k;
    if (item >= B * TILES) return;
    const int b = item / TILES, tile = item % TILES;
    const int x = (tile % TX) * PPW + lane / LPP, y = (tile / TX) * TH + wave;
    const int d0 = D * part / k, d1 = D * (part + 1) / k;
    const size_t sd = (size_t)H * W * G * EB;
    char *p = out + (size_t)b * D * sd + (size_t)d0 * sd + ((size_t)y * W + x) * G * EB + (size_t)m * VB;
    float a = seed + lane, c = 1.0001f;
    if (tid == 0) pad[0] = a;
    for (int d = d0; d < d1; ++d, p += sd) {
        for (int i = 0; i < nfma; ++i) a = fmaf(a, c, 0.5f);
        if (VB == 16) *reinterpret_cast<float4 *>(p) = make_float4(a, a, a, a);
        else *reinterpret_cast<float2 *>(p) = make_float2(a, a);
    }
}

template <int EB, int VB>
void run(const char *name, int k, int nfma, int ldsbytes) {
    constexpr int LPP = G * EB / VB, PPW = 64 / LPP;
    const int items = B * (W / PPW) * (H / 8);
    const size_t bytes = (size_t)B * D * H * W * G * EB;
    const int NB = 8;
    std::vector<char *> bufs(NB);
    for (auto &q : bufs) { hipMalloc(&q, bytes); hipMemset(q, 0, bytes); }
    hipFuncSetAttribute((const void *)probe<EB, VB>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsbytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, sum = 0.f;
    const int iters = 40;
    for (int it = 0; it < iters + 4; ++it) {
        hipEventRecord(e0);
        probe<EB, VB><<<items * k, 512, ldsbytes>>>(bufs[it % NB], k, nfma, (float)it);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it >= 4) { sum += ms; best = ms < best ? ms : best; }
    }
    printf("%-34s k=%d items=%4d wgs=%4d nfma=%3d lds=%6d: avg %6.1f us  min %6.1f us  %6.0f GB/s (avg)  %5.1f%% of 8 TB/s\n", name, k, items, items * k, nfma,
           ldsbytes, sum / iters * 1e3, best * 1e3, bytes / (sum / iters * 1e-3) * 1e-9, bytes / (sum / iters * 1e-3) * 1e-9 / 80.0);
    for (auto q : bufs) hipFree(q);
}

int main() {
    for (int lds : {38272, 70000}) {           // two workgroups per CU (as the kernel) / one
        for (int nfma : {0, 40, 160}) {
            run<4, 16>("fp32, 16 B/lane, 1 KB/wave-step", 2, nfma, lds);
            run<2, 8>("fp16,  8 B/lane, 512 B/wave-step", 2, nfma, lds);
            run<2, 16>("fp16, 16 B/lane, 1 KB/wave-step", 4, nfma, lds);
            run<2, 16>("fp16, 16 B/lane, 1 KB/wave-step", 2, nfma, lds);
        }
    }
    return 0;
}
