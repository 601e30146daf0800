// Error plumbing + the small standalone kernels of the C ABI (schedule).
#include <string.h>

#include "md_common.hpp"

static thread_local char g_err[512] = "";

void md_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *md_last_error(void) { return g_err; }
extern "C" int md_abi_version(void) { return 17; }

// ---------------------------------------------------------------- per-kernel timing (measurement only)
#include <vector>
namespace {
struct TimingRec { const char *name; hipEvent_t a, b; };
int g_timing = 0;   // bit mask of kernel classes, md_kernel_timing_enable
std::vector<TimingRec> g_recs;
}  // namespace

// Work counters of the channels-last plane-sweep kernels (diagnostics: how often windows are staged, taps miss them, cells
// change).  8 x u64 in device memory, allocated on first enable; behind them MD_STATS_WG_CAP more: the lifetime of every
// workgroup of the launch in shader cycles (entry 8 + blockIdx.x), i.e. how evenly the launch's work was spread.
namespace { unsigned long long *g_stats = nullptr; bool g_stats_on = false; }
constexpr int MD_STATS_WG_CAP = 8192;
unsigned long long *md_stats_buffer() { return g_stats_on ? g_stats : nullptr; }
extern "C" int md_costvol_stats_wg(unsigned long long *out, int cap) {
    MD_REQUIRE(out && cap > 0, "md_costvol_stats_wg: null argument");
    if (!g_stats) return 0;
    const int n = cap < MD_STATS_WG_CAP ? cap : MD_STATS_WG_CAP;
    MD_CHECK_HIP(hipDeviceSynchronize());
    MD_CHECK_HIP(hipMemcpy(out, g_stats + 8, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return n;
}
extern "C" int md_costvol_stats(int enable, unsigned long long *out8) {
    if (enable && !g_stats) MD_CHECK_HIP(hipMalloc(&g_stats, (8 + MD_STATS_WG_CAP) * sizeof(unsigned long long)));
    if (out8 && g_stats) {
        MD_CHECK_HIP(hipDeviceSynchronize());
        MD_CHECK_HIP(hipMemcpy(out8, g_stats, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    if (g_stats) MD_CHECK_HIP(hipMemset(g_stats, 0, (8 + MD_STATS_WG_CAP) * sizeof(unsigned long long)));
    g_stats_on = enable != 0;
    return MD_OK;
}

// A start / stop event pair for ONE kernel dispatch (hipExtLaunchKernelGGL ties them to the dispatch's own begin / end
// timestamps -- the clock rocprofv3's kernel trace reads -- so host launch latency is not part of the interval); null, null
// when timing is off.
void md_timing_pair(const char *name, hipEvent_t *start, hipEvent_t *stop) {
    *start = *stop = nullptr;
    if (!g_timing) return;
    // classes (a timed dispatch costs the stream ~5 us: time only what is asked for): 2 = plane sweep + md_conv3d_*,
    // 4 = photometric / smoothness / packing, 8 = md_bn_*
    const int cls = strncmp(name, "md_bn_", 6) == 0 ? 8 : (strncmp(name, "md_costvol", 10) == 0 || strncmp(name, "md_conv3d", 9) == 0) ? 2 : 4;
    if (!(g_timing & cls)) return;
    TimingRec r{name, nullptr, nullptr};
    if (hipEventCreate(&r.a) != hipSuccess) return;
    if (hipEventCreate(&r.b) != hipSuccess) { (void)hipEventDestroy(r.a); return; }
    g_recs.push_back(r);
    *start = r.a;
    *stop = r.b;
}
extern "C" int md_kernel_timing_enable(int on) {
    for (auto &r : g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_recs.clear();
    g_timing = on == 1 ? 14 : on;   // 1 = every class, otherwise the mask of classes (md_timing_pair)
    return MD_OK;
}
extern "C" int md_kernel_timing_list(const char *name, double *us, int cap) {
    MD_REQUIRE(name && (us || cap == 0), "md_kernel_timing_list: null argument");
    int n = 0;
    for (auto &r : g_recs) {
        if (strcmp(r.name, name) != 0) continue;
        float ms = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        if (n < cap) us[n] = 1e3 * ms;
        ++n;
    }
    return n;
}
extern "C" int md_kernel_timing_read(const char *name, double *avg_us, double *min_us, int *launches) {
    MD_REQUIRE(name && avg_us && min_us && launches, "md_kernel_timing_read: null argument");
    double sum = 0.0, mn = 0.0;
    int n = 0;
    for (auto &r : g_recs) {
        if (strcmp(r.name, name) != 0) continue;
        float ms = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        const double us = 1e3 * ms;
        sum += us;
        mn = (n == 0 || us < mn) ? us : mn;
        ++n;
    }
    *avg_us = n ? sum / n : 0.0;
    *min_us = mn;
    *launches = n;
    return MD_OK;
}

namespace {

// schedule_depth_rangev2 / _zv2 (reference layers.py:256-284, 370-398): one thread per (b, k, pixel)
__global__ __launch_bounds__(256) void schedule_kernel(const float *__restrict__ prior, const float *__restrict__ ztrans,
                                                       int hw, int D, float scale_fac, int type, float *__restrict__ out) {
    const int b = blockIdx.z, k = blockIdx.y;
    const float one_pf = 1.f + (ztrans ? scale_fac * ztrans[b] : scale_fac);
    for (int p = blockIdx.x * 256 + threadIdx.x; p < hw; p += gridDim.x * 256)
        out[((size_t)b * D + k) * hw + p] = md_hypothesis(prior[(size_t)b * hw + p], one_pf, k, D, type);
}

}  // namespace

extern "C" int md_schedule_depth_range(const float *prior, const float *ztrans, int B, int h, int w, int D,
                                       float scale_fac, int type, float *out, md_stream_t stream) {
    MD_REQUIRE(prior && out, "md_schedule_depth_range: null tensor");
    MD_REQUIRE(B > 0 && h > 0 && w > 0 && D > 1, "md_schedule_depth_range: bad dims B=%d h=%d w=%d D=%d", B, h, w, D);
    MD_REQUIRE(type >= 0 && type <= 2, "md_schedule_depth_range: bad type %d", type);
    MD_REQUIRE(D <= 65535 && B <= 65535, "md_schedule_depth_range: D/B too large");
    const int hw = h * w;
    dim3 grid(md_cdiv(hw, 256) < 64 ? md_cdiv(hw, 256) : 64, D, B);
    hipLaunchKernelGGL(schedule_kernel, grid, dim3(256), 0, (hipStream_t)stream, prior, ztrans, hw, D, scale_fac, type, out);
    MD_CHECK_LAUNCH("md_schedule_depth_range");
    return MD_OK;
}
