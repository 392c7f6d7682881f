// The C API's host side (pt_capi.cpp: threaded pt_resolve / pt_resolve_float, host-only scenes, the shared hierarchy cache hit
// from several threads) under ThreadSanitizer.  Built and run by tests/test_host_sanitizers.py without a GPU: the kernel
// launchers are stubbed, nothing here renders.  argv[1] = models directory (with trailing slash).
#include "pt_hip.h"
#include "pt_kernels.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
namespace pt {   // what pt_capi.cpp / pt_frame.cpp link against in the real library (pt_kernels.hip, pt_filters.hip)
hipError_t launch_integrator(const RenderArgs &, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_trace_rays(const RenderArgs &, const float *, const float *, int, int32_t *, float *, hipStream_t) { return hipErrorNoDevice; }
hipError_t integrator_waves_per_cu(const RenderArgs &, int *) { return hipErrorNoDevice; }
void integrator_plan_tiles(RenderArgs &, int, int) {}
hipError_t launch_gauss(const float *, float *, const float *, int, int, int, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_median(const float *, float *, int, int, int, hipStream_t) { return hipErrorNoDevice; }
}
int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const int W = 1024, H = 700;
    const size_t n = size_t(W) * H;
    std::mt19937 rng(3);
    std::uniform_real_distribution<float> u(0, 1);
    std::vector<float> s(3 * n), s2(3 * n), rgb(3 * n), rgb1(3 * n);
    std::vector<int32_t> c(n);
    for (size_t p = 0; p < n; ++p) {
        c[p] = int(rng() % 4);
        for (int k = 0; k < 3; ++k) { s[3 * p + k] = u(rng) * c[p]; s2[3 * p + k] = s[3 * p + k] * s[3 * p + k] / (c[p] ? c[p] : 1) * (p % 7 ? 1.0f : 1.5f); }
    }
    std::vector<uint8_t> bgr(3 * n), bgr1(3 * n);
    float d[3], d1[3], d2[3];
    if (pt_resolve(W, H, s.data(), s2.data(), c.data(), 1 / 2.2f, bgr.data(), d) != PT_OK) return 1;          // threaded (>= 2^18 pixels)
    if (pt_resolve_float(W, H, s.data(), s2.data(), c.data(), 1 / 2.2f, rgb.data(), d2) != PT_OK) return 1;
    // the same image in strips small enough for the sequential path, statistics compared through the whole-image sums below
    const int strip = 100;   // 1024 x 100 < 2^18
    for (int y = 0; y < H; y += strip) {
        const int h = H - y < strip ? H - y : strip;
        if (pt_resolve(W, h, &s[3 * size_t(y) * W], &s2[3 * size_t(y) * W], &c[size_t(y) * W], 1 / 2.2f, &bgr1[3 * size_t(y) * W], d1) != PT_OK) return 1;
    }
    if (std::memcmp(bgr.data(), bgr1.data(), bgr.size()) != 0) { std::printf("threaded resolve differs from the sequential one\n"); return 1; }
    if (std::memcmp(d, d2, sizeof d) != 0) { std::printf("resolve and resolve_float disagree on the statistics\n"); return 1; }
    // several threads asking one scene (and a host-only copy of it) for the hierarchy of the same and of different eps
    pt_scene *sc = nullptr, *cp = nullptr;
    if (pt_scene_load_obj(argv[1], "Tor.obj", -1, &sc) != PT_OK || pt_scene_clone_to_device(sc, -1, &cp) != PT_OK) return 1;
    std::vector<std::thread> th;
    int bad = 0;
    for (int t = 0; t < 6; ++t)
        th.emplace_back([&, t] {
            int32_t counts[4];
            const float eps = t % 2 ? 1e-4f : 1e-3f;
            if (pt_scene_cull_layout(t % 3 ? sc : cp, eps, counts, nullptr, nullptr) != PT_OK || counts[0] <= 0) __atomic_add_fetch(&bad, 1, __ATOMIC_RELAXED);
        });
    for (auto &x : th) x.join();
    double secs[2];
    pt_scene_timings(cp, secs);
    pt_scene_destroy(sc);
    pt_scene_destroy(cp);
    std::printf("resolve ok, dispersion %.6f %.6f %.6f, hierarchy threads bad %d, builds %.4f s\n", d[0], d[1], d[2], bad, secs[1]);
    return bad ? 1 : 0;
}
