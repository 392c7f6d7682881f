// The C API's error paths (pt_capi.cpp, pt_frame.cpp) under AddressSanitizer + UBSan, on a box with or without a GPU: a scene
// that cannot be finished (no device, ordinal out of range) is destroyed inside the call -- nothing may touch it afterwards --
// and a skybox belongs to the handle it was set on.  Built and run by tests/test_host_sanitizers.py; the kernel launchers are
// stubbed, nothing here renders.  argv[1] = models directory (with trailing slash), argv[2] = a 24-bit BMP to use as skybox.
#include "pt_hip.h"
#include "pt_kernels.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
namespace pt {   // what pt_capi.cpp / pt_frame.cpp link against in the real library (pt_kernels.hip, pt_filters.hip)
hipError_t launch_integrator(const RenderArgs &, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_trace_rays(const RenderArgs &, const float *, const float *, int, int32_t *, float *, hipStream_t) { return hipErrorNoDevice; }
hipError_t integrator_waves_per_cu(const RenderArgs &, int *) { return hipErrorNoDevice; }
void integrator_plan_tiles(RenderArgs &, int, int) {}
hipError_t launch_gauss(const float *, float *, const float *, int, int, int, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_median(const float *, float *, int, int, int, hipStream_t) { return hipErrorNoDevice; }
}
#define EXPECT(cond)                                                           \
    do {                                                                       \
        if (!(cond)) {                                                         \
            std::printf("line %d: %s  (last error: %s)\n", __LINE__, #cond, pt_last_error()); \
            return 1;                                                          \
        }                                                                      \
    } while (0)
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const int bad_device = 9999;
    pt_scene *sc = reinterpret_cast<pt_scene *>(0x1);
    // the documented PT_ERR_NO_DEVICE path of every loader: the half-built scene is freed inside the call
    EXPECT(pt_scene_load_obj(argv[1], "Tor.obj", bad_device, &sc) == PT_ERR_NO_DEVICE && sc == nullptr);
    EXPECT(pt_scene_load_obj(argv[1], "NoSuchFile.obj", -1, &sc) == PT_ERR_IO && sc == nullptr);
    const float tri[PT_TRIANGLE_FLOATS] = {0, 0, 1, -1, 0, 0, 1, 1, 0, 1, 0, 1, 1, 1};
    const int32_t tm[1] = {0};
    const float mat[PT_MATERIAL_FLOATS] = {1, 1, 1, 0, 0, 0, 0.5f, 0.5f, 0.5f, 96.0f};
    EXPECT(pt_scene_create(tri, tm, 1, mat, 1, bad_device, &sc) == PT_ERR_NO_DEVICE && sc == nullptr);
    const int32_t tm_bad[1] = {3};
    EXPECT(pt_scene_create(tri, tm_bad, 1, mat, 1, -1, &sc) == PT_ERR_INVALID_ARGUMENT && sc == nullptr);
    // a host-only scene, copies of it, and a copy that fails
    EXPECT(pt_scene_load_obj(argv[1], "Tor.obj", -1, &sc) == PT_OK && sc);
    double secs[2] = {-1, -1};
    EXPECT(pt_scene_timings(sc, secs) == PT_OK && secs[0] > 0);
    pt_scene *cp = nullptr, *cp2 = nullptr, *none = reinterpret_cast<pt_scene *>(0x1);
    EXPECT(pt_scene_clone_to_device(sc, bad_device, &none) == PT_ERR_NO_DEVICE && none == nullptr);
    // a skybox belongs to the handle it is set on: a copy made from `sc` inherits it, clearing it on the copy leaves `sc` alone
    EXPECT(pt_scene_set_skybox_bmp(sc, argv[2]) == PT_OK);
    EXPECT(pt_scene_clone_to_device(sc, -1, &cp) == PT_OK && cp);
    EXPECT(pt_scene_set_skybox_bmp(cp, "") == PT_OK);
    EXPECT(pt_scene_clone_to_device(sc, -1, &cp2) == PT_OK && cp2);
    int32_t sky_dims[3][2];
    EXPECT(pt_scene_skybox_size(sc, &sky_dims[0][0], &sky_dims[0][1]) == PT_OK);
    EXPECT(pt_scene_skybox_size(cp, &sky_dims[1][0], &sky_dims[1][1]) == PT_OK);
    EXPECT(pt_scene_skybox_size(cp2, &sky_dims[2][0], &sky_dims[2][1]) == PT_OK);
    EXPECT(sky_dims[0][0] > 0 && sky_dims[0][1] > 0 && sky_dims[1][0] == 0 && sky_dims[1][1] == 0);
    EXPECT(sky_dims[2][0] == sky_dims[0][0] && sky_dims[2][1] == sky_dims[0][1]);
    EXPECT(pt_scene_set_skybox_bmp(sc, "/nonexistent/sky.bmp") == PT_ERR_IO);
    EXPECT(pt_scene_skybox_size(sc, &sky_dims[0][0], &sky_dims[0][1]) == PT_OK && sky_dims[0][0] == sky_dims[2][0]);   // unchanged by the failed call
    // a frame over a device that does not exist: refused, nothing leaks, the scene stays usable
    pt_frame *fr = reinterpret_cast<pt_frame *>(0x1);
    const int32_t devs[1] = {bad_device};
    EXPECT(pt_frame_create(sc, devs, 1, 64, 64, 0, &fr) == PT_ERR_NO_DEVICE && fr == nullptr);
    // render entry points on a host-only scene
    pt_render_params p;
    std::memset(&p, 0, sizeof p);
    p.width = 8; p.height = 8; p.row_end = 8; p.pass_count = 1; p.max_ray_reflections = 3; p.eps = 1e-4f; p.error = -1.0f; p.seed = 42;
    std::vector<float> s(8 * 8 * 3), s2(8 * 8 * 3);
    std::vector<int32_t> c(8 * 8);
    EXPECT(pt_render_host(sc, &p, s.data(), s2.data(), c.data(), nullptr) == PT_ERR_NO_DEVICE);
    pt_session *se = reinterpret_cast<pt_session *>(0x1);
    EXPECT(pt_session_create(sc, 8, 8, 0, 8, &se) == PT_ERR_NO_DEVICE && se == nullptr);
    int32_t counts[4] = {0, 0, 0, 0};
    EXPECT(pt_scene_cull_layout(cp2, 1e-4f, counts, nullptr, nullptr) == PT_OK && counts[0] > 0);
    pt_scene_destroy(sc);        // any order
    EXPECT(pt_scene_cull_layout(cp, 1e-4f, counts, nullptr, nullptr) == PT_OK && counts[0] > 0);
    pt_scene_destroy(cp2);
    pt_scene_destroy(cp);
    std::printf("capi error paths ok\n");
    return 0;
}
