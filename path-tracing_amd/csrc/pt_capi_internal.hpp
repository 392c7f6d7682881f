// Types shared by the translation units behind the C ABI (pt_capi.cpp: scenes, sessions, resolve, BMP; pt_frame.cpp: the
// multi-device frame).  Nothing here is part of the ABI.
#pragma once
#include "../../include/pt_hip.h"

#include <hip/hip_runtime_api.h>

#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "pt_kernels.hpp"
#include "pt_scene.hpp"

// The host side of a scene: parsed model, its device-independent tables and the culling hierarchies built so far (one per
// eps).  Immutable once the scene is finished and shared by every per-device copy of it (pt_scene_clone_to_device, pt_frame):
// the OBJ is parsed once and the hierarchy built once per (scene, eps), however many devices render it.
struct pt_scene_host {
    pt::HostScene host;
    pt::DeviceTables tables;
    std::mutex cull_mutex;
    std::vector<std::shared_ptr<const pt::CullTables>> cull_cache;   // most recent last; a handful of eps values at most
    double load_seconds = 0;                                         // parsing + per-triangle tables
    double cull_build_seconds = 0;                                   // host time spent in build_cull_tables (pt_render -TIMING)
};

// Skybox texels (B,G,R; top-down rows; no padding).  A skybox belongs to a pt_scene handle, not to the shared model: a per-device
// copy inherits the one of the handle it was made FROM (the pointer is copied; texels are immutable once set).
struct pt_sky_texels {
    std::vector<uint8_t> texels;
    int w = 0, h = 0;
};

// Device copy of one CullTables (they depend on eps; a scene keeps the one of the last eps it rendered with).
struct DeviceCull {
    float eps = 0;
    bool valid = false;
    std::shared_ptr<const pt::CullTables> host;
    pt::ClusterDesc *clusters = nullptr;
    pt::SphereRec *spheres = nullptr;
    pt::CullRec *bary = nullptr;
    pt::CullRec *bary_all = nullptr;
    pt::ExactRec *exact_slot = nullptr;
    pt::BvhNode *bvh = nullptr;
};

// What one stream of launches needs besides the scene: the scheduler words of the integrator (ticket + per-tile chunk
// counters), the statistics block and the timing events.  Every pt_session owns one, so that sessions of ONE scene (row bands
// of a frame on one device) run concurrently; a scene has one of its own for pt_render_device / pt_render_host.
struct LaunchCtx {
    uint32_t *d_sched = nullptr;
    size_t sched_words = 0;
    unsigned long long *d_stats = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr;
    bool has_prev = false;          // launches of one context are ordered on the device: they share its scheduler words
    hipStream_t prev_stream = nullptr;
    uint32_t last_chunks = 0;       // of the last launch enqueued (reported with its statistics)
    bool stats_pending = false;
    std::mutex mutex;               // one call at a time per context (enqueue + the optional wait for statistics)
};

struct pt_scene {
    std::shared_ptr<pt_scene_host> shared;
    int device = -1;
    DeviceCull cull;
    pt::ExactRec *d_exact = nullptr;
    pt::MatRec *d_mats = nullptr;
    int cu_count = 256;            // compute units of the scene's device
    std::shared_ptr<const pt_sky_texels> sky;   // this handle's skybox (nullptr = none); copies made from it inherit it
    uint8_t *d_sky = nullptr;      // the same texels on this copy's device
    int sky_w = 0, sky_h = 0;
    // ensure_cull + the enqueue of a launch happen under launch_mutex (a concurrent render with another eps must not free
    // the tables in between); nothing waits for the device while holding it.
    std::mutex launch_mutex;
    LaunchCtx ctx;                 // pt_render_device / pt_render_host / pt_trace_rays_host
    // pt_render_host: device band kept between calls + the stream its kernel runs on (guarded by host_mutex)
    std::mutex host_mutex;
    float *d_host_band = nullptr;
    size_t host_band_floats = 0;
    hipStream_t host_stream = nullptr;
};

// A row band's accumulators kept on the device between pass slices (pt_session_*).
struct pt_session {
    pt_scene *scene = nullptr;
    int32_t width = 0, height = 0, row_begin = 0, row_end = 0, row_stride = 1;
    size_t n = 0;                 // pixels of the band
    float *d_band = nullptr;      // owned: sum[3n] | sum2[3n] | count[n], each plane 256-byte aligned; nullptr if the planes are borrowed
    float *d_sum = nullptr, *d_sum2 = nullptr;
    int32_t *d_count = nullptr;
    hipStream_t stream = nullptr;
    LaunchCtx ctx;
};

namespace ptc {

int fail(int code, const std::string &msg);
int hip_fail(hipError_t e, const char *what);
#define PT_HIP_TRY(expr)                                       \
    do {                                                       \
        hipError_t e_ = (expr);                                \
        if (e_ != hipSuccess) return ptc::hip_fail(e_, #expr); \
    } while (0)

// No exception may cross the C boundary: allocation failures and anything else become status codes.
template <class F>
int guarded(F &&f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return fail(PT_ERR_OUT_OF_MEMORY, "out of host memory");
    } catch (const std::exception &e) {
        return fail(PT_ERR_INVALID_ARGUMENT, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(PT_ERR_INVALID_ARGUMENT, "internal error");
    }
}

int check_params(const pt_scene *scene, const pt_render_params *p);
// A session whose planes live in memory the caller owns (the root band of a frame renders straight into the frame's planes).
int session_create_on(pt_scene *scene, int32_t width, int32_t height, int32_t row_begin, int32_t row_end, float *d_sum,
                      float *d_sum2, int32_t *d_count, pt_session **out, int32_t row_stride = 1);
// rows the accumulator planes of a call hold (pt_band_rows)
int32_t band_rows(const pt_render_params *p);
// pt_session_render in two halves: enqueue the slice (never waits for the device), then -- if statistics were asked for --
// wait for it and read them.  A frame enqueues on every device before it waits on any.
int session_enqueue(pt_session *s, const pt_render_params *p, bool want_stats);
int session_collect(pt_session *s, pt_render_stats *stats);
void ctx_destroy(LaunchCtx &c);

}  // namespace ptc
