// C ABI of libpt_hip.so (include/pt_hip.h).  Host side only: owns the device copies of the scene tables, maps HIP
// errors to status codes, and implements the reference's host-side resolve and BMP writer.  (pt_frame.cpp holds the
// multi-device frame on top of what is here.)
#include "pt_capi_internal.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "pt_filters.hpp"

#ifdef PT_TEST_HOOKS
static int g_items_per_slot = 0;
static int g_force_tile_width = 0;   // 1 = always 8 x 8 tiles, 2 = always 16 x 8 where the instantiation has them, 3 = the same and 32 x 8 for adaptive launches, 0 = by tile count
static int g_regen_min_dead = 0;     // test build: overrides RenderArgs::regen_min_dead (0 = the library's)
static int g_chunk_min = 0;          // test build: passes the scheduler's last geometric chunk holds at least (0 = the library's)
#endif

namespace ptc {

thread_local std::string g_error;

int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
int hip_fail(hipError_t e, const char *what) {
    const int code = (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
                         ? PT_ERR_NO_DEVICE
                         : (e == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_HIP);
    return fail(code, std::string(what) + ": " + hipGetErrorString(e));
}

void ctx_destroy(LaunchCtx &c) {
    if (c.d_sched) (void)hipFree(c.d_sched);
    if (c.d_stats) (void)hipFree(c.d_stats);
    if (c.ev0) (void)hipEventDestroy(c.ev0);
    if (c.ev1) (void)hipEventDestroy(c.ev1);
    if (c.ev_done) (void)hipEventDestroy(c.ev_done);
    c.d_sched = nullptr;
    c.d_stats = nullptr;
    c.ev0 = c.ev1 = c.ev_done = nullptr;
    c.sched_words = 0;
    c.has_prev = false;
}

}  // namespace ptc

using ptc::fail;
using ptc::guarded;
using ptc::hip_fail;

namespace {

// The layouts the kernels read pack indices into bit fields; a hierarchy that does not fit them must be refused, never
// truncated: (ray, slot) pairs keep the slot in 24 bits (pt_kernels.hip: `e & 0xFFFFFF`), a box-tree node keeps its child
// base in 20 bits (BvhNode::meta, `base << 12`: a child node, or a leaf's first slot / 8 -- fewer leaves than nodes; stack entries have 26), a sphere tree has at most kMaxLevels levels.
int check_table_limits(unsigned long long n_slots, unsigned long long n_bvh_nodes, long long n_levels, long long bvh_depth = 0) {
    if (n_slots >= (1ull << 24))
        return fail(PT_ERR_UNSUPPORTED, "the culling hierarchy has " + std::to_string(n_slots) + " slots; (ray, slot) work items hold 24 bits");
    if (n_bvh_nodes >= (1ull << 20))
        return fail(PT_ERR_UNSUPPORTED, "the box tree has " + std::to_string(n_bvh_nodes) + " nodes; a node's child base holds 20 bits");
    // a leaf's base is its first slot / 8 in the global slot order: with a box tree the slots themselves must stay below 2^23
    // (the builder puts the tree's slots first, so a leaf's base is below the node count anyway: this is the field's own bound)
    if (n_bvh_nodes > 0 && n_slots >= (1ull << 23))
        return fail(PT_ERR_UNSUPPORTED, "the culling hierarchy has " + std::to_string(n_slots) + " slots under a box tree; a leaf's base (first slot / 8) holds 20 bits");
    if (bvh_depth > pt::kMaxBvhDepth)
        return fail(PT_ERR_UNSUPPORTED, "the box tree has " + std::to_string(bvh_depth) + " levels; the walk's stack slack holds " + std::to_string(pt::kMaxBvhDepth));
    if (n_levels > pt::kMaxLevels)
        return fail(PT_ERR_UNSUPPORTED, "a sphere tree has " + std::to_string(n_levels) + " levels; the walk holds " + std::to_string(pt::kMaxLevels));
    return PT_OK;
}

// The culling hierarchy of a scene for one eps: built once per (scene, eps) on the host, whatever number of devices,
// sessions or diagnostic calls ask for it.  (The test-hook build rebuilds every time: its mutations change the result.)
int get_cull(pt_scene_host &h, float eps, std::shared_ptr<const pt::CullTables> &out) {
    std::lock_guard<std::mutex> lock(h.cull_mutex);
#ifndef PT_TEST_HOOKS
    for (const auto &t : h.cull_cache)
        if (std::memcmp(&t->eps, &eps, sizeof eps) == 0) {
            out = t;
            return PT_OK;
        }
#endif
    const auto t0 = std::chrono::steady_clock::now();
    auto t = std::make_shared<pt::CullTables>();
    pt::build_cull_tables(h.host, eps, *t);
    t->eps = eps;
    h.cull_build_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long long levels = 0;
    for (const auto &c : t->clusters)
        if (c.kind == 0) levels = std::max<long long>(levels, c.n_levels);
    const int rc = check_table_limits(t->slot_tri.size(), t->bvh.size(), levels, t->bvh_depth);
    if (rc != PT_OK) return rc;
    // the small-scene kernels' closest-hit key packs (original index, slot) into 16 bits each
    if (!t->big && (t->slot_tri.size() >= 65536u || h.host.n_tri() >= 65536))
        return fail(PT_ERR_UNSUPPORTED, "a scene on the sphere-tree path has " + std::to_string(t->slot_tri.size()) + " slots; its closest-hit key holds 16 bits");
    if (h.cull_cache.size() >= 4) h.cull_cache.erase(h.cull_cache.begin());
    h.cull_cache.push_back(t);
    out = t;
    return PT_OK;
}

int upload(pt_scene *s, int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(PT_ERR_NO_DEVICE, "no HIP device is visible: the integrator has no CPU fallback");
    if (device >= n) return fail(PT_ERR_NO_DEVICE, "device ordinal " + std::to_string(device) + " out of range");
    s->device = device;
    PT_HIP_TRY(hipSetDevice(device));
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) s->cu_count = cus;
    const auto &t = s->shared->tables;
    PT_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_exact), t.exact.size() * sizeof(pt::ExactRec) + 64));
    PT_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_mats), t.mats.size() * sizeof(pt::MatRec) + 64));
    if (!t.exact.empty()) PT_HIP_TRY(hipMemcpy(s->d_exact, t.exact.data(), t.exact.size() * sizeof(pt::ExactRec), hipMemcpyHostToDevice));
    if (!t.mats.empty()) PT_HIP_TRY(hipMemcpy(s->d_mats, t.mats.data(), t.mats.size() * sizeof(pt::MatRec), hipMemcpyHostToDevice));
    if (s->sky && !s->sky->texels.empty()) {   // (a copy's skybox is the one of the handle it was made from)
        const auto &sky = s->sky->texels;
        PT_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_sky), sky.size() + 64));
        PT_HIP_TRY(hipMemcpy(s->d_sky, sky.data(), sky.size(), hipMemcpyHostToDevice));
        s->sky_w = s->sky->w;
        s->sky_h = s->sky->h;
    }
    return PT_OK;
}

// events / statistics block of a launch context, on first use (the scene's device is current)
int ctx_ready(LaunchCtx &c) {
    if (!c.ev_done) PT_HIP_TRY(hipEventCreateWithFlags(&c.ev_done, hipEventDisableTiming));
    return PT_OK;
}
int ctx_stats_ready(LaunchCtx &c) {
    if (!c.d_stats) PT_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c.d_stats), 24 * sizeof(unsigned long long)));
    if (!c.ev0) PT_HIP_TRY(hipEventCreate(&c.ev0));
    if (!c.ev1) PT_HIP_TRY(hipEventCreate(&c.ev1));
    return PT_OK;
}

template <class T>
int upload_vec(const std::vector<T> &v, T **dst) {
    if (*dst) {
        (void)hipFree(*dst);
        *dst = nullptr;
    }
    PT_HIP_TRY(hipMalloc(reinterpret_cast<void **>(dst), v.size() * sizeof(T) + 256));
    if (!v.empty()) PT_HIP_TRY(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return PT_OK;
}

// The cull hierarchy's radii and margins depend on eps (-EPS): upload on first use, replace if eps changes.
// Callers hold scene->launch_mutex from here until their kernel has been enqueued: a concurrent render with another eps
// must not free the tables between this call and that launch.
int ensure_cull(pt_scene *s, float eps) {
    DeviceCull &c = s->cull;
    if (c.valid && std::memcmp(&c.eps, &eps, sizeof eps) == 0) return PT_OK;
    std::shared_ptr<const pt::CullTables> t;
    int rc = get_cull(*s->shared, eps, t);
    if (rc != PT_OK) return rc;
    if (c.valid) PT_HIP_TRY(hipDeviceSynchronize());   // a previous launch may still read the old tables
    c.valid = false;
    c.host = t;
    if ((rc = upload_vec(t->clusters, &c.clusters)) != PT_OK) return rc;
    if ((rc = upload_vec(t->spheres, &c.spheres)) != PT_OK) return rc;
    if ((rc = upload_vec(t->bary, &c.bary)) != PT_OK) return rc;
    if (c.bary_all) {
        (void)hipFree(c.bary_all);
        c.bary_all = nullptr;
    }
    if (!t->bary_all.empty() && (rc = upload_vec(t->bary_all, &c.bary_all)) != PT_OK) return rc;
    if ((rc = upload_vec(t->exact_slot, &c.exact_slot)) != PT_OK) return rc;
    if (c.bvh) {
        (void)hipFree(c.bvh);
        c.bvh = nullptr;
    }
    if (!t->bvh.empty() && (rc = upload_vec(t->bvh, &c.bvh)) != PT_OK) return rc;
    c.eps = eps;
    c.valid = true;
    return PT_OK;
}

struct SceneDeleter {
    void operator()(pt_scene *s) const { pt_scene_destroy(s); }
};
using ScenePtr = std::unique_ptr<pt_scene, SceneDeleter>;   // frees host and device side on every early return / exception

int finish_scene(ScenePtr s, int device, pt_scene **out) {
    pt_scene_host &h = *s->shared;
    if (h.host.n_tri() >= (1 << 23))   // a first, cheap bound; what the layouts really hold is checked on the built hierarchy (get_cull)
        return fail(PT_ERR_INVALID_ARGUMENT, "more than 8 388 607 triangles");
    for (int m : h.host.tri_mat)
        if (m < 0 || m >= h.host.n_mat()) return fail(PT_ERR_INVALID_ARGUMENT, "triangle refers to material " + std::to_string(m));
    pt::build_device_tables(h.host, h.tables);
    if (device >= 0) {
        const int rc = upload(s.get(), device);
        if (rc != PT_OK) return rc;
    }
    *out = s.release();
    return PT_OK;
}

// The part of the kernel arguments that describes the scene (its tables for the current eps).
void fill_scene_args(const pt_scene *scene, float eps, pt::RenderArgs &a) {
    std::memset(&a, 0, sizeof a);
    const pt::CullTables &t = *scene->cull.host;
    const pt::CullConstants &cc = t.cc, &ca = t.cc_all;
    a.clusters = scene->cull.clusters;
    a.spheres = scene->cull.spheres;
    a.bary = scene->cull.bary;
    a.bary_all = scene->cull.bary_all;
    a.a_max_all = ca.a_max; a.m0_all = ca.m0; a.t_guard_all = ca.t_guard;
    a.exact = scene->d_exact;
    a.exact_slot = scene->cull.exact_slot;
    a.bvh = scene->cull.bvh;
    a.n_bvh = static_cast<uint32_t>(t.bvh.size());
    a.bvh_err = t.bvh_err;
    a.mats = scene->d_mats;
    a.n_mats = static_cast<int32_t>(scene->shared->tables.mats.size());
    a.sky = scene->d_sky;
    a.sky_w = scene->d_sky ? scene->sky_w : 0;
    a.sky_h = scene->d_sky ? scene->sky_h : 0;
    a.n_clusters = static_cast<int32_t>(t.clusters.size());
    a.n_tri = scene->shared->host.n_tri();
    a.big = t.big ? 1 : 0;
    a.n_slots = static_cast<uint32_t>(t.slot_tri.size());
    a.eps = eps;
    a.k1 = cc.k1; a.k2 = cc.k2; a.a_max = cc.a_max; a.m0 = cc.m0; a.m0_quad = cc.m0_quad; a.t_guard = cc.t_guard;
    a.r_org = t.r_org;
    a.may_leave_envelope = t.may_leave_envelope ? 1 : 0;
    a.last_segment_filter = 1;
#ifdef PT_TEST_HOOKS
    if (pt::g_cull_mutation.no_last_segment_filter) a.last_segment_filter = 0;
#endif
    a.regen_min_dead = 64;   // (set per launch from -MRR: enqueue_render)
    a.emis_clusters = t.emis_clusters;
    a.emis_large_w0 = t.emis_large_w0;
    a.emis_bvh = t.emis_bvh ? 1u : 0u;
}

void zero_stats(const pt_scene *scene, pt_render_stats *stats) {
    std::memset(stats, 0, sizeof *stats);
    stats->n_triangles = scene->shared->host.n_tri();
}

// Enqueue one integrator launch on `stream` for the context `ctx` (its scheduler words, its statistics block).  Never waits
// for the device except to grow the scheduler words.  The caller holds ctx.mutex.
int enqueue_render(pt_scene *scene, LaunchCtx &ctx, const pt_render_params *p, float *d_sum, float *d_sum2, int32_t *d_count,
                   hipStream_t stream, bool want_stats) {
    ctx.stats_pending = false;
    ctx.last_chunks = 0;
    PT_HIP_TRY(hipSetDevice(scene->device));
    std::lock_guard<std::mutex> launch_lock(scene->launch_mutex);
    const int crc = ensure_cull(scene, p->eps);
    if (crc != PT_OK) return crc;
    pt::RenderArgs a;
    fill_scene_args(scene, p->eps, a);
    a.vec_ok = (p->width % 4 == 0) &&
               ((reinterpret_cast<uintptr_t>(d_sum) | reinterpret_cast<uintptr_t>(d_sum2) | reinterpret_cast<uintptr_t>(d_count)) % 16 == 0);
    a.sum = d_sum;
    a.sum2 = d_sum2;
    a.count = d_count;
    a.width = p->width; a.height = p->height; a.row_begin = p->row_begin; a.row_end = p->row_end;
    a.row_stride = std::max(1, p->row_stride);
    a.band_rows = ptc::band_rows(p);
    a.regen_min_dead = pt::regen_min_dead_for(p->max_ray_reflections);
#ifdef PT_TEST_HOOKS
    if (g_regen_min_dead > 0) a.regen_min_dead = static_cast<uint32_t>(g_regen_min_dead);
#endif
    a.pass_begin = p->pass_begin; a.pass_count = p->pass_count; a.mrr = p->max_ray_reflections;
    a.error = p->error; a.seed = p->seed;
    if (p->row_end == p->row_begin) return PT_OK;
    int rc = ctx_ready(ctx);
    if (rc != PT_OK) return rc;
    if (want_stats) {
        if ((rc = ctx_stats_ready(ctx)) != PT_OK) return rc;
        a.stats = ctx.d_stats;
    }
    // one wave = one tile of 8 rows; how many pixels wide depends on the instantiation this launch runs
#ifdef PT_TEST_HOOKS
    pt::integrator_plan_tiles(a, scene->cu_count, g_force_tile_width);
#else
    pt::integrator_plan_tiles(a, scene->cu_count);
#endif
    const uint32_t n_tiles = a.n_tiles;
    // Scheduler: cut the pass range into chunks so that the tail of the launch is balanced with small work items.  A tile's
    // chunks run in order and each re-reads and re-writes the tile's accumulators, so there should be few of them: chunk c
    // takes 3/4 of the passes that are left, down to single passes (256 passes: 192 + 48 + 12 + 3 + 1; until round 4 the last
    // chunk held 8 to 31 passes -- 192 + 48 + 16 -- and launches below 32 passes were one chunk: 16 passes at 1080p 5.26 -> 4.71 ms,
    // 64 passes 18.81 -> 18.20, 256 passes 72.70 -> 72.08, profiles/r04_ab_logs.txt chunks2).
    // Wave slots of the chip FOR THE INSTANTIATION THIS LAUNCH RUNS: its occupancy is the compiler's and the LDS budget's
    // business, asked from the runtime once per instantiation instead of assumed.
    int waves_per_cu = 24;
    PT_HIP_TRY(pt::integrator_waves_per_cu(a, &waves_per_cu));
    const uint32_t slots = static_cast<uint32_t>(scene->cu_count) * static_cast<uint32_t>(std::max(1, waves_per_cu));
    uint32_t n_chunks = 1;
    int32_t chunk_passes = 0;
    // (launches that fill a statistics block keep the old floor of 8: every work item ends with a dozen atomic adds to the same
    // few words, and 65 000 more items cost the 16-pass 1080p frame 6.7 -> 11.2 ms)
    // (and the regenerating kernels under a sky, where a chunk's end is a tail of idle lanes; the 8 x 8 kernel with its accumulators
    // in LDS stops at 2: chunks2)
    int chunk_min = (want_stats || a.sky != nullptr) ? 8 : a.narrow ? 2 : 1;
#ifdef PT_TEST_HOOKS
    if (g_chunk_min > 0) chunk_min = g_chunk_min;
#endif
    if (n_tiles >= slots / 2u)
        while (n_chunks < 6u && (p->pass_count >> (2u * n_chunks)) >= chunk_min) ++n_chunks;
    // Between about one and two tiles per wave slot the first of those chunks is too coarse -- all tiles' 3/4 of the passes: the
    // chip runs one full round of them and a second one half empty.  There the pass range is cut into EQUAL chunks, 8 to 32
    // work items per wave slot: Tor.obj 1366 x 768 x 256 spp 47.5 -> 40.2 ms, 960 x 540 25.7 -> 21.9 ms, and the 32 x 8 tiles of
    // adaptive 1080p launches (1.58 per slot) 65.2 -> 55.8 ms; from 2.3 tiles per slot up the 3/4 scheme wins again; the open
    // scene under a sky at 960 x 540 22.2 -> 20.1 ms (profiles/r04_ab_logs.txt, chunks1).
    int items_per_slot = 0;
    {
        const unsigned long long t100 = 100ull * n_tiles;
        if (a.sky != nullptr) {   // (regenerating kernels: a chunk's end is a tail of idle lanes, so fewer, longer chunks and a narrower range)
            if (t100 >= 75ull * slots && t100 < 190ull * slots) items_per_slot = 8;
        } else if (t100 >= 75ull * slots && t100 < 230ull * slots) {
            items_per_slot = t100 < 120ull * slots ? 8 : t100 < 190ull * slots ? 16 : 32;
        }
    }
#ifdef PT_TEST_HOOKS
    if (g_items_per_slot != 0) items_per_slot = std::max(0, g_items_per_slot);   // scheduler tuning, test build only (< 0: never equal chunks)
#endif
    if (items_per_slot > 0) {   // equal chunks, about items_per_slot work items per wave slot
        n_chunks = (static_cast<uint32_t>(items_per_slot) * slots + n_tiles - 1u) / n_tiles;
        n_chunks = std::max(1u, std::min(n_chunks, static_cast<uint32_t>(std::max(1, p->pass_count / 4))));
        chunk_passes = std::max(1, (p->pass_count + static_cast<int32_t>(n_chunks) - 1) / static_cast<int32_t>(n_chunks));
        n_chunks = static_cast<uint32_t>(std::max(1, (p->pass_count + chunk_passes - 1) / chunk_passes));
        if (n_chunks == 1u) chunk_passes = 0;
    }
    if (static_cast<unsigned long long>(n_tiles) * n_chunks > 0x7fffffffull) return fail(PT_ERR_INVALID_ARGUMENT, "too many work items");
    if (ctx.has_prev && ctx.prev_stream != stream) PT_HIP_TRY(hipStreamWaitEvent(stream, ctx.ev_done, 0));
    if (ctx.sched_words < 1 + static_cast<size_t>(n_tiles)) {
        if (ctx.has_prev) PT_HIP_TRY(hipStreamSynchronize(stream));   // an earlier launch of this context may still use the old words
        if (ctx.d_sched) (void)hipFree(ctx.d_sched);
        ctx.d_sched = nullptr;
        ctx.sched_words = 0;
        PT_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx.d_sched), (1 + static_cast<size_t>(n_tiles)) * sizeof(uint32_t)));
        ctx.sched_words = 1 + static_cast<size_t>(n_tiles);
    }
    PT_HIP_TRY(hipMemsetAsync(ctx.d_sched, 0, (1 + static_cast<size_t>(n_tiles)) * sizeof(uint32_t), stream));
    a.sched = ctx.d_sched;
    a.n_tiles = n_tiles;
    a.n_chunks = n_chunks;
    a.chunk_passes = chunk_passes;
    if (want_stats) {
        PT_HIP_TRY(hipMemsetAsync(ctx.d_stats, 0, 24 * sizeof(unsigned long long), stream));
        PT_HIP_TRY(hipEventRecord(ctx.ev0, stream));
    }
    PT_HIP_TRY(pt::launch_integrator(a, stream));
    if (want_stats) PT_HIP_TRY(hipEventRecord(ctx.ev1, stream));
    PT_HIP_TRY(hipEventRecord(ctx.ev_done, stream));
    ctx.has_prev = true;
    ctx.prev_stream = stream;
    ctx.last_chunks = n_chunks;
    ctx.stats_pending = want_stats;
    return PT_OK;
}

// Wait for the launch enqueue_render put on `stream` and read its statistics.  The caller holds ctx.mutex.
int collect_stats(pt_scene *scene, LaunchCtx &ctx, hipStream_t stream, pt_render_stats *stats) {
    zero_stats(scene, stats);
    if (!ctx.stats_pending) return PT_OK;   // an empty band
    ctx.stats_pending = false;
    PT_HIP_TRY(hipSetDevice(scene->device));
    unsigned long long h[24];
    PT_HIP_TRY(hipMemcpyAsync(h, ctx.d_stats, sizeof h, hipMemcpyDeviceToHost, stream));
    PT_HIP_TRY(hipStreamSynchronize(stream));
    float ms = -1.0f;
    PT_HIP_TRY(hipEventElapsedTime(&ms, ctx.ev0, ctx.ev1));
    stats->samples_traced = h[0];
    stats->segments = h[1];
    stats->contributing = h[2];
    stats->exact_tests = h[3];
    stats->misses = h[4];
    stats->wave_segments = h[5];
    stats->wave_node_rounds = h[6];
    stats->wave_exact_iterations = h[7];
    stats->kernel_ms = ms;
    stats->n_chunks = static_cast<int32_t>(ctx.last_chunks);
    stats->partial_commit_rounds = static_cast<int32_t>(std::min<unsigned long long>(h[8], 0x7fffffffull));
    stats->verify_checked = h[9];      // both stay 0 unless this is a verification build (-DPT_VERIFY_BRUTE / -DPT_VERIFY_SHIPPED)
    stats->verify_mismatches = h[10];
#ifdef PT_TEST_HOOKS
    if (h[10] != 0) {   // verification build: one disagreeing segment, for diagnosis
        auto f = [](unsigned long long w, int hi) { const uint32_t b = static_cast<uint32_t>(hi ? w >> 32 : w); float x; std::memcpy(&x, &b, 4); return x; };
        std::fprintf(stderr, "PT_VERIFY example: all-triangles loop -> triangle %d (key %016llx), culled search -> triangle %d (key %016llx); "
                             "ray o = (%.9g, %.9g, %.9g) d = (%.9g, %.9g, %.9g)\n",
                     static_cast<int>(h[11] & 0xFFFFFFFFu), h[11], static_cast<int>(h[12] & 0xFFFFFFFFu), h[12],
                     f(h[13], 1), f(h[13], 0), f(h[14], 1), f(h[14], 0), f(h[15], 1), f(h[15], 0));
    }
#endif
#ifdef PT_PHASE_TIMERS
    std::fprintf(stderr, "PT_PHASE_TIMERS cycles:");
    for (int k = 0; k < 8; ++k) std::fprintf(stderr, " %llu", h[16 + k]);
    std::fprintf(stderr, "\n");
#endif
    return PT_OK;
}

}  // namespace

namespace ptc {

// Rows the accumulator planes of a call hold: the band's rows, or with a row stride eight per tile row of the band.
int32_t band_rows(const pt_render_params *p) {
    const int32_t rows = p->row_end - p->row_begin;
    if (p->row_stride <= 1 || rows <= 0) return rows;
    const int32_t period = 8 * p->row_stride;
    return 8 * ((rows + period - 1) / period);
}

int check_params(const pt_scene *scene, const pt_render_params *p) {
    if (!scene || !p) return fail(PT_ERR_INVALID_ARGUMENT, "null scene or params");
    if (scene->device < 0) return fail(PT_ERR_NO_DEVICE, "scene was created without a device (device < 0)");
    if (p->width <= 0 || p->height <= 0) return fail(PT_ERR_INVALID_ARGUMENT, "width and height must be positive");
    if (p->row_begin < 0 || p->row_end > p->height || p->row_begin > p->row_end)
        return fail(PT_ERR_INVALID_ARGUMENT, "row band outside the image");
    if (p->row_stride < 0 || (p->row_stride > 1 && p->row_begin % 8 != 0))
        return fail(PT_ERR_INVALID_ARGUMENT, "row_stride must be 0 / 1 (contiguous rows) or n > 1 with row_begin a multiple of 8 (every n-th tile row of 8 rows)");
    if (p->pass_begin < 0 || p->pass_count < 0) return fail(PT_ERR_INVALID_ARGUMENT, "negative pass range");
    if (static_cast<long long>(p->width) * p->height > 0x7fffffffLL)
        return fail(PT_ERR_INVALID_ARGUMENT, "image has more than 2^31 pixels");
    if (p->rng_policy == PT_RNG_REFERENCE_STREAM)
        return fail(PT_ERR_UNSUPPORTED, "PT_RNG_REFERENCE_STREAM: the reference's serial minstd_rand0 streams (material.h:16-20) have no "
                                        "parallel evaluation order; the device implements PT_RNG_COUNTER");
    if (p->rng_policy != PT_RNG_COUNTER) return fail(PT_ERR_INVALID_ARGUMENT, "unknown rng_policy");
    return PT_OK;
}

int session_create_on(pt_scene *scene, int32_t width, int32_t height, int32_t row_begin, int32_t row_end, float *d_sum,
                      float *d_sum2, int32_t *d_count, pt_session **out, int32_t row_stride) {
    if (!out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    pt_render_params p;
    std::memset(&p, 0, sizeof p);
    p.width = width; p.height = height; p.row_begin = row_begin; p.row_end = row_end; p.row_stride = row_stride;
    const int rc = check_params(scene, &p);
    if (rc != PT_OK) return rc;
    PT_HIP_TRY(hipSetDevice(scene->device));
    std::unique_ptr<pt_session> s(new pt_session);
    s->scene = scene;
    s->width = width; s->height = height; s->row_begin = row_begin; s->row_end = row_end; s->row_stride = std::max(1, row_stride);
    s->n = static_cast<size_t>(band_rows(&p)) * width;
    PT_HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    hipError_t e = hipSuccess;
    if (d_sum) {   // borrowed planes: the caller zeroes and frees them
        s->d_sum = d_sum; s->d_sum2 = d_sum2; s->d_count = d_count;
    } else {
        const size_t plane_floats = (3 * s->n + 63) / 64 * 64;
        const size_t bytes = (2 * plane_floats + s->n) * sizeof(float) + 256;
        e = hipMalloc(reinterpret_cast<void **>(&s->d_band), bytes);
        if (e == hipSuccess) e = hipMemsetAsync(s->d_band, 0, bytes, s->stream);
        s->d_sum = s->d_band; s->d_sum2 = s->d_band + plane_floats;
        s->d_count = reinterpret_cast<int32_t *>(s->d_band + 2 * plane_floats);
    }
    if (e != hipSuccess) {
        if (s->d_band) (void)hipFree(s->d_band);
        (void)hipStreamDestroy(s->stream);
        return hip_fail(e, "pt_session_create");
    }
    *out = s.release();
    return PT_OK;
}

int session_enqueue(pt_session *s, const pt_render_params *p, bool want_stats) {
    if (!s || !p) return fail(PT_ERR_INVALID_ARGUMENT, "null session or params");
    if (p->width != s->width || p->height != s->height || p->row_begin != s->row_begin || p->row_end != s->row_end ||
        std::max(1, p->row_stride) != s->row_stride)
        return fail(PT_ERR_INVALID_ARGUMENT, "params describe another band than the session's");
    const int rc = check_params(s->scene, p);
    if (rc != PT_OK) return rc;
    return enqueue_render(s->scene, s->ctx, p, s->d_sum, s->d_sum2, s->d_count, s->stream, want_stats);
}

int session_collect(pt_session *s, pt_render_stats *stats) {
    if (!s || !stats) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    return collect_stats(s->scene, s->ctx, s->stream, stats);
}

}  // namespace ptc

using ptc::check_params;

extern "C" {

int pt_abi_version(void) { return PT_ABI_VERSION; }

void *pt_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        ptc::g_error = "pt_host_alloc: hipHostMalloc failed";
        return nullptr;
    }
    return p;
}

void pt_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int pt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *pt_last_error(void) { return ptc::g_error.c_str(); }

int pt_scene_timings(const pt_scene *scene, double *seconds) {
    if (!scene || !seconds) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(scene->shared->cull_mutex);
    seconds[0] = scene->shared->load_seconds;
    seconds[1] = scene->shared->cull_build_seconds;
    return PT_OK;
}

int pt_table_limits_check(uint64_t n_slots, uint64_t n_bvh_nodes, int32_t n_levels) {
    return guarded([&] { return check_table_limits(n_slots, n_bvh_nodes, n_levels); });
}
int pt_table_limits_check_tree(uint64_t n_slots, uint64_t n_bvh_nodes, int32_t n_levels, int32_t bvh_depth) {
    return guarded([&] { return check_table_limits(n_slots, n_bvh_nodes, n_levels, bvh_depth); });
}
static_assert(PT_MAX_BVH_DEPTH == pt::kMaxBvhDepth, "the ABI header states the box tree's depth limit");

int pt_scene_skybox_size(const pt_scene *scene, int32_t *width, int32_t *height) {
    if (!scene || !width || !height) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    *width = scene->sky ? scene->sky->w : 0;
    *height = scene->sky ? scene->sky->h : 0;
    return PT_OK;
}

static int scene_load_obj_impl(const char *model_dir, const char *model_name, int device, pt_scene **out) {
    if (!model_dir || !model_name || !out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    ScenePtr s(new pt_scene);
    s->shared = std::make_shared<pt_scene_host>();
    std::string err;
    bool io = false;
    const auto t0 = std::chrono::steady_clock::now();
    if (!pt::load_obj(model_dir, model_name, s->shared->host, err, io)) return fail(io ? PT_ERR_IO : PT_ERR_PARSE, err);
    // (finish_scene destroys the pt_scene on failure -- no device, bad ordinal, upload error --: keep the host side alive here)
    const std::shared_ptr<pt_scene_host> h = s->shared;
    const int rc = finish_scene(std::move(s), device, out);
    h->load_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();   // (includes the first device's upload)
    return rc;
}

static int scene_create_impl(const float *triangles, const int32_t *triangle_material, int32_t n_triangles, const float *materials,
                    int32_t n_materials, int device, pt_scene **out) {
    if (!out || n_triangles < 0 || n_materials < 0 || (n_triangles > 0 && (!triangles || !triangle_material)) ||
        (n_materials > 0 && !materials))
        return fail(PT_ERR_INVALID_ARGUMENT, "null table or negative count");
    *out = nullptr;
    ScenePtr s(new pt_scene);
    s->shared = std::make_shared<pt_scene_host>();
    pt::HostScene &h = s->shared->host;
    h.tri.assign(triangles, triangles + static_cast<size_t>(n_triangles) * PT_TRIANGLE_FLOATS);
    h.tri_mat.assign(triangle_material, triangle_material + n_triangles);
    h.mat.assign(materials, materials + static_cast<size_t>(n_materials) * PT_MATERIAL_FLOATS);
    return finish_scene(std::move(s), device, out);
}

static int scene_clone_impl(const pt_scene *src, int device, pt_scene **out) {
    if (!src || !out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    ScenePtr s(new pt_scene);
    s->shared = src->shared;   // parsed model, tables, hierarchies built so far
    s->sky = src->sky;         // the skybox of the handle the copy is made from
    if (device >= 0) {
        const int rc = upload(s.get(), device);
        if (rc != PT_OK) return rc;
    }
    *out = s.release();
    return PT_OK;
}

static int scene_set_skybox_bmp_impl(pt_scene *scene, const char *path) {
    if (!scene) return fail(PT_ERR_INVALID_ARGUMENT, "null scene");
    std::vector<uint8_t> texels;
    int w = 0, h = 0;
    if (path && *path) {
        // bitmap_image(filename) -> load_bitmap(), bitmap_image.hpp:1508-1603: the same checks, reported instead of printed
        FILE *f = std::fopen(path, "rb");
        if (!f) return fail(PT_ERR_IO, std::string("skybox: cannot open ") + path);
        uint8_t hdr[54];
        const bool got = std::fread(hdr, 1, 54, f) == 54;
        auto u16 = [&](int at) { return static_cast<uint32_t>(hdr[at]) | (static_cast<uint32_t>(hdr[at + 1]) << 8); };
        auto u32 = [&](int at) { return u16(at) | (u16(at + 2) << 16); };
        std::string why;
        if (!got) why = "file shorter than its headers";
        else if (u16(0) != 19778) why = "invalid type value " + std::to_string(u16(0)) + " expected 19778";
        else if (u16(28) != 24) why = "invalid bit depth " + std::to_string(u16(28)) + " expected 24";
        else if (u32(14) != 40) why = "invalid BIH size " + std::to_string(u32(14)) + " expected 40";
        if (why.empty()) {
            const uint32_t uw = u32(18), uh = u32(22);
            const uint32_t pad = (4u - (3u * uw) % 4u) % 4u;
            std::fseek(f, 0, SEEK_END);
            const size_t physical = static_cast<size_t>(std::ftell(f));
            const size_t logical = static_cast<size_t>(uh) * uw * 3 + static_cast<size_t>(uh) * pad + 54;
            if (physical != logical || uw == 0 || uh == 0) {
                why = "mismatch between logical (" + std::to_string(logical) + ") and physical (" + std::to_string(physical) + ") sizes";
            } else {
                texels.resize(static_cast<size_t>(uw) * uh * 3);
                std::fseek(f, 54, SEEK_SET);
                uint8_t padbuf[4];
                for (uint32_t i = 0; i < uh && why.empty(); ++i) {   // rows are stored bottom-up
                    if (std::fread(&texels[static_cast<size_t>(uh - i - 1) * uw * 3], 1, static_cast<size_t>(uw) * 3, f) != static_cast<size_t>(uw) * 3 ||
                        (pad && std::fread(padbuf, 1, pad, f) != pad))
                        why = "short read";
                }
                w = static_cast<int>(uw);
                h = static_cast<int>(uh);
            }
        }
        std::fclose(f);
        if (!why.empty()) return fail(PT_ERR_PARSE, std::string("skybox ") + path + ": " + why);
    }
    if (scene->device >= 0) {
        PT_HIP_TRY(hipSetDevice(scene->device));
        PT_HIP_TRY(hipDeviceSynchronize());
        if (scene->d_sky) {
            (void)hipFree(scene->d_sky);
            scene->d_sky = nullptr;
        }
        if (!texels.empty()) {
            PT_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&scene->d_sky), texels.size() + 64));
            PT_HIP_TRY(hipMemcpy(scene->d_sky, texels.data(), texels.size(), hipMemcpyHostToDevice));
        }
    }
    // (per-device copies made from THIS handle afterwards inherit the skybox; other handles of the same model keep theirs)
    scene->sky_w = w;
    scene->sky_h = h;
    if (texels.empty()) {
        scene->sky.reset();
    } else {
        auto sk = std::make_shared<pt_sky_texels>();
        sk->texels.swap(texels);
        sk->w = w;
        sk->h = h;
        scene->sky = std::move(sk);
    }
    return PT_OK;
}

int pt_scene_counts(const pt_scene *scene, int32_t *n_triangles, int32_t *n_materials) {
    if (!scene) return fail(PT_ERR_INVALID_ARGUMENT, "null scene");
    if (n_triangles) *n_triangles = scene->shared->host.n_tri();
    if (n_materials) *n_materials = scene->shared->host.n_mat();
    return PT_OK;
}

int pt_scene_get_triangles(const pt_scene *scene, float *triangles, int32_t *triangle_material) {
    if (!scene) return fail(PT_ERR_INVALID_ARGUMENT, "null scene");
    const pt::HostScene &h = scene->shared->host;
    if (triangles) std::memcpy(triangles, h.tri.data(), h.tri.size() * sizeof(float));
    if (triangle_material) std::memcpy(triangle_material, h.tri_mat.data(), h.tri_mat.size() * sizeof(int32_t));
    return PT_OK;
}

int pt_scene_get_materials(const pt_scene *scene, float *materials) {
    if (!scene || !materials) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    std::memcpy(materials, scene->shared->host.mat.data(), scene->shared->host.mat.size() * sizeof(float));
    return PT_OK;
}

void pt_scene_destroy(pt_scene *s) {
    if (!s) return;
    if (s->device >= 0) {
        (void)hipSetDevice(s->device);
        if (s->cull.clusters) (void)hipFree(s->cull.clusters);
        if (s->cull.spheres) (void)hipFree(s->cull.spheres);
        if (s->cull.bary) (void)hipFree(s->cull.bary);
        if (s->cull.bary_all) (void)hipFree(s->cull.bary_all);
        if (s->cull.exact_slot) (void)hipFree(s->cull.exact_slot);
        if (s->cull.bvh) (void)hipFree(s->cull.bvh);
        if (s->d_exact) (void)hipFree(s->d_exact);
        if (s->d_mats) (void)hipFree(s->d_mats);
        if (s->d_sky) (void)hipFree(s->d_sky);
        if (s->d_host_band) (void)hipFree(s->d_host_band);
        if (s->host_stream) (void)hipStreamDestroy(s->host_stream);
        ptc::ctx_destroy(s->ctx);
    }
    delete s;
}

static int render_device_impl(pt_scene *scene, const pt_render_params *p, float *d_sum, float *d_sum2, int32_t *d_count,
                     void *hip_stream, pt_render_stats *stats) {
    const int rc = check_params(scene, p);
    if (rc != PT_OK) return rc;
    if (!d_sum || !d_sum2 || !d_count) return fail(PT_ERR_INVALID_ARGUMENT, "null accumulator pointer");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    std::lock_guard<std::mutex> ctx_lock(scene->ctx.mutex);
    const int r = enqueue_render(scene, scene->ctx, p, d_sum, d_sum2, d_count, stream, stats != nullptr);
    if (r != PT_OK || !stats) return r;
    return collect_stats(scene, scene->ctx, stream, stats);
}

static int trace_rays_host_impl(pt_scene *scene, int32_t n_rays, const float *origins, const float *directions, float eps,
                       int32_t *hit_index, float *hit_t) {
    if (!scene) return fail(PT_ERR_INVALID_ARGUMENT, "null scene");
    if (scene->device < 0) return fail(PT_ERR_NO_DEVICE, "scene was created without a device (device < 0)");
    if (n_rays < 0 || (n_rays > 0 && (!origins || !directions || !hit_index || !hit_t)))
        return fail(PT_ERR_INVALID_ARGUMENT, "null ray buffer or negative count");
    if (n_rays == 0) return PT_OK;
    PT_HIP_TRY(hipSetDevice(scene->device));
    std::lock_guard<std::mutex> launch_lock(scene->launch_mutex);
    const int crc = ensure_cull(scene, eps);
    if (crc != PT_OK) return crc;
    pt::RenderArgs a;
    fill_scene_args(scene, eps, a);
    float *d_o = nullptr, *d_d = nullptr, *d_t = nullptr;
    int32_t *d_i = nullptr;
    const size_t n = static_cast<size_t>(n_rays);
    int result = PT_OK;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_o), n * 12);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_d), n * 12);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_t), n * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_i), n * 4);
    if (e == hipSuccess) e = hipMemcpy(d_o, origins, n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_d, directions, n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = pt::launch_trace_rays(a, d_o, d_d, n_rays, d_i, d_t, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(hit_index, d_i, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(hit_t, d_t, n * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) result = hip_fail(e, "pt_trace_rays_host");
    if (d_o) (void)hipFree(d_o);
    if (d_d) (void)hipFree(d_d);
    if (d_t) (void)hipFree(d_t);
    if (d_i) (void)hipFree(d_i);
    return result;
}

static int render_host_impl(pt_scene *scene, const pt_render_params *p, float *sum, float *sum2, int32_t *count,
                   pt_render_stats *stats) {
    const int rc = check_params(scene, p);
    if (rc != PT_OK) return rc;
    if (!sum || !sum2 || !count) return fail(PT_ERR_INVALID_ARGUMENT, "null accumulator pointer");
    PT_HIP_TRY(hipSetDevice(scene->device));
    const int rows = ptc::band_rows(p);
    const size_t W = static_cast<size_t>(p->width), n = static_cast<size_t>(rows) * W;
    if (n == 0) {
        if (stats) zero_stats(scene, stats);
        return PT_OK;
    }
    std::lock_guard<std::mutex> host_lock(scene->host_mutex);
    // device band (sum | sum2 | count planes, 256-byte aligned planes), grown on demand and kept
    const size_t plane = (3 * n + 63) / 64 * 64;
    if (scene->host_band_floats < 2 * plane + n) {
        if (scene->d_host_band) (void)hipFree(scene->d_host_band);
        scene->d_host_band = nullptr;
        scene->host_band_floats = 0;
        PT_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&scene->d_host_band), (2 * plane + n) * sizeof(float) + 256));
        scene->host_band_floats = 2 * plane + n;
    }
    if (!scene->host_stream) PT_HIP_TRY(hipStreamCreateWithFlags(&scene->host_stream, hipStreamNonBlocking));
    float *d_sum = scene->d_host_band, *d_sum2 = d_sum + plane;
    int32_t *d_count = reinterpret_cast<int32_t *>(d_sum2 + plane);
    // One launch for the whole band between plain synchronous copies.  (Measured and dropped: cutting the band into row
    // slabs so that copies overlap kernels.  A tile's passes run strictly in order, so every launch lasts at least one
    // tile's whole pass chain -- 13.6 ms at 256 spp whatever the slab's height -- and ten slabs took 189 ms where one
    // launch takes 96 ms + 3.4 ms of copies; profiles/r02_host_path_slabs.txt.  hipMemcpyAsync into pageable memory ran
    // at about 1 GB/s here, the synchronous call at PCIe speed.)
    PT_HIP_TRY(hipMemcpy(d_sum, sum, n * 12, hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemcpy(d_sum2, sum2, n * 12, hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemcpy(d_count, count, n * 4, hipMemcpyHostToDevice));
    const int r = render_device_impl(scene, p, d_sum, d_sum2, d_count, scene->host_stream, stats);
    if (r != PT_OK) return r;
    PT_HIP_TRY(hipStreamSynchronize(scene->host_stream));
    PT_HIP_TRY(hipMemcpy(sum, d_sum, n * 12, hipMemcpyDeviceToHost));
    PT_HIP_TRY(hipMemcpy(sum2, d_sum2, n * 12, hipMemcpyDeviceToHost));
    PT_HIP_TRY(hipMemcpy(count, d_count, n * 4, hipMemcpyDeviceToHost));
    return PT_OK;
}

static int session_render_impl(pt_session *s, const pt_render_params *p, pt_render_stats *stats) {
    if (!s) return fail(PT_ERR_INVALID_ARGUMENT, "null session or params");
    std::lock_guard<std::mutex> ctx_lock(s->ctx.mutex);
    const int r = ptc::session_enqueue(s, p, stats != nullptr);
    if (r != PT_OK || !stats) return r;
    return ptc::session_collect(s, stats);
}

static int session_read_impl(pt_session *s, float *sum, float *sum2, int32_t *count) {
    if (!s || !sum || !sum2 || !count) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (s->n == 0) return PT_OK;
    PT_HIP_TRY(hipSetDevice(s->scene->device));
    // wait for the session's kernels, then plain synchronous copies: the runtime's fast path for pageable destinations
    // (hipMemcpyAsync into pageable memory ran at about 1 GB/s here)
    PT_HIP_TRY(hipStreamSynchronize(s->stream));
    PT_HIP_TRY(hipMemcpy(sum, s->d_sum, 3 * s->n * sizeof(float), hipMemcpyDeviceToHost));
    PT_HIP_TRY(hipMemcpy(sum2, s->d_sum2, 3 * s->n * sizeof(float), hipMemcpyDeviceToHost));
    PT_HIP_TRY(hipMemcpy(count, s->d_count, s->n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return PT_OK;
}

static int session_clear_impl(pt_session *s) {
    if (!s) return fail(PT_ERR_INVALID_ARGUMENT, "null session");
    PT_HIP_TRY(hipSetDevice(s->scene->device));
    PT_HIP_TRY(hipMemsetAsync(s->d_sum, 0, 3 * s->n * sizeof(float), s->stream));
    PT_HIP_TRY(hipMemsetAsync(s->d_sum2, 0, 3 * s->n * sizeof(float), s->stream));
    PT_HIP_TRY(hipMemsetAsync(s->d_count, 0, s->n * sizeof(int32_t), s->stream));
    return PT_OK;
}


static int scene_cull_tables_impl(pt_scene *scene, float eps, int32_t *counts, float *clusters, float *spheres, float *bary,
                         float *constants) {
    if (!scene || !counts) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    std::shared_ptr<const pt::CullTables> tp;
    const int rc = get_cull(*scene->shared, eps, tp);
    if (rc != PT_OK) return rc;
    const pt::CullTables &t = *tp;
    counts[0] = static_cast<int32_t>(t.clusters.size());
    counts[1] = static_cast<int32_t>(t.spheres.size());
    counts[2] = static_cast<int32_t>(t.bary.size());
    int32_t large = 0;
    for (const auto &c : t.clusters)
        if (c.kind == 1) large += static_cast<int32_t>(c.n_tri);
    counts[3] = large;
    if (clusters) std::memcpy(clusters, t.clusters.data(), t.clusters.size() * sizeof(pt::ClusterDesc));
    if (spheres) std::memcpy(spheres, t.spheres.data(), t.spheres.size() * sizeof(pt::SphereRec));
    if (bary) std::memcpy(bary, t.bary.data(), t.bary.size() * sizeof(pt::CullRec));
    if (constants) {
        constants[0] = t.cc.k1; constants[1] = t.cc.k2; constants[2] = t.cc.a_max; constants[3] = t.cc.m0;
        constants[4] = t.cc.t_guard;
    }
    return PT_OK;
}

static int scene_cull_layout_impl(pt_scene *scene, float eps, int32_t *counts, int32_t *slot_triangle, void *bvh_nodes) {
    if (!scene || !counts) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    std::shared_ptr<const pt::CullTables> tp;
    const int rc = get_cull(*scene->shared, eps, tp);
    if (rc != PT_OK) return rc;
    const pt::CullTables &t = *tp;
    counts[0] = static_cast<int32_t>(t.slot_tri.size());
    counts[1] = static_cast<int32_t>(t.bvh.size());
    counts[2] = static_cast<int32_t>(t.bvh_inner);
    counts[3] = static_cast<int32_t>(t.clusters.size());
    if (slot_triangle)
        for (size_t k = 0; k < t.slot_tri.size(); ++k) slot_triangle[k] = t.slot_tri[k] == pt::kNoTriangle ? -1 : static_cast<int32_t>(t.slot_tri[k]);
    if (bvh_nodes && !t.bvh.empty()) std::memcpy(bvh_nodes, t.bvh.data(), t.bvh.size() * sizeof(pt::BvhNode));
    return PT_OK;
}

}  // extern "C"

// main.cpp:162-185 for the rows [y0, y1): per-pixel variance estimate d (written to contrib: d, or 1 for a pixel without samples,
// main.cpp:165-168), running max / min of d in row order, and the tonemapped pixel through `put`.
template <class Put>
static void resolve_rows(int32_t width, int y0, int y1, const float *sum, const float *sum2, const int32_t *count, float gamma,
                         float *contrib, float &max_d, float &min_d, Put &&put) {
    for (int y = y0; y < y1; ++y) {
        for (int x = 0; x < width; ++x) {
            const size_t p = static_cast<size_t>(y) * width + x;
            if (!count[p]) {   // main.cpp:165-168
                contrib[p] = 1.0f;
                put(p, nullptr);
                continue;
            }
            const float n = static_cast<float>(count[p]);
            float c[3], dd[3];
            for (int k = 0; k < 3; ++k) {
                const float mean = sum[3 * p + k] / n;
                dd[k] = sum2[3 * p + k] / n - mean * mean;
                c[k] = std::pow(sum[3 * p + k] / n, gamma) * 255.0f;   // main.cpp:179-182
            }
            const float d = dd[0] + dd[1] + dd[2];
            if (d > max_d) max_d = d;
            if (d < min_d) min_d = d;
            contrib[p] = d;
            put(p, c);
        }
    }
}

// The resolve is the reference's, value for value; only its schedule differs: bands of rows on the host's cores (powf per
// channel is 20 ms of one core at 1080p), then what depends on the pixel ORDER in order -- the bands' max / min combined first
// to last with the reference's own strict comparisons (ties, signed zeros: the first one in pixel order stays), and the float
// sum of the per-pixel terms as one sequential pass.
template <class Put>
static void resolve_all(int32_t width, int32_t height, const float *sum, const float *sum2, const int32_t *count, float gamma,
                        float *dispersion, Put &&put) {
    const size_t n_px = static_cast<size_t>(width) * height;
    std::vector<float> contrib(n_px);
    unsigned n_thr = n_px >= (1u << 18) ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    n_thr = std::min<unsigned>(n_thr, static_cast<unsigned>(height));
    std::vector<float> mx(n_thr, 0.0f), mn(n_thr, INFINITY);
    auto band = [&](unsigned t) {
        const int y0 = static_cast<int>(static_cast<long long>(height) * t / n_thr), y1 = static_cast<int>(static_cast<long long>(height) * (t + 1) / n_thr);
        resolve_rows(width, y0, y1, sum, sum2, count, gamma, contrib.data(), mx[t], mn[t], put);
    };
    if (n_thr == 1) {
        band(0);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < n_thr; ++t) th.emplace_back(band, t);
        band(0);
        for (auto &t : th) t.join();
    }
    float max_d = 0.0f, min_d = INFINITY, avg_d = 0.0f;
    for (unsigned t = 0; t < n_thr; ++t) {
        if (mx[t] > max_d) max_d = mx[t];
        if (mn[t] < min_d) min_d = mn[t];
    }
    for (size_t p = 0; p < n_px; ++p) avg_d += contrib[p];
    avg_d /= width * height;
    if (dispersion) {
        dispersion[0] = max_d;
        dispersion[1] = min_d;
        dispersion[2] = avg_d;
    }
}

extern "C" {

int pt_resolve(int32_t width, int32_t height, const float *sum, const float *sum2, const int32_t *count, float gamma,
               uint8_t *bgr, float *dispersion) {
    if (width <= 0 || height <= 0 || !sum || !sum2 || !count || !bgr)
        return fail(PT_ERR_INVALID_ARGUMENT, "null buffer or empty image");
    return guarded([&] {
        resolve_all(width, height, sum, sum2, count, gamma, dispersion, [bgr](size_t p, const float *c) {
            if (!c) {   // image.clear(), main.cpp:106: pixels without samples stay black
                bgr[3 * p + 0] = bgr[3 * p + 1] = bgr[3 * p + 2] = 0;
                return;
            }
            // set_pixel(x, y, float, float, float): float -> unsigned char (bitmap_image.hpp:194-206)
            bgr[3 * p + 0] = static_cast<uint8_t>(static_cast<int>(c[2]));
            bgr[3 * p + 1] = static_cast<uint8_t>(static_cast<int>(c[1]));
            bgr[3 * p + 2] = static_cast<uint8_t>(static_cast<int>(c[0]));
        });
        return static_cast<int>(PT_OK);
    });
}

int pt_resolve_float(int32_t width, int32_t height, const float *sum, const float *sum2, const int32_t *count, float gamma,
                     float *rgb, float *dispersion) {
    if (width <= 0 || height <= 0 || !sum || !sum2 || !count || !rgb)
        return fail(PT_ERR_INVALID_ARGUMENT, "null buffer or empty image");
    return guarded([&] {
        resolve_all(width, height, sum, sum2, count, gamma, dispersion, [rgb, sum](size_t p, const float *c) {
            // color_map keeps its raw sums where nothing was counted
            for (int k = 0; k < 3; ++k) rgb[3 * p + k] = c ? c[k] : sum[3 * p + k];
        });
        return static_cast<int>(PT_OK);
    });
}

static int post_filter_host_impl(int device, int32_t width, int32_t height, float *rgb, int32_t gauss, int32_t median) {
    if (width <= 0 || height <= 0 || !rgb) return fail(PT_ERR_INVALID_ARGUMENT, "null buffer or empty image");
    if (gauss < 0 || median < 0) return fail(PT_ERR_INVALID_ARGUMENT, "negative filter size");
    if (median * median / 2 > pt::kMedianMaxRank) return fail(PT_ERR_INVALID_ARGUMENT, "-MEDIAN window larger than 11 is not supported");
    if (!gauss && !median) return PT_OK;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev)
        return fail(PT_ERR_NO_DEVICE, "no usable HIP device for the post filters (there is no CPU fallback)");
    PT_HIP_TRY(hipSetDevice(device));
    const size_t bytes = static_cast<size_t>(width) * height * 3 * sizeof(float);
    float *d_a = nullptr, *d_b = nullptr, *d_w = nullptr;
    int result = PT_OK;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_a), bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_b), bytes);
    if (e == hipSuccess) e = hipMemcpy(d_a, rgb, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && gauss) {   // main.cpp:187-189
        const float r = static_cast<float>(gauss), pi = 3.141593f;
        const int rs = static_cast<int>(std::ceil(r * 2.57));
        const int side = 2 * rs + 1;
        std::vector<float> w(static_cast<size_t>(side) * side);
        for (int dy = -rs; dy <= rs; ++dy)
            for (int dx = -rs; dx <= rs; ++dx) {
                const int dsq = dx * dx + dy * dy;
                w[static_cast<size_t>(dy + rs) * side + (dx + rs)] = std::exp(-dsq / (2 * r * r)) / (pi * 2 * r * r);   // main.cpp:25
            }
        e = hipMalloc(reinterpret_cast<void **>(&d_w), w.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(d_w, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = pt::launch_gauss(d_a, d_b, d_w, width, height, rs, nullptr);
        std::swap(d_a, d_b);
    }
    if (e == hipSuccess && median) {   // main.cpp:190-192
        e = pt::launch_median(d_a, d_b, width, height, median, nullptr);
        std::swap(d_a, d_b);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(rgb, d_a, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) result = hip_fail(e, "pt_post_filter_host");
    if (d_a) (void)hipFree(d_a);
    if (d_b) (void)hipFree(d_b);
    if (d_w) (void)hipFree(d_w);
    return result;
}

int pt_quantize(int32_t width, int32_t height, const float *rgb, const int32_t *count, uint8_t *bgr) {
    if (width <= 0 || height <= 0 || !rgb || !count || !bgr) return fail(PT_ERR_INVALID_ARGUMENT, "null buffer or empty image");
    const size_t n = static_cast<size_t>(width) * height;
    std::memset(bgr, 0, n * 3);
    for (size_t p = 0; p < n; ++p) {   // main.cpp:193-201: only pixels with samples are written
        if (!count[p]) continue;
        bgr[3 * p + 0] = static_cast<uint8_t>(static_cast<int>(rgb[3 * p + 2]));
        bgr[3 * p + 1] = static_cast<uint8_t>(static_cast<int>(rgb[3 * p + 1]));
        bgr[3 * p + 2] = static_cast<uint8_t>(static_cast<int>(rgb[3 * p + 0]));
    }
    return PT_OK;
}

int pt_write_bmp(const char *path, int32_t width, int32_t height, const uint8_t *bgr) {
    if (!path || width <= 0 || height <= 0 || !bgr) return fail(PT_ERR_INVALID_ARGUMENT, "null argument or empty image");
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(PT_ERR_IO, std::string("cannot open ") + path + " for writing");
    const uint32_t row_bytes = static_cast<uint32_t>(width) * 3u;
    const uint32_t size_image = ((row_bytes + 3u) & 0x0000FFFCu) * static_cast<uint32_t>(height);   // sic: 16-bit mask
    uint8_t hdr[54];
    std::memset(hdr, 0, sizeof hdr);
    auto put32 = [&](int at, uint32_t v) { for (int i = 0; i < 4; ++i) hdr[at + i] = static_cast<uint8_t>(v >> (8 * i)); };
    auto put16 = [&](int at, uint16_t v) { hdr[at] = static_cast<uint8_t>(v); hdr[at + 1] = static_cast<uint8_t>(v >> 8); };
    put16(0, 19778);
    put32(2, 54u + size_image);
    put32(10, 54u);
    put32(14, 40u);
    put32(18, static_cast<uint32_t>(width));
    put32(22, static_cast<uint32_t>(height));
    put16(26, 1);
    put16(28, 24);
    put32(34, size_image);
    bool ok = std::fwrite(hdr, 1, sizeof hdr, f) == sizeof hdr;
    const uint32_t pad = (4u - row_bytes % 4u) % 4u;
    const uint8_t zeros[4] = {0, 0, 0, 0};
    for (int i = 0; ok && i < height; ++i) {
        ok = std::fwrite(bgr + static_cast<size_t>(height - i - 1) * row_bytes, 1, row_bytes, f) == row_bytes;
        if (ok && pad) ok = std::fwrite(zeros, 1, pad, f) == pad;
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? PT_OK : fail(PT_ERR_IO, std::string("short write to ") + path);
}

// ---- the guarded entry points (definitions above are the bodies) ----

int pt_scene_load_obj(const char *model_dir, const char *model_name, int device, pt_scene **out) {
    return guarded([&] { return scene_load_obj_impl(model_dir, model_name, device, out); });
}

int pt_scene_create(const float *triangles, const int32_t *triangle_material, int32_t n_triangles, const float *materials, int32_t n_materials, int device, pt_scene **out) {
    return guarded([&] { return scene_create_impl(triangles, triangle_material, n_triangles, materials, n_materials, device, out); });
}

int pt_scene_clone_to_device(const pt_scene *scene, int device, pt_scene **out) {
    return guarded([&] { return scene_clone_impl(scene, device, out); });
}

int pt_scene_set_skybox_bmp(pt_scene *scene, const char *path) {
    return guarded([&] { return scene_set_skybox_bmp_impl(scene, path); });
}

int pt_render_device(pt_scene *scene, const pt_render_params *p, float *d_sum, float *d_sum2, int32_t *d_count, void *hip_stream, pt_render_stats *stats) {
    return guarded([&] { return render_device_impl(scene, p, d_sum, d_sum2, d_count, hip_stream, stats); });
}

int pt_trace_rays_host(pt_scene *scene, int32_t n_rays, const float *origins, const float *directions, float eps, int32_t *hit_index, float *hit_t) {
    return guarded([&] { return trace_rays_host_impl(scene, n_rays, origins, directions, eps, hit_index, hit_t); });
}

int pt_render_host(pt_scene *scene, const pt_render_params *p, float *sum, float *sum2, int32_t *count, pt_render_stats *stats) {
    return guarded([&] { return render_host_impl(scene, p, sum, sum2, count, stats); });
}

int32_t pt_band_rows(const pt_render_params *params) { return params ? ptc::band_rows(params) : 0; }

int pt_session_create_strided(pt_scene *scene, int32_t width, int32_t height, int32_t row_begin, int32_t row_end, int32_t row_stride,
                              pt_session **out) {
    return guarded([&] { return ptc::session_create_on(scene, width, height, row_begin, row_end, nullptr, nullptr, nullptr, out, row_stride); });
}

int pt_session_create(pt_scene *scene, int32_t width, int32_t height, int32_t row_begin, int32_t row_end, pt_session **out) {
    return guarded([&] { return ptc::session_create_on(scene, width, height, row_begin, row_end, nullptr, nullptr, nullptr, out, 1); });
}

int pt_session_render(pt_session *session, const pt_render_params *params, pt_render_stats *stats) {
    return guarded([&] { return session_render_impl(session, params, stats); });
}

int pt_session_wait(pt_session *session) {
    return guarded([&] {
        if (!session) return fail(PT_ERR_INVALID_ARGUMENT, "null session");
        PT_HIP_TRY(hipSetDevice(session->scene->device));
        PT_HIP_TRY(hipStreamSynchronize(session->stream));
        return static_cast<int>(PT_OK);
    });
}

int pt_session_read(pt_session *session, float *sum, float *sum2, int32_t *count) {
    return guarded([&] { return session_read_impl(session, sum, sum2, count); });
}

int pt_session_clear(pt_session *session) {
    return guarded([&] { return session_clear_impl(session); });
}

void pt_session_destroy(pt_session *s) {
    if (!s) return;
    (void)hipSetDevice(s->scene->device);
    if (s->stream) {
        (void)hipStreamSynchronize(s->stream);
        (void)hipStreamDestroy(s->stream);
    }
    if (s->d_band) (void)hipFree(s->d_band);
    ptc::ctx_destroy(s->ctx);
    delete s;
}

int pt_scene_cull_tables(pt_scene *scene, float eps, int32_t *counts, float *clusters, float *spheres, float *bary, float *constants) {
    return guarded([&] { return scene_cull_tables_impl(scene, eps, counts, clusters, spheres, bary, constants); });
}

int pt_scene_cull_layout(pt_scene *scene, float eps, int32_t *counts, int32_t *slot_triangle, void *bvh_nodes) {
    return guarded([&] { return scene_cull_layout_impl(scene, eps, counts, slot_triangle, bvh_nodes); });
}

int pt_post_filter_host(int device, int32_t width, int32_t height, float *rgb, int32_t gauss, int32_t median) {
    return guarded([&] { return post_filter_host_impl(device, width, height, rgb, gauss, median); });
}

#ifdef PT_TEST_HOOKS
// Test build only (libpt_testhooks.so).  family: "sphere_r2", "m0", "k12", "a_max", "quad_slack" (scale on that family of
// conservative margins; 1 = as shipped), "no_absorb" (0/1), "reset".  Affects scenes whose cull tables are built afterwards.
// Scheduler / instantiation choices of launches enqueued afterwards: "items_per_slot" (n > 0: equal pass chunks, about n work items
// per wave slot; < 0: the 3/4 - of - the - rest chunks always; 0: the library's rule), "chunk_min" (the smallest last chunk of that
// scheme), "tile_width" (1: 8 x 8 tiles always; 2: 16 x 8 where the instantiation has them, batches over 16 x 8 tiles for adaptive
// launches; 3: the same with 32 x 8 batch tiles; 0: by tile count), "regen_min_dead" (path regeneration threshold).
// Test builds only: both forms of the box tree's child test (pt_kernels.hip: box_children_kept / box_children_kept_h) on n
// caller-supplied items -- nodes: n x 64 bytes (BvhNode), rays: n x 6 floats (origin, unit direction), t_best: n floats --
// out: 2 n masks (float form, half-precision form).  Host pointers; device 0.
int pt_test_box_masks(const void *nodes, const float *rays, const float *t_best, float err, int32_t n, uint32_t *out) {
    if (!nodes || !rays || !t_best || !out || n < 0) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return PT_OK;
    PT_HIP_TRY(hipSetDevice(0));
    void *d_nodes = nullptr, *d_rays = nullptr, *d_t = nullptr, *d_out = nullptr;
    const size_t nn = static_cast<size_t>(n);
    hipError_t e = hipMalloc(&d_nodes, nn * 64);
    if (e == hipSuccess) e = hipMalloc(&d_rays, nn * 24);
    if (e == hipSuccess) e = hipMalloc(&d_t, nn * 4);
    if (e == hipSuccess) e = hipMalloc(&d_out, nn * 8);
    if (e == hipSuccess) e = hipMemcpy(d_nodes, nodes, nn * 64, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, nn * 24, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_t, t_best, nn * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = pt::launch_box_masks(static_cast<const pt::BvhNode *>(d_nodes), static_cast<const float *>(d_rays), static_cast<const float *>(d_t), err, n,
                                                  static_cast<uint32_t *>(d_out), nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d_out, nn * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_nodes); (void)hipFree(d_rays); (void)hipFree(d_t); (void)hipFree(d_out);
    if (e != hipSuccess) return hip_fail(e, "pt_test_box_masks");
    return PT_OK;
}

int pt_test_set_mutation(const char *family, double value) {
    if (!family) return PT_ERR_INVALID_ARGUMENT;
    const std::string f = family;
    pt::CullMutation &m = pt::g_cull_mutation;
    if (f == "reset") { m = pt::CullMutation(); g_items_per_slot = 0; g_force_tile_width = 0; g_regen_min_dead = 0; g_chunk_min = 0; }
    else if (f == "regen_min_dead") g_regen_min_dead = static_cast<int>(value);
    else if (f == "chunk_min") g_chunk_min = static_cast<int>(value);
    else if (f == "sphere_r2") m.sphere_r2 = value;
    else if (f == "m0") m.m0 = value;
    else if (f == "k12") m.k12 = value;
    else if (f == "a_max") m.a_max = value;
    else if (f == "quad_slack") m.quad_slack = value;
    else if (f == "box") m.box = value;
    else if (f == "box_err") m.box_err = value;
    else if (f == "no_absorb") m.no_absorb = value != 0;
    else if (f == "no_last_segment_filter") m.no_last_segment_filter = value != 0;
    else if (f == "emis_drop") m.emis_drop = value != 0;
    else if (f == "order_mode") m.order_mode = static_cast<int>(value);
    else if (f == "bvh_fill") m.bvh_fill = value;
    else if (f == "bvh_mode") m.bvh_mode = static_cast<int>(value);
    else if (f == "bvh_depth_cap") m.bvh_depth_cap = static_cast<int>(value);
    else if (f == "big_threshold") m.big_threshold = static_cast<int>(value);
    else if (f == "max_clusters") m.max_clusters = static_cast<int>(value);
    else if (f == "items_per_slot") g_items_per_slot = static_cast<int>(value);
    else if (f == "tile_width") g_force_tile_width = static_cast<int>(value);
    else return fail(PT_ERR_INVALID_ARGUMENT, "unknown mutation family " + f);
    return PT_OK;
}
#endif

}  // extern "C"
