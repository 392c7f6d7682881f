// pt_frame_*: one image rendered by several GPUs of one node from ONE host program (include/pt_hip.h).
//
// The reference splits the rows of the image over its OpenMP threads inside main()'s pass loop (main.cpp:115,132,141:
// `#pragma omp parallel for` over y) and every thread writes its rows of the shared accumulators.  Here the "threads" are
// devices: the image's rows are split into bands, each band has a pt_session on its device (the scene's tables are copied to
// every device, the model is parsed and its hierarchy built once), a pass slice is enqueued on ALL devices before anything
// waits, and the accumulators of the bands are brought together on the root device by ONE RCCL group of send / receive pairs
// over xGMI: 28 bytes per pixel, three planes per band.  The counter RNG is keyed by the GLOBAL pixel index
// (pt_kernels.hip), so the frame is bit-identical to the one-device frame for any number of bands.
//
// The split is INTERLEAVED (round 4): band b of n is every n-th tile row of 8 image rows, starting with rows 8 b ... (one launch per
// band: pt_render_params::row_stride).  Contiguous bands -- the split of rounds 1-3, and what the reference's static OpenMP schedule
// gives its threads -- cost unequal amounts: on the 3840 x 2160 frame cut in four, the bands through the torus took 90 and 78 ms, the
// outer ones 67 and 68 (in eight: 33.7 ... 45.8 ms), and a frame is as slow as its slowest band; interleaved, every device sees the same
// mix of rows.  A band's planes hold its tile rows packed, so the gather moves each band to the root in one piece per plane (into
// a staging buffer there) and three strided device copies per band put the tile rows where they belong in the root's full-frame
// planes.  Images with fewer tile rows than bands keep contiguous bands (received straight into their rows; the root's own band then
// renders into the planes directly).
//
// RCCL is used directly (ncclCommInitAll / ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd), no framework in between.
// librccl.so is opened on first use: a one-device frame never loads it (it is a 570 MB library).
//
// PT_FRAME_REHEARSE lets several bands share a device -- the way to run the N-band code path on a one-GPU box -- and then
// the gather is NOT a collective: the same (source, destination, words) transfers are issued as device-to-device copies on
// the bands' streams.  pt_frame_info reports which transport a frame uses; nothing falls back silently.
#include "pt_capi_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>

using ptc::fail;
using ptc::guarded;
using ptc::hip_fail;

namespace {

// ---- RCCL, resolved at run time -------------------------------------------------------------------------------
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char *why = dlerror();
            r.error = std::string("cannot load librccl.so: ") + (why ? why : "?");
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.handle, n);
            if (!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + n;
            return p;
        };
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

int rccl_fail(ncclResult_t e, const char *what) {
    Rccl &r = rccl();
    return fail(PT_ERR_HIP, std::string(what) + ": RCCL: " + (r.GetErrorString ? r.GetErrorString(e) : "error"));
}
#define PT_RCCL_TRY(expr)                                    \
    do {                                                     \
        ncclResult_t e_ = (expr);                            \
        if (e_ != ncclSuccess) return rccl_fail(e_, #expr);  \
    } while (0)

}  // namespace

// One transfer of the gather: `words` 32-bit words of band `band` from its device buffer to the root's frame planes.
struct FrameXfer {
    int band;
    const void *src;
    void *dst;        // contiguous bands: the band's rows in the root's frame planes; interleaved: the band's staging plane on the root
    size_t words;
    // interleaved split: where the staged plane's tile rows go (strided copy on the root after the transfer); 0 = none
    void *scatter_dst = nullptr;      // row 8 b of the root plane
    size_t tile_row_bytes = 0;        // 8 rows of the plane
    size_t full_tile_rows = 0;        // tile rows copied whole
    size_t tail_bytes = 0;            // rows of the band's last tile row that lie inside the image (0: it is whole too)
    bool staged = false;              // the band is on another device: its plane arrives in the root's staging buffer first
    size_t stage_offset = 0;          // ... at this offset (floats)
};

struct pt_frame {
    int32_t width = 0, height = 0;
    uint32_t flags = 0;
    int transport = PT_FRAME_TRANSPORT_NONE;
    std::vector<int> band_device;          // per band
    std::vector<int32_t> band_rows;        // 2 per band: [begin, end) (interleaved split: [8 b, height), every row_stride-th tile row)
    int32_t row_stride = 1;                // 1: contiguous bands; n > 1: interleaved tile rows
    float *d_staging = nullptr;            // root device, interleaved split: the bands' planes as they arrive, before the scatter
    std::vector<int> devices;              // distinct devices, devices[0] = root
    std::vector<pt_scene *> scenes;        // one per distinct device (owned)
    std::vector<int> band_scene;           // band -> index into scenes / devices
    std::vector<pt_session *> sessions;    // per band (owned)
    std::vector<char> band_on_root_planes; // the band renders straight into the root's frame planes
    float *d_frame = nullptr;              // root device: sum[3 W H] | sum2[3 W H] | count[W H], planes 256-byte aligned
    size_t plane_floats = 0;
    std::vector<FrameXfer> xfers;
    std::vector<ncclComm_t> comms;         // RCCL transport: one communicator per distinct device, rank = index
    hipStream_t gather_stream = nullptr;   // root device, RCCL transport only: the receives of the gather
    bool dirty = false;                    // a band was rendered or cleared since the last gather
    std::vector<float> band_kernel_ms;     // per band: kernel time of the last pt_frame_render that asked for statistics (-1 = none yet)
    size_t n_px() const { return static_cast<size_t>(width) * height; }
    float *root_sum() const { return d_frame; }
    float *root_sum2() const { return d_frame + plane_floats; }
    int32_t *root_count() const { return reinterpret_cast<int32_t *>(d_frame + 2 * plane_floats); }
};

namespace {

// Contiguous rows [r0, r1) of band `b` of `n`: bands differ by at most one row and cover the image exactly (the static
// schedule of the reference's `omp parallel for` over y gives its threads the same kind of split).
void band_rows_of(int32_t height, int n, int b, int32_t &r0, int32_t &r1) {
    const int32_t base = height / n, extra = height % n;
    r0 = b * base + std::min<int32_t>(b, extra);
    r1 = r0 + base + (b < extra ? 1 : 0);
}

int frame_create_impl(const pt_scene *scene, const int32_t *devices, int32_t n_bands, int32_t width, int32_t height, uint32_t flags,
                      pt_frame **out) {
    if (!out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (!scene || !devices || n_bands < 1) return fail(PT_ERR_INVALID_ARGUMENT, "null scene / device list or no band");
    if (width <= 0 || height <= 0 || static_cast<long long>(width) * height > 0x7fffffffLL)
        return fail(PT_ERR_INVALID_ARGUMENT, "bad image size");
    if (n_bands > height) return fail(PT_ERR_INVALID_ARGUMENT, "more bands than rows");
    if (flags & ~(PT_FRAME_REHEARSE | PT_FRAME_SELF_COLLECTIVE)) return fail(PT_ERR_INVALID_ARGUMENT, "unknown frame flag");
    const bool rehearse = (flags & PT_FRAME_REHEARSE) != 0, self_coll = (flags & PT_FRAME_SELF_COLLECTIVE) != 0;
    if (rehearse && self_coll) return fail(PT_ERR_INVALID_ARGUMENT, "PT_FRAME_REHEARSE and PT_FRAME_SELF_COLLECTIVE exclude each other");
    if (self_coll && n_bands != 1) return fail(PT_ERR_INVALID_ARGUMENT, "PT_FRAME_SELF_COLLECTIVE takes exactly one band");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(PT_ERR_NO_DEVICE, "no HIP device is visible: the integrator has no CPU fallback");
    struct FrameDeleter {
        void operator()(pt_frame *f) const { pt_frame_destroy(f); }
    };
    std::unique_ptr<pt_frame, FrameDeleter> f(new pt_frame);
    f->width = width; f->height = height; f->flags = flags;
    for (int b = 0; b < n_bands; ++b) {
        const int d = devices[b];
        if (d < 0 || d >= n_dev) return fail(PT_ERR_NO_DEVICE, "device ordinal " + std::to_string(d) + " out of range");
        auto it = std::find(f->devices.begin(), f->devices.end(), d);
        if (it != f->devices.end() && !rehearse)
            return fail(PT_ERR_INVALID_ARGUMENT, "device " + std::to_string(d) + " is named for two bands: the gather is a collective with one rank per "
                                                 "device; PT_FRAME_REHEARSE runs several bands per device with device copies instead");
        f->band_scene.push_back(static_cast<int>(it - f->devices.begin()));
        if (it == f->devices.end()) f->devices.push_back(d);
        f->band_device.push_back(d);
    }
    // the split: interleaved tile rows when every band gets at least one, contiguous bands otherwise (tiny images)
    const int32_t tile_rows = (height + 7) / 8;
    f->row_stride = (n_bands > 1 && tile_rows >= n_bands) ? n_bands : 1;
    for (int b = 0; b < n_bands; ++b) {
        int32_t r0, r1;
        if (f->row_stride > 1) { r0 = 8 * b; r1 = height; }
        else band_rows_of(height, n_bands, b, r0, r1);
        f->band_rows.push_back(r0);
        f->band_rows.push_back(r1);
    }
    f->transport = n_bands == 1 && !self_coll ? PT_FRAME_TRANSPORT_NONE : rehearse ? PT_FRAME_TRANSPORT_DEVICE_COPIES : PT_FRAME_TRANSPORT_RCCL;

    // the scene on every device: parsed once, hierarchy built once (shared host side), tables uploaded per device
    for (int d : f->devices) {
        pt_scene *s = nullptr;
        const int rc = pt_scene_clone_to_device(scene, d, &s);
        if (rc != PT_OK) return rc;
        f->scenes.push_back(s);
    }
    // the root's full-frame planes
    const int root = f->devices[0];
    PT_HIP_TRY(hipSetDevice(root));
    const size_t n = f->n_px();
    f->plane_floats = (3 * n + 63) / 64 * 64;
    const size_t bytes = (2 * f->plane_floats + n) * sizeof(float) + 256;
    PT_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&f->d_frame), bytes));
    PT_HIP_TRY(hipMemset(f->d_frame, 0, bytes));
    // sessions.  Band 0 (the root's) renders into the frame planes themselves; with the RCCL transport every other band is on
    // another device and has its own buffer; a rehearsal gives EVERY other band its own buffer, also on the root device, so
    // that the N-band gather really moves N - 1 bands.
    // (interleaved split: every band, the root's too, renders into packed planes of its own; bands on other devices are staged on
    // the root before their tile rows are put in place)
    const bool interleaved = f->row_stride > 1;
    size_t staging_floats = 0;
    for (int b = 0; b < n_bands; ++b) {
        const int32_t r0 = f->band_rows[2 * b], r1 = f->band_rows[2 * b + 1];
        const size_t first = static_cast<size_t>(r0) * width;
        const bool on_planes = !interleaved && b == 0 && !self_coll;
        pt_session *ses = nullptr;
        const int rc = on_planes ? ptc::session_create_on(f->scenes[f->band_scene[b]], width, height, r0, r1, f->root_sum() + 3 * first,
                                                          f->root_sum2() + 3 * first, f->root_count() + first, &ses)
                                 : ptc::session_create_on(f->scenes[f->band_scene[b]], width, height, r0, r1, nullptr, nullptr, nullptr, &ses,
                                                          f->row_stride);
        if (rc != PT_OK) return rc;
        f->sessions.push_back(ses);
        f->band_on_root_planes.push_back(on_planes ? 1 : 0);
        if (!on_planes && ses->n > 0) {
            f->xfers.push_back({b, ses->d_sum, f->root_sum() + 3 * first, 3 * ses->n});
            f->xfers.push_back({b, ses->d_sum2, f->root_sum2() + 3 * first, 3 * ses->n});
            f->xfers.push_back({b, ses->d_count, f->root_count() + first, ses->n});
            if (interleaved) {
                // band b holds tile rows b, b + n, ...: all whole except possibly the image's last one
                const size_t count = (static_cast<size_t>(tile_rows) - b + f->row_stride - 1) / f->row_stride;
                const bool has_last = (static_cast<size_t>(b) + (count - 1) * f->row_stride) == static_cast<size_t>(tile_rows) - 1;
                const size_t last_rows = static_cast<size_t>(height) - 8u * (static_cast<size_t>(tile_rows) - 1);
                const bool partial = has_last && last_rows < 8;
                const bool staged = f->band_device[b] != root;
                for (int k = 0; k < 3; ++k) {
                    FrameXfer &x = f->xfers[f->xfers.size() - 3 + k];
                    const size_t elem = k < 2 ? 12 : 4;      // bytes per pixel of the plane
                    x.scatter_dst = x.dst;                   // row 8 b of the root plane
                    x.tile_row_bytes = 8u * static_cast<size_t>(width) * elem;
                    x.full_tile_rows = partial ? count - 1 : count;
                    x.tail_bytes = partial ? last_rows * static_cast<size_t>(width) * elem : 0;
                    x.dst = nullptr;                         // staged bands: set once the staging buffer exists
                    x.staged = staged;
                    if (staged) {
                        x.stage_offset = staging_floats;
                        staging_floats += (x.words + 63) / 64 * 64;
                    }
                }
            }
        }
    }
    if (interleaved) {
        PT_HIP_TRY(hipSetDevice(root));
        if (staging_floats > 0) {
            PT_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&f->d_staging), staging_floats * sizeof(float)));
            for (FrameXfer &x : f->xfers)
                if (x.staged) x.dst = f->d_staging + x.stage_offset;
        }
        if (!f->gather_stream) PT_HIP_TRY(hipStreamCreateWithFlags(&f->gather_stream, hipStreamNonBlocking));   // staged bands are put in place on it
    }
    if (f->transport == PT_FRAME_TRANSPORT_RCCL) {
        PT_HIP_TRY(hipSetDevice(root));
        if (!f->gather_stream) PT_HIP_TRY(hipStreamCreateWithFlags(&f->gather_stream, hipStreamNonBlocking));   // the receives of the gather
        Rccl &r = rccl();
        if (!r.error.empty()) return fail(PT_ERR_UNSUPPORTED, r.error);
        f->comms.assign(f->devices.size(), nullptr);
        PT_RCCL_TRY(r.CommInitAll(f->comms.data(), static_cast<int>(f->devices.size()), f->devices.data()));
    }
    *out = f.release();
    return PT_OK;
}

int frame_render_impl(pt_frame *f, const pt_render_params *p, pt_render_stats *stats) {
    if (!f || !p) return fail(PT_ERR_INVALID_ARGUMENT, "null frame or params");
    if (p->width != f->width || p->height != f->height) return fail(PT_ERR_INVALID_ARGUMENT, "params describe another image than the frame's");
    if (p->row_begin != 0 || p->row_end != f->height)
        return fail(PT_ERR_INVALID_ARGUMENT, "a frame renders the whole image: row_begin / row_end must be 0 / height (the frame owns the split)");
    const size_t nb = f->sessions.size();
    // every band is enqueued -- on every device -- before anything waits
    std::vector<std::unique_lock<std::mutex>> locks;
    for (size_t b = 0; b < nb; ++b) locks.emplace_back(f->sessions[b]->ctx.mutex);
    f->dirty = true;
    for (size_t b = 0; b < nb; ++b) {
        pt_render_params bp = *p;
        bp.row_begin = f->band_rows[2 * b];
        bp.row_end = f->band_rows[2 * b + 1];
        bp.row_stride = f->row_stride;
        const int rc = ptc::session_enqueue(f->sessions[b], &bp, stats != nullptr);
        if (rc != PT_OK) return rc;
    }
    if (!stats) return PT_OK;
    std::memset(stats, 0, sizeof *stats);
    stats->kernel_ms = -1.0f;
    f->band_kernel_ms.assign(nb, -1.0f);
    for (size_t b = 0; b < nb; ++b) {
        pt_render_stats bs;
        const int rc = ptc::session_collect(f->sessions[b], &bs);
        if (rc != PT_OK) return rc;
        f->band_kernel_ms[b] = bs.kernel_ms;
        stats->samples_traced += bs.samples_traced; stats->segments += bs.segments; stats->contributing += bs.contributing;
        stats->exact_tests += bs.exact_tests; stats->misses += bs.misses; stats->wave_segments += bs.wave_segments;
        stats->wave_node_rounds += bs.wave_node_rounds; stats->wave_exact_iterations += bs.wave_exact_iterations;
        stats->partial_commit_rounds += bs.partial_commit_rounds;
        stats->verify_checked += bs.verify_checked; stats->verify_mismatches += bs.verify_mismatches;
        stats->kernel_ms = std::max(stats->kernel_ms, bs.kernel_ms);   // the bands run side by side: the frame's time is the slowest band's
        stats->n_chunks = std::max(stats->n_chunks, bs.n_chunks);
        stats->n_triangles = bs.n_triangles;
    }
    return PT_OK;
}

int frame_gather_impl(pt_frame *f) {
    if (!f) return fail(PT_ERR_INVALID_ARGUMENT, "null frame");
    // (`dirty` is cleared only once the whole gather has been enqueued: after a failed one pt_frame_read tries again instead of
    // handing out the root's stale rows)
    if (f->xfers.empty()) {
        f->dirty = false;
        return PT_OK;
    }
    const int root = f->devices[0];
    // interleaved split: a band's packed plane -> its tile rows in the root's plane (every row_stride-th), on `stream` of the root device
    auto scatter = [&](const FrameXfer &x, const void *plane, hipStream_t stream) -> int {
        const size_t pitch = static_cast<size_t>(f->row_stride) * x.tile_row_bytes;
        if (x.full_tile_rows > 0)
            PT_HIP_TRY(hipMemcpy2DAsync(x.scatter_dst, pitch, plane, x.tile_row_bytes, x.tile_row_bytes, x.full_tile_rows, hipMemcpyDeviceToDevice, stream));
        if (x.tail_bytes > 0)
            PT_HIP_TRY(hipMemcpyAsync(static_cast<char *>(x.scatter_dst) + x.full_tile_rows * pitch,
                                      static_cast<const char *>(plane) + x.full_tile_rows * x.tile_row_bytes, x.tail_bytes, hipMemcpyDeviceToDevice, stream));
        return PT_OK;
    };
    if (f->transport == PT_FRAME_TRANSPORT_DEVICE_COPIES) {
        // rehearsal: the collective's transfers as plain copies, each on its band's stream (after that band's kernels)
        for (const FrameXfer &x : f->xfers) {
            pt_session *s = f->sessions[x.band];
            PT_HIP_TRY(hipSetDevice(s->scene->device));
            if (x.scatter_dst) {
                if (s->scene->device == root) {
                    const int rc = scatter(x, x.src, s->stream);
                    if (rc != PT_OK) return rc;
                } else {   // staged on the root, then put in place there once the copy has arrived
                    PT_HIP_TRY(hipMemcpyPeerAsync(x.dst, root, x.src, s->scene->device, x.words * 4, s->stream));
                    PT_HIP_TRY(hipStreamSynchronize(s->stream));      // (a rehearsal across devices: not a path anything times)
                    PT_HIP_TRY(hipSetDevice(root));
                    const int rc = scatter(x, x.dst, f->gather_stream);
                    if (rc != PT_OK) return rc;
                }
            } else if (s->scene->device == root) {
                PT_HIP_TRY(hipMemcpyAsync(x.dst, x.src, x.words * 4, hipMemcpyDeviceToDevice, s->stream));
            } else {
                PT_HIP_TRY(hipMemcpyPeerAsync(x.dst, root, x.src, s->scene->device, x.words * 4, s->stream));
            }
        }
        f->dirty = false;
        return PT_OK;
    }
    // ONE group: every band's three planes, sent on the band's stream (so after its kernels) and received on the root's
    // gather stream straight into the rows they belong to.  Sends and receives of a pair of ranks match in issue order.
    Rccl &r = rccl();
    PT_RCCL_TRY(r.GroupStart());
    ncclResult_t first_error = ncclSuccess;
    for (const FrameXfer &x : f->xfers) {
        pt_session *s = f->sessions[x.band];
        const int rank = f->band_scene[x.band];
        if (x.scatter_dst && !x.dst) continue;       // interleaved split, a band on the root device: no transfer, only the scatter below
        ncclResult_t e = r.Send(x.src, x.words, ncclFloat32, 0, f->comms[rank], s->stream);
        if (e == ncclSuccess) e = r.Recv(x.dst, x.words, ncclFloat32, rank, f->comms[0], f->gather_stream);
        if (e != ncclSuccess && first_error == ncclSuccess) first_error = e;
    }
    const ncclResult_t ge = r.GroupEnd();
    if (first_error != ncclSuccess) return rccl_fail(first_error, "ncclSend / ncclRecv");
    if (ge != ncclSuccess) return rccl_fail(ge, "ncclGroupEnd");
    // interleaved split: the staged planes' tile rows into place -- on the gather stream, behind the receives; the root's own band from
    // its buffer on its own stream, behind its kernels
    for (const FrameXfer &x : f->xfers) {
        if (!x.scatter_dst) continue;
        PT_HIP_TRY(hipSetDevice(root));
        const int rc = x.dst ? scatter(x, x.dst, f->gather_stream) : scatter(x, x.src, f->sessions[x.band]->stream);
        if (rc != PT_OK) return rc;
    }
    f->dirty = false;
    return PT_OK;
}

int frame_wait_impl(pt_frame *f) {
    if (!f) return fail(PT_ERR_INVALID_ARGUMENT, "null frame");
    for (pt_session *s : f->sessions) {
        PT_HIP_TRY(hipSetDevice(s->scene->device));
        PT_HIP_TRY(hipStreamSynchronize(s->stream));
    }
    if (f->gather_stream) {
        PT_HIP_TRY(hipSetDevice(f->devices[0]));
        PT_HIP_TRY(hipStreamSynchronize(f->gather_stream));
    }
    return PT_OK;
}

int frame_read_impl(pt_frame *f, float *sum, float *sum2, int32_t *count) {
    if (!f || !sum || !sum2 || !count) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    int rc;
    if (f->dirty && (rc = frame_gather_impl(f)) != PT_OK) return rc;
    if ((rc = frame_wait_impl(f)) != PT_OK) return rc;
    const size_t n = f->n_px();
    PT_HIP_TRY(hipMemcpy(sum, f->root_sum(), 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    PT_HIP_TRY(hipMemcpy(sum2, f->root_sum2(), 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    PT_HIP_TRY(hipMemcpy(count, f->root_count(), n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return PT_OK;
}

int frame_clear_impl(pt_frame *f) {
    if (!f) return fail(PT_ERR_INVALID_ARGUMENT, "null frame");
    // No wait for a gather still in flight: every band clears only what it renders into, on its own stream -- after its own
    // sends (same stream) -- and the rows of the root's planes that the receives write belong to other bands; nobody clears
    // them, the next gather overwrites them.  So a loop of clear / render / gather keeps all devices busy while the previous
    // frame's bands are still on their way to the root.
    for (pt_session *s : f->sessions) {
        const int rc = pt_session_clear(s);
        if (rc != PT_OK) return rc;
    }
    f->dirty = true;
    return PT_OK;
}

}  // namespace

extern "C" {

int pt_rccl_available(int32_t *version) {
    return guarded([&] {
        Rccl &r = rccl();
        if (!r.error.empty()) return fail(PT_ERR_UNSUPPORTED, r.error);
        int v = 0;
        if (r.GetVersion(&v) != ncclSuccess) return fail(PT_ERR_UNSUPPORTED, "ncclGetVersion failed");
        if (version) *version = v;
        return static_cast<int>(PT_OK);
    });
}

int pt_frame_create(const pt_scene *scene, const int32_t *devices, int32_t n_bands, int32_t width, int32_t height, uint32_t flags,
                    pt_frame **out) {
    return guarded([&] { return frame_create_impl(scene, devices, n_bands, width, height, flags, out); });
}

int pt_frame_info(const pt_frame *f, int32_t *n_bands, int32_t *band_rows, int32_t *band_device, int32_t *transport) {
    if (!f) return fail(PT_ERR_INVALID_ARGUMENT, "null frame");
    if (n_bands) *n_bands = static_cast<int32_t>(f->sessions.size());
    if (band_rows) std::copy(f->band_rows.begin(), f->band_rows.end(), band_rows);
    if (band_device) std::copy(f->band_device.begin(), f->band_device.end(), band_device);
    if (transport) *transport = f->transport;
    return PT_OK;
}

int pt_frame_row_stride(const pt_frame *f, int32_t *row_stride) {
    if (!f || !row_stride) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    *row_stride = f->row_stride;
    return PT_OK;
}

int pt_frame_band_kernel_ms(const pt_frame *f, float *ms) {
    if (!f || !ms) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
    for (size_t b = 0; b < f->sessions.size(); ++b) ms[b] = b < f->band_kernel_ms.size() ? f->band_kernel_ms[b] : -1.0f;
    return PT_OK;
}

int pt_frame_render(pt_frame *f, const pt_render_params *params, pt_render_stats *stats) {
    return guarded([&] { return frame_render_impl(f, params, stats); });
}

int pt_frame_gather(pt_frame *f) {
    return guarded([&] { return frame_gather_impl(f); });
}

int pt_frame_wait(pt_frame *f) {
    return guarded([&] { return frame_wait_impl(f); });
}

int pt_frame_read(pt_frame *f, float *sum, float *sum2, int32_t *count) {
    return guarded([&] { return frame_read_impl(f, sum, sum2, count); });
}

int pt_frame_clear(pt_frame *f) {
    return guarded([&] { return frame_clear_impl(f); });
}

void pt_frame_destroy(pt_frame *f) {
    if (!f) return;
    // a gather may still be in flight: its sends sit on the bands' streams, its receives on the root's gather stream -- both are
    // drained before the communicators go
    if (f->gather_stream && !f->devices.empty()) {
        (void)hipSetDevice(f->devices[0]);
        (void)hipStreamSynchronize(f->gather_stream);
    }
    for (pt_session *s : f->sessions) pt_session_destroy(s);   // waits for the band's stream
    if (!f->comms.empty()) {
        Rccl &r = rccl();
        for (ncclComm_t c : f->comms)
            if (c && r.CommDestroy) (void)r.CommDestroy(c);
    }
    if (!f->devices.empty()) {
        (void)hipSetDevice(f->devices[0]);
        if (f->gather_stream) {
            (void)hipStreamSynchronize(f->gather_stream);
            (void)hipStreamDestroy(f->gather_stream);
        }
        if (f->d_frame) (void)hipFree(f->d_frame);
        if (f->d_staging) (void)hipFree(f->d_staging);
    }
    for (pt_scene *s : f->scenes) pt_scene_destroy(s);
    delete f;
}

}  // extern "C"
