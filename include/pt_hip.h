/*
 * pt_hip.h -- C ABI of the MI355X radiance integrator (libpt_hip.so).
 *
 * The reference (Andareon/Path-Tracing) has no plugin/FFI interface; the seam this library cuts is the
 * pass loop of main() (main.cpp:110-160) expressed through the only object interface the reference has:
 *
 *   reference                                                     this ABI
 *   ------------------------------------------------------------  ---------------------------------------
 *   Scene::Scene(color_map&, color2_map&, samples_count&)          pt_scene_create / pt_scene_load_obj
 *     scene.h:17-18, scene.cpp:16-23                                 (accumulators are per-call arguments)
 *   void Scene::LoadModel(std::string)   scene.h:19, scene.cpp:26   pt_scene_load_obj
 *   Ray(begin,dir,depth,coords) + while(ray.IsValid())              pt_render_device / pt_render_host
 *     scene.TraceRay(ray)   main.cpp:116-140, scene.h:23, ray.h:21    (all passes x pixels x segments on the GPU)
 *   dispersion stats + tonemap + set_pixel   main.cpp:162-201       pt_resolve
 *   bitmap_image::save_image   bitmap_image.hpp:431-478             pt_write_bmp
 *   Config fields read by the path   config.h:16-29                 pt_render_params
 *
 * Conventions: plain C types only, caller owns every buffer it passes, every function returns a pt_status
 * (0 = ok) and never throws or exits; pt_last_error() returns the message of the calling thread's last failure.
 * Accumulators are row-major, pixel p = (y - row_begin) * width + x:
 *   sum  [3*p + c]  = sum of contributions        (color_map,     main.cpp:94)
 *   sum2 [3*p + c]  = sum of squared contributions (color2_map,    main.cpp:96)
 *   count[p]        = number of contributing paths (samples_count, main.cpp:98)
 * (the reference's [x][y] nesting is an artefact of vector<vector<>>, not a format).
 * A render call ADDS passes [pass_begin, pass_begin+pass_count) to the buffers it is given, so a frame can be
 * rendered in slices (previews, time limits: main.cpp:111-114,141-158) and an image in row bands (multi-GPU).
 * There is no CPU fallback: without a usable HIP device the render entry points fail with PT_ERR_NO_DEVICE.
 * Threads and streams: a pt_scene may be rendered from several host threads and on several streams.  Launches made through
 * pt_render_device / pt_render_host share the scene's scheduler state and are ordered on the device; every pt_session has
 * its own, so sessions of ONE scene (row bands of an image) run side by side, and launches of different scenes are
 * independent anyway.  Calls on one pt_session / pt_frame are serialised by the caller.  pt_scene_set_skybox_bmp and
 * pt_scene_destroy must not race with a render of the same scene.
 * Several GPUs: pt_frame_* (below) renders one image on the devices of one node from one host program -- row bands, one RCCL
 * group of sends / receives to the root -- the counterpart of the reference's `omp parallel for` over rows, main.cpp:115,132,141.
 */
#ifndef PT_HIP_H
#define PT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 5

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID_ARGUMENT = 1,
    PT_ERR_IO = 2,             /* scene.cpp:32-35: the reference prints and exit(1)s */
    PT_ERR_PARSE = 3,          /* malformed OBJ/MTL: undefined behaviour in the reference */
    PT_ERR_NO_DEVICE = 4,
    PT_ERR_HIP = 5,
    PT_ERR_OUT_OF_MEMORY = 6,
    PT_ERR_UNSUPPORTED = 7     /* a valid request this library does not implement (e.g. the reference's serial RNG streams) */
} pt_status;

typedef struct pt_scene pt_scene;   /* immutable after creation */

/* One triangle as the reference stores it (triangles.h:19-25): plane_[4], vertices_[3][3], square. */
#define PT_TRIANGLE_FLOATS 14
/* One material as parsed from the MTL (material.h:22-28): Kd[3], Ke[3], Ks[3], Ns. */
#define PT_MATERIAL_FLOATS 10

typedef struct pt_render_params {
    int32_t width, height;          /* --W / --H       config.h:16-17 */
    int32_t row_begin, row_end;     /* rows [row_begin,row_end) of the image are rendered; buffers hold only them */
    int32_t pass_begin, pass_count; /* passes (samples per pixel) to add: main.cpp:110 */
    int32_t max_ray_reflections;    /* -MRR            config.h:19 (maximum number of path SEGMENTS) */
    float eps;                      /* -EPS            config.h:22 */
    float error;                    /* -ERR            config.h:23 (adaptive sampling threshold; <0 disables) */
    uint32_t seed;                  /* -SEED           config.h:10 */
    int32_t rng_policy;             /* PT_RNG_COUNTER (0, the default of a zeroed struct) or PT_RNG_REFERENCE_STREAM */
    int32_t row_stride;             /* 0 or 1: the band is rows [row_begin, row_end).  n > 1: the band is every n-th TILE ROW of 8 image rows,
                                     * starting with rows row_begin .. row_begin + 7 (row_begin a multiple of 8), as far as they lie below
                                     * row_end; the buffers hold those tile rows packed, 8 rows each (pt_band_rows() rows in all; rows of the
                                     * last tile row at or beyond row_end are left untouched).  The interleaved split of a frame over the
                                     * devices of a node: device k of n takes row_begin = 8 k, row_stride = n, row_end = height -- every
                                     * device then sees the same mix of cheap and expensive rows (pt_frame_* does this itself). */
} pt_render_params;

/* Rows the sum / sum2 / count buffers of a call with these parameters hold: row_end - row_begin, or with a row stride 8 per tile row. */
int32_t pt_band_rows(const pt_render_params *params);

/* Random-number policies.
 * PT_RNG_COUNTER: stateless Philox4x32-10 keyed by (seed, global pixel, pass, segment) -- the policy the device
 *   implements; results are independent of tiling, banding and GPU count.
 * PT_RNG_REFERENCE_STREAM: the reference's two process-wide minstd_rand0 engines consumed in path order
 *   (material.h:16-20, main.cpp:91-92,126-128).  Draw k of the stream belongs to whichever path asks k-th, so every
 *   sample depends on all earlier ones: there is no parallel evaluation order, and the render entry points answer
 *   PT_ERR_UNSUPPORTED.  (The CPU oracle implements it for the reference's recorded md5s; tests/test_gpu_rng_policy.py
 *   states and checks the statistical tolerance between the two policies.) */
#define PT_RNG_COUNTER 0
#define PT_RNG_REFERENCE_STREAM 1

typedef struct pt_render_stats {
    uint64_t samples_traced;        /* primary rays generated (adaptive skips excluded) */
    uint64_t segments;              /* Scene::TraceRay equivalents */
    uint64_t contributing;          /* samples that reached an emitter */
    uint64_t exact_tests;           /* ray-triangle pairs that needed the reference's full arithmetic */
    uint64_t misses;                /* segments that found no triangle */
    uint64_t wave_segments;         /* wave-level: segment-loop iterations summed over wavefronts */
    uint64_t wave_node_rounds;      /* wave-level: rounds of the lane-balanced sphere-tree walk */
    uint64_t wave_exact_iterations; /* wave-level: rounds of the lane-balanced exact tests */
    float kernel_ms;                /* HIP-event time of the integrator kernel on the launch stream; <0 if not timed */
    int32_t n_triangles;
    int32_t n_chunks;               /* pass-range chunks per pixel tile in this launch (each reads + writes the tile once) */
    int32_t partial_commit_rounds;  /* wave-level: tree-walk rounds that could not commit all 64 lanes (queues full) */
    /* Verification build only (libpt_verify.so, -DPT_VERIFY_BRUTE; always 0 from the shipped library): after the culled
     * search every segment's ray is also run through Triangle::Intersect against ALL triangles (scene.cpp:116-120 as
     * written) on the device and the two closest hits are compared. */
    uint64_t verify_checked;        /* segments compared */
    uint64_t verify_mismatches;     /* segments whose (distance bits, triangle index) differed */
} pt_render_stats;

/* ---- scene ---------------------------------------------------------------------------------------- */

/* Parse an OBJ + its MTL with the reference's token-stream semantics (Scene::LoadModel, scene.cpp:26-109) and
 * upload the tables to `device` (HIP ordinal).  device < 0 builds a host-only scene (inspection, no rendering).
 * model_dir is Config::model_path (used as a prefix, e.g. "../models/"), model_name is Config::model_name. */
int pt_scene_load_obj(const char *model_dir, const char *model_name, int device, pt_scene **out);

/* Build a scene from already-prepared tables (same layout pt_scene_get_triangles returns). */
int pt_scene_create(const float *triangles, const int32_t *triangle_material, int32_t n_triangles,
                    const float *materials, int32_t n_materials, int device, pt_scene **out);

/* The same scene on another device (or host-only, device < 0): the parsed model, its tables and the culling hierarchies built
 * so far are SHARED with `scene` (reference-counted; either may be destroyed first), only the device copies are new.  The
 * copy inherits the skybox `scene` has at the time of the call. */
int pt_scene_clone_to_device(const pt_scene *scene, int device, pt_scene **out);

/* -SKYBOX (config.h:26, scene.cpp:20-22): load a 24-bit BMP with bitmap_image::load_bitmap's checks
 * (bitmap_image.hpp:1508-1603) as the scene's skybox; rays that hit nothing then add its bilinear sample to the
 * accumulators (scene.cpp:126-154).  NULL or "" removes the skybox.  Not to be called while a render is in flight.
 * A skybox belongs to the HANDLE it is set on (the reference's Scene holds its own bitmap, scene.h:14): copies made from this
 * handle afterwards (pt_scene_clone_to_device, pt_frame_create) inherit it, other copies of the same model keep theirs.  A
 * failed call leaves the handle's skybox as it was. */
int pt_scene_set_skybox_bmp(pt_scene *scene, const char *path);
/* Width and height of the handle's skybox, 0 x 0 if it has none. */
int pt_scene_skybox_size(const pt_scene *scene, int32_t *width, int32_t *height);

int pt_scene_counts(const pt_scene *scene, int32_t *n_triangles, int32_t *n_materials);
int pt_scene_get_triangles(const pt_scene *scene, float *triangles, int32_t *triangle_material);
int pt_scene_get_materials(const pt_scene *scene, float *materials);
void pt_scene_destroy(pt_scene *scene);

/* ---- the hot path --------------------------------------------------------------------------------- */

/* d_sum/d_sum2/d_count are DEVICE pointers on the scene's device, sized for the row band (any 4-byte alignment;
 * 16-byte-aligned planes with width % 4 == 0 are written back with 16-byte stores).  The kernel is enqueued on
 * `hip_stream` (a hipStream_t, NULL = the default stream) and the call returns without synchronising unless `stats`
 * is non-NULL (then it waits for the kernel and fills `stats`). */
int pt_render_device(pt_scene *scene, const pt_render_params *params, float *d_sum, float *d_sum2,
                     int32_t *d_count, void *hip_stream, pt_render_stats *stats);

/* Same, with HOST buffers (PCIe-inclusive convenience path): upload, one launch, download; the device band is kept by
 * the scene between calls.  A driver that adds many pass slices to one frame should use a pt_session instead, which
 * moves the accumulators only when it is read. */
int pt_render_host(pt_scene *scene, const pt_render_params *params, float *sum, float *sum2, int32_t *count,
                   pt_render_stats *stats);

/* A render session keeps one row band's accumulators ON THE DEVICE between calls: the progressive driver
 * (main.cpp:110-160: a preview every `update` passes, the -TL check before every pass) adds pass slices with
 * pt_session_render and reads the band back only when it needs a preview or the final image.
 * params->width/height/row_begin/row_end (and row_stride) must equal the session's; pass_begin/pass_count select the slice. */
typedef struct pt_session pt_session;
int pt_session_create(pt_scene *scene, int32_t width, int32_t height, int32_t row_begin, int32_t row_end,
                      pt_session **out);                       /* accumulators start at zero */
/* the same for an interleaved band (pt_render_params::row_stride; 0 / 1 = pt_session_create): params->row_stride must equal it */
int pt_session_create_strided(pt_scene *scene, int32_t width, int32_t height, int32_t row_begin, int32_t row_end, int32_t row_stride,
                              pt_session **out);
int pt_session_render(pt_session *session, const pt_render_params *params, pt_render_stats *stats);
int pt_session_wait(pt_session *session);                                           /* until every slice queued so far is done */
int pt_session_read(pt_session *session, float *sum, float *sum2, int32_t *count);   /* waits, then copies out */
int pt_session_clear(pt_session *session);
void pt_session_destroy(pt_session *session);

/* ---- one image on several GPUs --------------------------------------------------------------------- */

/* The reference splits the image's rows over its OpenMP threads inside the pass loop (main.cpp:115,132,141); a pt_frame splits
 * them over DEVICES: n_bands bands (band b on devices[b]), every band a pt_session on its device.  The split is INTERLEAVED: band
 * b is every n_bands-th tile row of 8 image rows from rows 8 b on (pt_render_params::row_stride), so that every device renders
 * the same mix of cheap and expensive rows -- contiguous bands of the 3840 x 2160 Tor.obj frame cost 67 ... 90 ms in four and
 * 34 ... 46 ms in eight, and a frame is as slow as its slowest band.  (An image with fewer tile rows than bands keeps contiguous
 * bands, which differ by at most one row.)  pt_frame_render enqueues a pass slice on every device before it waits for anything;
 * pt_frame_gather brings the bands' accumulators (28 bytes per pixel) to the root device devices[0] with ONE RCCL group of
 * ncclSend / ncclRecv pairs -- each band's planes in one piece, into a staging buffer on the root, from where three strided device
 * copies per band put the tile rows in place in the root's full-frame planes (contiguous bands are received straight into their
 * rows, and the root's own band renders into the planes directly); pt_frame_read copies the full frame to the host, where the
 * unmodified pt_resolve runs.  The result is bit-identical to the one-device frame for any n_bands (the RNG is keyed by the global
 * pixel index).
 * `scene` is only read (any device, or host-only): the frame makes its own per-device copies, which share the parsed
 * model and the hierarchy.
 * flags: PT_FRAME_REHEARSE -- devices[] may name a device several times (the N-band code path on a one-GPU box); the
 *   gather is then NOT a collective but the same transfers as device-to-device copies (pt_frame_info: transport).
 *   Without it, two bands on one device are refused: nothing falls back silently.
 *   PT_FRAME_SELF_COLLECTIVE -- test aid: ONE band, rendered into a band buffer of its own and gathered to the frame planes
 *   of the same device by an RCCL send / receive to self, so that the collective path can be exercised on one device.
 * librccl.so is loaded on first use, by a frame with the RCCL transport (or pt_rccl_available); a one-band frame never loads it. */
typedef struct pt_frame pt_frame;
#define PT_FRAME_REHEARSE 1u
#define PT_FRAME_SELF_COLLECTIVE 2u
#define PT_FRAME_TRANSPORT_NONE 0            /* one band: it renders into the frame planes */
#define PT_FRAME_TRANSPORT_RCCL 1            /* one group of ncclSend / ncclRecv */
#define PT_FRAME_TRANSPORT_DEVICE_COPIES 2   /* rehearsal: hipMemcpyAsync / hipMemcpyPeerAsync */
int pt_frame_create(const pt_scene *scene, const int32_t *devices, int32_t n_bands, int32_t width, int32_t height,
                    uint32_t flags, pt_frame **out);                                  /* accumulators start at zero */
/* band_rows: 2 per band, [begin, end) (interleaved split: [8 b, height) -- of which the band holds every row_stride-th tile row);
 * band_device: 1 per band; any pointer may be NULL */
int pt_frame_info(const pt_frame *frame, int32_t *n_bands, int32_t *band_rows, int32_t *band_device, int32_t *transport);
/* 1: contiguous bands; n > 1: the interleaved split, band b = tile rows b, b + n, ... */
int pt_frame_row_stride(const pt_frame *frame, int32_t *row_stride);
/* params->width / height must equal the frame's, row_begin / row_end must be 0 / height (the frame owns the split);
 * pass_begin / pass_count select the slice.  Returns without waiting unless stats != NULL (then: sums over the bands,
 * kernel_ms = the slowest band's). */
int pt_frame_render(pt_frame *frame, const pt_render_params *params, pt_render_stats *stats);
/* Diagnosis of a multi-device run: the kernel time of every band (ms[n_bands], HIP events on the band's stream) of the last
 * pt_frame_render that asked for statistics, -1 where there is none.  stats->kernel_ms of that call is the slowest band's. */
int pt_frame_band_kernel_ms(const pt_frame *frame, float *ms);
int pt_frame_gather(pt_frame *frame);   /* enqueue the one collective of the frame; asynchronous */
int pt_frame_wait(pt_frame *frame);     /* until everything enqueued so far -- kernels and gather -- is done */
int pt_frame_read(pt_frame *frame, float *sum, float *sum2, int32_t *count);   /* gathers if a band changed since the last gather, waits, copies out */
int pt_frame_clear(pt_frame *frame);
void pt_frame_destroy(pt_frame *frame);
/* Can RCCL be loaded and does it export what the gather calls?  version = ncclGetVersion's.  Needs no GPU. */
int pt_rccl_available(int32_t *version);

/* Closest hit for caller-supplied rays: the triangle loop of Scene::TraceRay (scene.cpp:114-120) on the GPU.
 * origins/directions: 3 floats per ray (HOST buffers).  hit_index[i] = index of the accepted triangle with the
 * smallest distance (lowest index on ties) or -1, hit_t[i] = that distance (+inf on a miss).
 * The culling hierarchy's float-error margins are derived for the rays the integrator itself produces: unit
 * directions (normalised as Ray's constructor does, ray.h:23) and origins with max |component| <= max(20, largest
 * |vertex coordinate|) + 1 (the camera at (0,0,-20), or a point on a surface).  A ray outside that envelope
 * (| |d|^2 - 1 | > 1e-5, a farther origin) is answered by the reference's own loop over ALL triangles on the
 * device instead, so every finite ray gets the reference's answer; only the speed differs.  A ray with a non-finite
 * component misses (all its distances are NaN, see the deviation below).
 * eps < 0 is allowed and means what it means in the reference: the last test of Triangle::Intersect,
 * abs(..) > eps (triangles.h:68), then rejects every triangle, so every ray misses.
 * Known deviation: a ray lying EXACTLY in a triangle's stored plane makes PlaneIntersect (triangles.h:10-13) return
 * 0/0 = NaN, which passes every comparison of Triangle::Intersect -- the reference then reports that triangle wherever
 * it is.  No geometric cull can follow that; for such a ray this function returns the closest regular hit instead.
 * The integrator cannot produce such rays (DESIGN.md "Known deviation").  The same holds for a triangle whose stored
 * plane is itself NaN (three collinear vertices and no `vn`: normalize(0)): the reference reports it for every ray,
 * this library never does. */
int pt_trace_rays_host(pt_scene *scene, int32_t n_rays, const float *origins, const float *directions, float eps,
                       int32_t *hit_index, float *hit_t);

/* ---- resolve + image output (host side, as in the reference) --------------------------------------- */

/* main.cpp:162-201: per-pixel mean, gamma tonemap *255, float->uint8 truncation (bitmap_image.hpp:194-206),
 * dispersion statistics.  bgr: height*width*3 bytes, top-down rows, B,G,R order; pixels without samples stay 0.
 * dispersion[0..2] = max, min, average exactly as they are embedded in the reference's output file name. */
int pt_resolve(int32_t width, int32_t height, const float *sum, const float *sum2, const int32_t *count,
               float gamma, uint8_t *bgr, float *dispersion);

/* The same in three steps, for the optional post filters (-GAUSS / -MEDIAN, main.cpp:187-192):
 *   pt_resolve_float     main.cpp:162-185: statistics + the tonemapped FLOAT image (rgb: height*width*3, r,g,b order;
 *                        pixels without samples keep their raw sums, as color_map does)
 *   pt_post_filter_host  GaussBlur (main.cpp:11-33) if gauss != 0, then MedianFilter (main.cpp:49-80) if median != 0,
 *                        on HIP device `device`, in place on the host image; median <= 11
 *   pt_quantize          main.cpp:193-201: float -> uint8 truncation, only for pixels with samples */
int pt_resolve_float(int32_t width, int32_t height, const float *sum, const float *sum2, const int32_t *count,
                     float gamma, float *rgb, float *dispersion);
int pt_post_filter_host(int device, int32_t width, int32_t height, float *rgb, int32_t gauss, int32_t median);
int pt_quantize(int32_t width, int32_t height, const float *rgb, const int32_t *count, uint8_t *bgr);

/* bitmap_image::save_image (bitmap_image.hpp:431-478): 54-byte header, bottom-up rows padded to 4 bytes. */
int pt_write_bmp(const char *path, int32_t width, int32_t height, const uint8_t *bgr);

/* ---- diagnostics ---------------------------------------------------------------------------------- */

/* The culling hierarchy built for `eps` (host side; works on device < 0 scenes).  counts[4] = clusters, sphere
 * records, barycentric records, triangles handled by the barycentric class.  Pass NULL tables to query counts only.
 * clusters: 16 words each (centre[3], r2, then as uint32 bit patterns first_tri, n_tri, kind, data_off, n_levels,
 * level_off[7]: the sphere tree of a small-triangle cluster, see path-tracing_amd/csrc/pt_scene.hpp);
 * spheres: 4 floats each (centre[3], r2); bary: 12 floats each; constants: k1, k2, a_max, m0, t_guard. */
int pt_scene_cull_tables(pt_scene *scene, float eps, int32_t *counts, float *clusters, float *spheres, float *bary,
                         float *constants);

/* The order in which the hierarchy lists the triangles ("slots": the table builder groups triangles spatially, so the
 * hierarchy does not depend on the file order).  counts[4] = slots, box-tree nodes, inner nodes among them (the others are leaves of 8 slots each), clusters;
 * slot_triangle[k] = original triangle index of slot k, or -1 for a padding slot; bvh_nodes = the box tree of a big
 * scene, 64 bytes per node (path-tracing_amd/csrc/pt_scene.hpp: BvhNode).  Pass NULL to skip either.
 * In pt_scene_cull_tables the cluster fields first_tri / n_tri are slot ranges. */
int pt_scene_cull_layout(pt_scene *scene, float eps, int32_t *counts, int32_t *slot_triangle, void *bvh_nodes);

/* The kernels' tables pack indices into bit fields: (ray, slot) work items hold a slot in 24 bits; a box-tree node keeps its
 * base in 20 bits of BvhNode::meta (bits 12-31; bit 11 is the leaf flag) -- an inner node's first child, so fewer than 2^20
 * nodes, or a leaf's first slot / 8 in the GLOBAL slot order, so with a box tree fewer than 2^23 slots --; a sphere tree has at
 * most 8 levels; and the box tree, whose depth is variable, at most PT_MAX_BVH_DEPTH levels (the walk's stack slack is sized for
 * that).  A hierarchy beyond any of these is refused with PT_ERR_UNSUPPORTED when it is built (first render with an eps,
 * pt_scene_cull_tables / _layout), never truncated; this is the check itself, for counts.  bvh_depth = levels of the box tree
 * (root = 1; 0 = no box tree). */
#define PT_MAX_BVH_DEPTH 9
int pt_table_limits_check(uint64_t n_slots, uint64_t n_bvh_nodes, int32_t n_levels);
int pt_table_limits_check_tree(uint64_t n_slots, uint64_t n_bvh_nodes, int32_t n_levels, int32_t bvh_depth);

/* Host seconds spent on this scene's model so far: seconds[0] = parsing + per-triangle tables (pt_scene_load_obj /
 * pt_scene_create), seconds[1] = building culling hierarchies (one per eps, shared by all per-device copies). */
int pt_scene_timings(const pt_scene *scene, double *seconds);

/* Page-locked host memory for accumulator buffers: transfers to and from it run at PCIe speed without the runtime's
 * staging copies (a first pageable transfer in a fresh process cost 0.1 s for a 1080p band here).  Optional: every entry
 * point also accepts ordinary (pageable) memory.  Returns NULL on failure; pt_host_free(NULL) is a no-op. */
void *pt_host_alloc(size_t bytes);
void pt_host_free(void *p);

/* ---- misc ------------------------------------------------------------------------------------------- */
int pt_abi_version(void);
int pt_device_count(void);
const char *pt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PT_HIP_H */
