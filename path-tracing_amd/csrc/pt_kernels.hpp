// Launch interface of the integrator kernel (pt_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "pt_scene.hpp"

namespace pt {

constexpr uint32_t kPhiloxKey1 = 0x50544831u;   // "PTH1": second Philox key word (the first is the seed)
// Path regeneration (pt_kernels.hip: REGEN; skybox instantiations): new paths are started once this many ray slots of the wave
// wait for one.  1 = at once; 64 = only when no ray of the wave is alive, i.e. the passes stay in step.  The crossover is
// between -MRR 3 and -MRR 5 (profiles/r04_regen_sweep.jsonl): short paths leave few lanes idle and the primary-ray code costs a
// third of a segment whenever it runs.  From -MRR 5 up: 4 -- on an open scene 1.3 % behind "at once" (7 517 / 7 617 Msamples/s), and
// a CLOSED room that happens to have a skybox (two thirds of a lane ends per iteration there) keeps its passes in step instead
// of running that code in every other iteration (5 753 against 5 287 at 1, 5 790 at 8; r04_ab_logs.txt regen3).
constexpr uint32_t regen_min_dead_for(int mrr) { return mrr >= 5 ? 4u : 64u; }

struct RenderArgs {
    const ClusterDesc *clusters;   // cull hierarchy (pt_scene.hpp: CullTables), read through the scalar cache
    const SphereRec *spheres;
    const CullRec *bary;
    const CullRec *bary_all;        // big scenes: one record per triangle for the pair pre-filter, else nullptr
    float a_max_all, m0_all, t_guard_all;
    const ExactRec *exact;    // n_tri records in the ORIGINAL triangle order: shading, the reference's all-triangles loop
    const ExactRec *exact_slot;   // the same records in slot order, `orig` = original index: the exact test of (ray, slot) pairs
    const BvhNode *bvh;       // big scenes: box tree over the small triangles, else nullptr
    uint32_t n_bvh;           // number of nodes
    float bvh_err;            // relative rounding allowance of the slab arithmetic
    const MatRec *mats;
    int32_t n_mats;
    const uint8_t *sky;       // skybox texels (B,G,R bytes, top-down rows) or nullptr
    int32_t sky_w, sky_h;
    float *sum, *sum2;        // row band, 3 floats per pixel
    int32_t *count;
    unsigned long long *stats;   // 24 counters (11 used; 16.. = phase timers of diagnostic builds) or nullptr
    int32_t n_clusters, n_tri;
    int32_t big;                 // CullTables::big: the hierarchy is the box tree (big-scene instantiations), not sphere trees
    uint32_t n_slots;            // slots of the hierarchy (>= n_tri: the box tree pads its leaves)
    int32_t width, height, row_begin, row_end;
    int32_t row_stride;                 // 1: the band is rows [row_begin, row_end); n > 1: every n-th tile row (8 image rows) from row_begin on, below row_end
    int32_t band_rows;                  // rows the band's planes hold (= row_end - row_begin without a stride; 8 per tile row with one)
    int32_t pass_begin, pass_count, mrr;
    float eps, error;
    uint32_t seed;
    float k1, k2, a_max, m0, m0_quad, t_guard;   // cull margins (pt_scene.hpp: CullConstants)
    int32_t blocks_x;                   // ceil(width / tile width of the instantiation launched)
    int32_t narrow;                     // 1: the statistics-free small-scene kernel runs its one-pixel-per-lane variant (8 x 8 tiles)
    int32_t adapt_pool;                 // 2 / 4: the adaptive-sampling instantiation over 16 x 8 / 32 x 8 tiles runs (batches); 0: none
    uint32_t *sched;                    // [0] ticket counter, [1 + tile] chunks of that tile already published; zeroed per launch
    uint32_t n_tiles, n_chunks;         // work items = n_tiles * n_chunks, chunk-major
    int32_t chunk_passes;               // passes per chunk; 0 = geometric chunks (see the kernel)
    int32_t vec_ok;                     // sum / sum2 / count are 16-byte aligned and width % 4 == 0: 16-byte write-back allowed
    float r_org;                        // origins with a component beyond this are outside the cull margins' envelope
    int32_t may_leave_envelope;         // 0: no triangle of this scene can be hit outside the envelope, the integrator skips the test
    // A path's last segment (depth + 1 == mrr) can only contribute by hitting an emitter: the statistics-free, skybox-free
    // instantiations search the emitters alone first (CullTables::emis_*) and run the full search only for rays that hit one.
    uint32_t regen_min_dead;            // skybox instantiations (path regeneration): new paths start once this many ray slots of the wave wait for one
    uint32_t last_segment_filter;       // 0 = off
    uint32_t emis_clusters, emis_large_w0, emis_bvh;
#ifdef PT_BLOCK_PROFILE
    uint32_t *blockprof;                // diagnostic build only (tools/asm_profile.py): execution counters of the instrumented code object
#endif
};

hipError_t launch_integrator(const RenderArgs &args, hipStream_t stream);
// waves of the instantiation such a launch runs that one compute unit holds at a time (runtime occupancy query, cached)
hipError_t integrator_waves_per_cu(const RenderArgs &args, int *waves);
// cuts the launch's row band (width, row_begin, row_end, band_rows, scene, stats already set) into the tiles of the instantiation it will
// run: fills narrow, adapt_pool, blocks_x and n_tiles
// (force: 0 = by tile count, 1 = always 8 x 8 tiles, 2 = always 16 x 8, 3 = always 16 x 8 and 32 x 8 with adaptive sampling on: test builds)
void integrator_plan_tiles(RenderArgs &args, int cu_count, int force = 0);
// diagnostic: both forms of the box tree's child test on (node, ray, t_best) items; out[2 i] = float form, out[2 i + 1] = half-precision form
hipError_t launch_box_masks(const BvhNode *d_nodes, const float *d_rays, const float *d_t_best, float err, int n, uint32_t *d_out, hipStream_t stream);
hipError_t launch_trace_rays(const RenderArgs &args, const float *d_origins, const float *d_directions, int n_rays,
                             int32_t *d_hit_index, float *d_hit_t, hipStream_t stream);

}  // namespace pt
