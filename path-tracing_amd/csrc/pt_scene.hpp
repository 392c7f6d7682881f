// Host-side scene: OBJ/MTL ingestion with the reference's token semantics and the tables the kernels read.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pt {

// What the reference keeps per triangle (triangles.h:19-25) and per material (material.h:22-28), flattened.
struct HostScene {
    std::vector<float> tri;        // n_tri * 14: plane[4], v0, v1, v2, square
    std::vector<int32_t> tri_mat;  // n_tri
    std::vector<float> mat;        // n_mat * 10: Kd, Ke, Ks, Ns
    int n_tri() const { return static_cast<int>(tri_mat.size()); }
    int n_mat() const { return static_cast<int>(mat.size() / 10); }
};

// Scene::LoadModel (scene.cpp:26-109).  Returns false and sets `err` on I/O or index errors
// (the reference exits or has undefined behaviour there).
bool load_obj(const std::string &model_dir, const std::string &model_name, HostScene &out, std::string &err, bool &io_error);

// Adds one triangle the way Triangle's constructor + SetNormal do (triangles.h:27-44).
void append_triangle(HostScene &s, const float v0[3], const float v1[3], const float v2[3], const float *vn_or_null,
                     int material);

// ---------------------------------------------------------------------------------------------------
// Device tables.
// ---------------------------------------------------------------------------------------------------
// Cull record: 12 floats per triangle, read wave-uniformly (scalar loads).  Rows are affine functions of a point:
//   plane(P) = n.P + w            (the reference's plane_, so t = -plane(o)/(n.d) is the reference's PlaneIntersect)
//   u(P), v(P)                    barycentric coordinates of P's orthogonal projection onto the triangle's own plane
struct CullRec {
    float n[3], w;
    float au[3], cu;
    float av[3], cv;
};
// Exact record: 16 floats, gathered per lane only for the few triangles that survive culling.  Two tables hold them:
// one in the ORIGINAL triangle order (shading, the reference's all-triangles loop) and one in SLOT order (below), where
// `orig` carries the original index for the (distance, index) tie-break of scene.cpp:116-120.
struct ExactRec {
    float plane[4];
    float v0[3], square;
    float v1[3];
    int32_t material;
    float v2[3];
    int32_t orig;
};
// Material record (Factory's lobe table, material.h:58-106).
struct MatRec {
    float kd[3], chance0;
    float ks[3], chance1;
    int32_t n_lobes, kind0, kind1, pad;   // kind: 0 emissive, 1 glossy, 2 diffuse
};

static const int kChunk = 32;    // large class: triangles per candidate-mask word
static const int kFan = 8;       // small class: children per node of the sphere tree
static const int kMaxLevels = 8; // small class: tree levels per cluster (8^8 triangles)
// Pixel tile of one wave: kTileW x kTileH = 64 pixels.
#ifndef PT_TILE_W
#define PT_TILE_W 8
#endif
static const int kTileW = PT_TILE_W, kTileH = 64 / PT_TILE_W;
// The box tree's depth is variable (build_bvh_sah).  The kernel's walk commits the top item of a full node stack whatever
// its children need (pt_kernels.hip, "Rare: not everything fits"): in that mode the stack holds a depth-first path, at most 7
// waiting siblings per inner level above kNodeStack, and the stack has 64 entries of slack: 7 x (depth - 1) <= 64 allows 10
// levels; 9 are allowed (same value as PT_MAX_BVH_DEPTH of the ABI header).  A deeper SAH tree is replaced by the uniform-depth
// tree (build_bvh: at most 8 levels below 2^24 triangles).
static const int kMaxBvhDepth = 9;
// Above kBigSceneTriangles triangles a scene gets ONE box tree over all its small triangles and the big-scene kernels (deep
// queues, pair pre-filter); up to it, sphere trees per connected group and the small-scene kernels.  The switch is made by the
// table builder (CullTables::big) -- the kernels and launchers only read the flag -- at the crossover measured on the MI355X
// (profiles/r04_t_sweep.jsonl: the torus 1 ... 32 times in the room through either path; 526 triangles: sphere trees 5 000
// against 3 910 Msamples/s, 1 038: 3 550 / 3 500, 1 550: 2 840 / 3 340, 2 062: 2 380 / 3 120 -- the constant was 2 048 until
// round 4, a 20 % step at the switch).  The small-scene kernels pack (original index, slot) into 16 bits each: never beyond
// kSmallSceneMaxTriangles whatever a test hook asks for.
static const int kBigSceneTriangles = 1024;
static const int kSmallSceneMaxTriangles = 16384;
static const int kMaxClusters = 8;            // small scenes: more connected groups than this are merged into one cluster

// SLOT ORDER.  The culling hierarchy does not follow the file order of the triangles: the table builder groups them
// spatially and lays them out in its own order; "slot" = position in that order (pairs, candidate masks and cluster
// ranges are all in slots).  Slots without a triangle (padding of the box tree's leaves) map to kNoTriangle.
static const uint32_t kNoTriangle = 0xFFFFFFFFu;

// Box tree of big scenes: one node = a frame (lower corner + one power-of-two step) and up to 8 children as 8-bit boxes in
// that frame.  64 bytes = four 16-byte loads per (ray, node) item.  All leaves are on one level: nodes [0, leaf0) are
// internal (children = nodes child_base + c), nodes [leaf0, n) are leaves (children = slots 8 * (node - leaf0) + c).
// Child boxes are rounded OUTWARD, so a dequantised box always contains the true one.
struct BvhNode {
    float org[3];
    uint32_t meta;       // biased exponent of the step (bits 0-7) | children - 1 (bits 8-10) | leaf (bit 11) | base (bits 12-31):
                         // inner node: index of its first child node (children are consecutive); leaf: first slot / 8
    uint8_t lo[3][8];    // per axis, per child: lower bound in steps from org
    uint8_t hi[3][8];    // upper bound
};
static_assert(sizeof(BvhNode) == 64, "BvhNode is four 16-byte loads");

// Bounding sphere used by the hierarchical cull: a ray is kept for the node iff its distance to `c` is <= sqrt(r2).
// r2 already contains every slack that makes the test conservative (see DESIGN.md "Culling").
struct SphereRec {
    float c[3], r2;
};
// A cluster = a maximal run of consecutive triangles (file order) of one class.
//   kind 0 (small triangles): an implicit 8-ary tree of bounding spheres over the run.  Level 0 = one sphere per
//       triangle, node j of level L covers triangles [8^L j, 8^L (j+1)); every level is padded to a multiple of 8
//       with spheres that keep nothing.  Level L starts at SphereRec index data_off + level_off[L] (level_off[0] = 0);
//       the top level (n_levels - 1) has at most 8 nodes.
//   kind 1 (large triangles): data_off indexes CullRec, one barycentric cull record per triangle, padded to 32.
//       level_off[w] is then the QUAD MASK of word w (w < 7): bit k set = slots k, k+1 hold one quad record (plane,
//       alpha row, beta row) for triangles k and k+1, two halves of a parallelogram in one stored plane.
struct ClusterDesc {
    float c[3], r2;                                  // bounding sphere of the whole run
    uint32_t first_tri, n_tri, kind, data_off;
    uint32_t n_levels;
    uint32_t level_off[kMaxLevels - 1];              // offsets of levels 1..7
};

// Margins of the barycentric cull test for one render call (depend on eps).
struct CullConstants {
    float k1, k2;     // |t_cull - t_reference| <= (k2 + k1*|t|) / |n.d|
    float a_max;      // scales a distance error into barycentric units
    float m0;         // barycentric slack an accepted point can have (eps / area, float error of the area sum)
    float m0_quad;    // m0 + the deviation of fused quads from exact parallelograms
    float t_guard;    // beyond this |t| the cull test abstains
};

// Everything the cull stage reads; depends on eps, so a scene caches one set per eps value.
struct CullTables {
    std::vector<uint32_t> slot_tri;  // slot -> original triangle index (kNoTriangle for padding slots)
    std::vector<ExactRec> exact_slot;// exact records in slot order
    std::vector<SphereRec> spheres;
    std::vector<CullRec> bary;
    std::vector<CullRec> bary_all;   // big scenes only: a barycentric record for EVERY slot (pair pre-filter)
    CullConstants cc_all;            // its margins (they must cover the smallest triangle of the scene)
    std::vector<ClusterDesc> clusters;
    std::vector<BvhNode> bvh;        // big scenes only: box tree over the small triangles (slots [0, 8 * leaves))
    uint32_t bvh_inner = 0;          // inner nodes of the box tree (the other nodes are leaves: 8 slots each)
    uint32_t bvh_depth = 0;          // levels of the box tree (root = 1, 0 = no tree); at most kMaxBvhDepth
    bool big = false;                // the scene takes the box-tree path (big-scene kernels)
    float bvh_err = 0;               // relative rounding allowance of the kernel's slab arithmetic
    CullConstants cc;
    float eps = 0;
    float r_org = 0;                 // the margins hold for ray origins with every |component| <= r_org
    bool may_leave_envelope = false; // some triangle can be "hit" at a point outside that envelope (near-degenerate triangles)
    // Where the emitters are (triangles whose material has an emissive lobe): a path's LAST segment can only contribute by
    // hitting one, so the integrator first searches among them alone (pt_kernels.hip: `emis_only`).
    uint32_t emis_clusters = 0xFFFFFFFFu;  // bit c: cluster c (< 32) holds an emitter; clusters >= 32 count as holding one
    uint32_t emis_large_w0 = 0xFFFFFFFFu;  // large class of at most 32 slots: its emitters (all ones if the class is larger)
    bool emis_bvh = true;                  // the box tree of a big scene holds an emitter
};

#ifdef PT_TEST_HOOKS
// Test build only (libpt_testhooks.so): scale factors on each family of conservative margins, so that the test suite can
// show it notices a cull that is too tight (tests/test_gpu_mutation.py).  The shipped library has no such knob.
struct CullMutation {
    double sphere_r2 = 1, m0 = 1, k12 = 1, a_max = 1, quad_slack = 1, box = 1, box_err = 1;
    int no_absorb = 0;
    int emis_drop = 0;                // table builder: 1 = the lowest emitter bit of the large class is cleared (a WRONG table: negative control)
    int no_last_segment_filter = 0;   // integrator: 1 = a path's last segment searches all triangles like every other segment
    double bvh_fill = 0.5;    // box tree: target fill of a node's children (builder tuning; uniform-depth builder)
    int bvh_depth_cap = kMaxBvhDepth;   // SAH trees deeper than this are replaced by the uniform-depth tree (tests raise it to see the refusal)
    int max_clusters = -1;    // small scenes: more connected groups than this are merged into one sphere tree (-1 = kMaxClusters)
    int big_threshold = -1;   // triangles above which a scene takes the box-tree path: -1 = the library's (kBigSceneTriangles)
    int bvh_mode = -1;        // box tree builder: -1 = the library's default, 0 = uniform depth, 1 = binary SAH collapsed to 8-wide nodes
    int order_mode = 0;   // small-scene clusters: 0 = cheaper of (cells, patches), 1 = as filed, 2 = cells, 3 = patches
};
extern CullMutation g_cull_mutation;
#define PT_MUT(field) (::pt::g_cull_mutation.field)
#else
#define PT_MUT(field) 1.0
#endif

struct DeviceTables {
    std::vector<ExactRec> exact;   // n_tri
    std::vector<MatRec> mats;      // n_mat
};

void build_device_tables(const HostScene &s, DeviceTables &out);
void build_cull_tables(const HostScene &s, float eps, CullTables &out);

}  // namespace pt
