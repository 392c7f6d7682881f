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
// Exact record: 16 floats, gathered per lane only for the few triangles that survive culling.
struct ExactRec {
    float plane[4];
    float v0[3], square;
    float v1[3];
    int32_t material;
    float v2[3], pad;
};
// Material record (Factory's lobe table, material.h:58-106).
struct MatRec {
    float kd[3], chance0;
    float ks[3], chance1;
    int32_t n_lobes, kind0, kind1, pad;   // kind: 0 emissive, 1 glossy, 2 diffuse
};

static const int kChunk = 32;   // triangles per candidate-mask word

// Geometry statistics from which the culling margins are derived for a given eps (see DESIGN.md "Culling").
struct CullGeometry {
    double r_max = 0;          // largest |coordinate| of any vertex or of the camera origin
    double a_max = 0;          // largest gradient of any barycentric function (1 / smallest triangle height)
    double inv_2s_max = 0;     // max over triangles of 1/(2*S)
    double diam2_2s_max = 0;   // max over triangles of diam^2/(2*S)
};

struct DeviceTables {
    std::vector<CullRec> cull;     // padded with zero records to a multiple of kChunk
    std::vector<ExactRec> exact;   // n_tri
    std::vector<MatRec> mats;      // n_mat
    CullGeometry geo;
};

void build_device_tables(const HostScene &s, DeviceTables &out);

// Margins of the conservative cull test for one render call (depend on eps).
struct CullConstants {
    float k1, k2;     // |t_cull - t_reference| <= (k2 + k1*|t|) / |n.d|
    float a_max;      // scales a distance error into barycentric units
    float m0;         // barycentric slack an accepted point can have (eps / area, float error of the area sum)
    float t_guard;    // beyond this |t| the cull test abstains
};
CullConstants cull_constants(const CullGeometry &g, float eps);

}  // namespace pt
