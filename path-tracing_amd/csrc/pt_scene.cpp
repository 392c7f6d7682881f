// Scene ingestion and device-table construction (host side, plain C++).
//
// Follows the behaviour of Scene::LoadModel (scene.cpp:26-109), Triangle's constructor and SetNormal
// (triangles.h:27-44) and Factory (material.h:58-106) of the reference; the code is new.
#include "pt_scene.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace pt {

namespace {

// GLM scalar semantics used at set-up time: dot3 = x+y+z of the products (left to right),
// normalize = v * (1/sqrt(dot)), cross as in glm/detail/func_geometric.inl.
inline float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

void set_plane_from_normal(float *rec, const float *normal) {   // Triangle::SetNormal, triangles.h:40-44
    const float inv = 1.0f / std::sqrt(dot3(normal, normal));
    const float n[3] = {normal[0] * inv, normal[1] * inv, normal[2] * inv};
    rec[0] = n[0];
    rec[1] = n[1];
    rec[2] = n[2];
    rec[3] = -dot3(n, rec + 4);
}

struct ObjIndex {
    int v = -1, vn = -1;
};

// "a/b/c" -> indices 0 and 2, each atoi()-1 (scene.cpp:6-14,90-96); missing fields give -1.
ObjIndex parse_face_group(const std::string &g) {
    ObjIndex r;
    const size_t s1 = g.find('/');
    r.v = std::atoi(g.substr(0, s1).c_str()) - 1;
    if (s1 != std::string::npos) {
        const size_t s2 = g.find('/', s1 + 1);
        if (s2 != std::string::npos) {
            const size_t s3 = g.find('/', s2 + 1);
            r.vn = std::atoi(g.substr(s2 + 1, s3 == std::string::npos ? std::string::npos : s3 - s2 - 1).c_str()) - 1;
        }
    }
    return r;
}

// The MTL reader of scene.cpp:45-71: every run of the outer loop appends one material, fields are picked out of
// a flat token stream, and the stream's eof flag (not its fail flag) ends both loops.
void read_mtl(std::istream &in, std::vector<float> &mat) {
    std::string tok = "1";
    while (!in.eof()) {
        float rec[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        while (!in.eof() && tok != "newmtl") in >> tok;
        in >> tok;   // the material's name; it then goes through the same dispatch as any other token
        while (!in.eof() && tok != "newmtl") {
            if (tok == "Kd") in >> rec[0] >> rec[1] >> rec[2];
            else if (tok == "Ke") in >> rec[3] >> rec[4] >> rec[5];
            else if (tok == "Ks") in >> rec[6] >> rec[7] >> rec[8];
            else if (tok == "Ns") in >> rec[9];
            in >> tok;
            if (in.fail() && !in.eof()) return;   // the reference would spin forever on a non-numeric field
        }
        mat.insert(mat.end(), rec, rec + 10);
    }
}

}  // namespace

void append_triangle(HostScene &s, const float v0[3], const float v1[3], const float v2[3], const float *vn, int material) {
    float rec[14];
    std::memcpy(rec + 4, v0, 12);
    std::memcpy(rec + 7, v1, 12);
    std::memcpy(rec + 10, v2, 12);
    const float ab[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    const float ac[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    const float c[3] = {ab[1] * ac[2] - ac[1] * ab[2], ab[2] * ac[0] - ac[2] * ab[0], ab[0] * ac[1] - ac[0] * ab[1]};
    set_plane_from_normal(rec, c);        // triangles.h:34
    rec[13] = std::sqrt(dot3(c, c));      // triangles.h:35
    if (vn) set_plane_from_normal(rec, vn);   // scene.cpp:102-104
    s.tri.insert(s.tri.end(), rec, rec + 14);
    s.tri_mat.push_back(material);
}

bool load_obj(const std::string &dir, const std::string &name, HostScene &out, std::string &err, bool &io_error) {
    io_error = false;
    std::ifstream obj(dir + name);
    if (!obj.is_open()) {
        err = "cannot open " + dir + name;
        io_error = true;
        return false;
    }
    std::vector<float> pos, nrm;
    int current_material = 0;
    std::string tok;
    while (!obj.eof()) {
        obj >> tok;
        if (obj.eof()) break;
        if (obj.fail()) {
            err = "unreadable token stream in " + name;
            return false;
        }
        if (tok == "mtllib") {
            std::string mtl_name;
            obj >> mtl_name;
            std::ifstream mtl(dir + mtl_name);
            if (!mtl.is_open()) {   // the reference loops forever here
                err = "cannot open " + dir + mtl_name;
                io_error = true;
                return false;
            }
            read_mtl(mtl, out.mat);
        } else if (tok == "v") {
            float p[3] = {0, 0, 0};
            obj >> p[0] >> p[1] >> p[2];
            pos.insert(pos.end(), p, p + 3);
        } else if (tok == "vt") {
            float uv[2];
            obj >> uv[0] >> uv[1];
        } else if (tok == "vn") {
            float n[3] = {0, 0, 0};
            obj >> n[0] >> n[1] >> n[2];
            nrm.insert(nrm.end(), n, n + 3);
        } else if (tok == "f") {
            ObjIndex idx[3];
            for (auto &g : idx) {
                std::string group;
                obj >> group;
                g = parse_face_group(group);
            }
            const int nv = static_cast<int>(pos.size() / 3), nn = static_cast<int>(nrm.size() / 3);
            for (const auto &g : idx)
                if (g.v < 0 || g.v >= nv) {
                    err = "face refers to vertex " + std::to_string(g.v + 1) + " of " + std::to_string(nv);
                    return false;
                }
            if (current_material < 0 || current_material >= out.n_mat()) {
                err = "usemtl " + std::to_string(current_material) + " with " + std::to_string(out.n_mat()) + " materials";
                return false;
            }
            if (idx[0].vn >= nn) {
                err = "face refers to normal " + std::to_string(idx[0].vn + 1) + " of " + std::to_string(nn);
                return false;
            }
            append_triangle(out, &pos[3 * idx[0].v], &pos[3 * idx[1].v], &pos[3 * idx[2].v],
                            idx[0].vn >= 0 ? &nrm[3 * idx[0].vn] : nullptr, current_material);
        } else if (tok == "usemtl") {
            obj >> current_material;   // scene.cpp:105-106: an int, used directly as the index
            if (obj.fail() && !obj.eof()) {
                err = "usemtl expects the integer index of a material (as the reference does)";
                return false;
            }
        }
        if (obj.fail() && !obj.eof()) {
            err = "malformed numeric field after '" + tok + "'";
            return false;
        }
    }
    return true;
}

void build_device_tables(const HostScene &s, DeviceTables &out) {
    const int T = s.n_tri();
    out.exact.resize(T);
    for (int i = 0; i < T; ++i) {
        const float *r = &s.tri[14 * static_cast<size_t>(i)];
        ExactRec &e = out.exact[i];
        std::memcpy(e.plane, r, 16);
        std::memcpy(e.v0, r + 4, 12);
        e.square = r[13];
        std::memcpy(e.v1, r + 7, 12);
        e.material = s.tri_mat[i];
        std::memcpy(e.v2, r + 10, 12);
        e.orig = i;
    }
    out.mats.resize(s.n_mat());
    for (int m = 0; m < s.n_mat(); ++m) {   // Factory, material.h:58-106
        const float *p = &s.mat[10 * static_cast<size_t>(m)];
        MatRec &d = out.mats[m];
        std::memset(&d, 0, sizeof d);
        std::memcpy(d.kd, p, 12);
        std::memcpy(d.ks, p + 6, 12);
        const float Ns = p[9];
        const bool ke = p[3] != 0.0f || p[4] != 0.0f || p[5] != 0.0f;
        const bool ks = p[6] != 0.0f || p[7] != 0.0f || p[8] != 0.0f;
        int n = 0;
        int kind[2] = {0, 0};
        float chance[2] = {0, 0};
        if (ke) {
            kind[n] = 0; chance[n] = 1.0f; ++n;
        } else {
            if (Ns != 0.0f && ks) { kind[n] = 1; chance[n] = Ns / 1000; ++n; }
            if (1 - Ns / 1000 > 0) { kind[n] = 2; chance[n] = 1 - Ns / 1000; ++n; }
        }
        d.n_lobes = n; d.kind0 = kind[0]; d.kind1 = kind[1];
        d.chance0 = chance[0]; d.chance1 = chance[1];
    }
}

#ifdef PT_TEST_HOOKS
CullMutation g_cull_mutation;
#endif

namespace {

struct V3 {
    double x, y, z;
};
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 crs(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dt(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double nrm(V3 a) { return std::sqrt(dt(a, a)); }

// What an accepted hit point of one triangle can be, in exact arithmetic (DESIGN.md "Culling"):
// barycentric coordinates >= -m_geo, distance from the triangle's own plane <= h_max.
struct TriGeo {
    V3 v[3];
    double area2;    // parallelogram area S
    double diam;     // longest edge
    double m_geo;    // (eps + E_fp) / (2 S)
    double h_max;    // sqrt(2 S E + E^2) / perimeter
    double a_max;    // largest barycentric gradient
    bool degenerate;
};

const double kU = 5.9604644775390625e-08;   // unit roundoff of binary32

TriGeo tri_geometry(const float *r, double eps) {
    TriGeo g;
    g.v[0] = {r[4], r[5], r[6]};
    g.v[1] = {r[7], r[8], r[9]};
    g.v[2] = {r[10], r[11], r[12]};
    const V3 e1 = sub(g.v[1], g.v[0]), e2 = sub(g.v[2], g.v[0]), e3 = sub(g.v[2], g.v[1]);
    g.area2 = nrm(crs(e1, e2));
    g.diam = std::max(nrm(e1), std::max(nrm(e2), nrm(e3)));
    const double perim = nrm(e1) + nrm(e2) + nrm(e3);
    // float error of the reference's |S - s1 - s2 - s3| for a point near the triangle: three cross products of
    // vectors no longer than ~diam, three lengths, three subtractions
    const double e_fp = 48.0 * kU * (g.diam + 1e-3) * (g.diam + 1e-3);   // first-order worst case 37 u, largest seen 32 u (tests/test_cull_margins_host.py)
    const double big_e = std::fabs(eps) + e_fp;
    // A triangle whose area is within a few eps of zero is accepted by the reference for points that have nothing to
    // do with it (all three computed sub-areas can vanish far away): never cull it.
    g.degenerate = !(g.area2 > 4.0 * big_e) || !std::isfinite(g.area2) || !std::isfinite(g.diam);
    if (g.degenerate) {
        g.m_geo = g.h_max = g.a_max = INFINITY;
    } else {
        g.m_geo = big_e / (2.0 * g.area2);
        g.h_max = std::sqrt(2.0 * g.area2 * big_e + big_e * big_e) / perim;
        g.a_max = g.diam / g.area2;   // gradients of the barycentric functions are 1/height; smallest height = S/diam
    }
    return g;
}

// Ritter's bounding sphere of a point set, then grown to cover every point exactly.
void bounding_sphere(const std::vector<V3> &pts, V3 &c, double &rad) {
    if (pts.empty()) { c = {0, 0, 0}; rad = 0; return; }
    auto far_from = [&](V3 p) {
        size_t best = 0; double bd = -1;
        for (size_t i = 0; i < pts.size(); ++i) { const double d = nrm(sub(pts[i], p)); if (d > bd) { bd = d; best = i; } }
        return best;
    };
    const V3 a = pts[far_from(pts[0])];
    const V3 b = pts[far_from(a)];
    c = {(a.x + b.x) / 2, (a.y + b.y) / 2, (a.z + b.z) / 2};
    rad = nrm(sub(a, b)) / 2;
    for (int it = 0; it < 2; ++it)
        for (const V3 &p : pts) {
            const V3 d = sub(p, c);
            const double dist = nrm(d);
            if (dist > rad) {
                const double nr = (rad + dist) / 2, k = (nr - rad) / dist;
                c = {c.x + d.x * k, c.y + d.y * k, c.z + d.z * k};
                rad = nr;
            }
        }
    for (const V3 &p : pts) rad = std::max(rad, nrm(sub(p, c)));
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// Slot order: which triangles are culled how, and in what order the tables list them
// ---------------------------------------------------------------------------------------------------
namespace {

struct Centroid {
    double c[3];
};

// Total order on triangles that depends on their GEOMETRY only (the file order enters last, for exact duplicates), so
// that a shuffled OBJ gives the same hierarchy.
struct GeoLess {
    const HostScene *s;
    const std::vector<Centroid> *cen;
    int axis;
    bool operator()(int a, int b) const {
        for (int k = 0; k < 3; ++k) {
            const double x = (*cen)[a].c[(axis + k) % 3], y = (*cen)[b].c[(axis + k) % 3];
            if (x != y) return x < y;
        }
        const int m = std::memcmp(&s->tri[14 * static_cast<size_t>(a)], &s->tri[14 * static_cast<size_t>(b)], 14 * sizeof(float));
        if (m != 0) return m < 0;
        return a < b;
    }
};

int longest_axis(const std::vector<int> &ids, size_t b, size_t e, const std::vector<Centroid> &cen) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = b; i < e; ++i)
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], cen[ids[i]].c[k]);
            hi[k] = std::max(hi[k], cen[ids[i]].c[k]);
        }
    int ax = 0;
    for (int k = 1; k < 3; ++k)
        if (hi[k] - lo[k] > hi[ax] - lo[ax]) ax = k;
    return ax;
}

// Reorders ids[b, e) into `sizes.size()` consecutive groups of the given sizes, each spatially compact: recursive
// bisection of the group list along the longest axis of the centroids.
void partition_groups(std::vector<int> &ids, size_t b, const std::vector<size_t> &sizes, size_t g0, size_t g1, const HostScene &s,
                      const std::vector<Centroid> &cen) {
    if (g1 - g0 <= 1) return;
    const size_t gm = g0 + (g1 - g0) / 2;
    size_t left = 0, total = 0;
    for (size_t g = g0; g < g1; ++g) {
        if (g < gm) left += sizes[g];
        total += sizes[g];
    }
    // the cut that leaves the two most compact halves: smallest sum of (bounding-sphere radius)^2 x triangles over the
    // three axes (a ray meets a sphere with probability ~ r^2)
    int best_axis = longest_axis(ids, b, b + total, cen);
    if (total <= 4096) {
        double best_cost = INFINITY;
        std::vector<int> tmp(ids.begin() + b, ids.begin() + b + total);
        for (int axis = 0; axis < 3; ++axis) {
            std::nth_element(tmp.begin(), tmp.begin() + left, tmp.end(), GeoLess{&s, &cen, axis});
            double cost = 0;
            for (int half = 0; half < 2; ++half) {
                const size_t h0 = half ? left : 0, h1 = half ? total : left;
                std::vector<V3> pts;
                for (size_t i = h0; i < h1; ++i)
                    for (int v = 0; v < 3; ++v) {
                        const float *p = &s.tri[14 * static_cast<size_t>(tmp[i]) + 4 + 3 * v];
                        pts.push_back({p[0], p[1], p[2]});
                    }
                std::sort(pts.begin(), pts.end(), [](const V3 &x, const V3 &y) { return x.x != y.x ? x.x < y.x : x.y != y.y ? x.y < y.y : x.z < y.z; });
                V3 c; double rad;
                bounding_sphere(pts, c, rad);
                cost += rad * rad * static_cast<double>(h1 - h0);
            }
            if (cost < best_cost) { best_cost = cost; best_axis = axis; }
        }
    }
    const GeoLess less{&s, &cen, best_axis};
    std::nth_element(ids.begin() + b, ids.begin() + b + left, ids.begin() + b + total, less);
    partition_groups(ids, b, sizes, g0, gm, s, cen);
    partition_groups(ids, b + left, sizes, gm, g1, s, cen);
}

// Surface-area-heuristic version for the box tree of big scenes: splits ids[b, e) into k consecutive children of AT MOST
// `cap` triangles each (sizes appended to `sizes`), by recursive bisection: each cut goes, along the best of the three
// axes, where area(left) * n_left + area(right) * n_right of the triangles' bounding boxes is smallest among the
// positions the capacities allow -- so children follow the objects of the scene (the gaps between them) instead of
// cutting through them at equal counts.
struct VBox {
    double lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; } }
    void add(const VBox &o) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], o.lo[k]); hi[k] = std::max(hi[k], o.hi[k]); } }
    double area() const {
        const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return x < 0 ? 0.0 : 2.0 * (x * y + y * z + z * x);
    }
};
void split_sah(std::vector<int> &ids, size_t b, size_t e, size_t k, size_t cap, const HostScene &s, const std::vector<Centroid> &cen,
               const std::vector<VBox> &tbox, std::vector<size_t> &sizes) {
    const size_t n = e - b;
    if (k <= 1) {
        sizes.push_back(n);
        return;
    }
    const size_t k1 = k / 2, k2 = k - k1;
    // n_left must leave no more than k2 * cap on the right, no more than k1 * cap on the left, and at least k1 / k2
    // triangles on each side (no empty child)
    const size_t lo = std::max(k1, n > k2 * cap ? n - k2 * cap : 0), hi = std::min(n - k2, k1 * cap);
    double best_cost = INFINITY;
    int best_axis = 0;
    size_t best_at = (lo + hi) / 2;
    std::vector<int> sorted(ids.begin() + b, ids.begin() + e), best_order;
    std::vector<double> right_area(n + 1);
    for (int axis = 0; axis < 3; ++axis) {
        const GeoLess less{&s, &cen, axis};
        std::sort(sorted.begin(), sorted.end(), less);
        VBox acc;
        acc.reset();
        right_area[n] = 0;
        for (size_t i = n; i-- > 0;) {
            acc.add(tbox[sorted[i]]);
            right_area[i] = acc.area();
        }
        acc.reset();
        for (size_t i = 1; i < n; ++i) {   // left = sorted[0, i)
            acc.add(tbox[sorted[i - 1]]);
            if (i < lo || i > hi) continue;
            const double cost = acc.area() * static_cast<double>(i) + right_area[i] * static_cast<double>(n - i);
            if (cost < best_cost) {
                best_cost = cost;
                best_axis = axis;
                best_at = i;
                (void)best_axis;
            }
        }
        if (best_axis == axis && best_cost < INFINITY) best_order = sorted;
    }
    if (!best_order.empty()) std::copy(best_order.begin(), best_order.end(), ids.begin() + b);
    split_sah(ids, b, b + best_at, k1, cap, s, cen, tbox, sizes);
    split_sah(ids, b + best_at, e, k2, cap, s, cen, tbox, sizes);
}

// Orders ids[b, e) so that every aligned run of 8^L consecutive entries (L = 1, 2, ...) is spatially compact: the
// implicit 8-ary sphere tree of a small-scene cluster is laid over this order.
void arrange_implicit(std::vector<int> &ids, size_t b, size_t e, const HostScene &s, const std::vector<Centroid> &cen) {
    const size_t n = e - b;
    if (n <= static_cast<size_t>(kFan)) {   // a leaf group: canonical order, whatever order the triangles arrived in
        std::sort(ids.begin() + b, ids.begin() + e, GeoLess{&s, &cen, 0});
        return;
    }
    size_t cap = kFan;   // capacity of one child subtree
    while (cap * kFan < n) cap *= kFan;
    std::vector<size_t> sizes;
    for (size_t left = n; left > 0; left -= std::min(left, cap)) sizes.push_back(std::min(left, cap));
    partition_groups(ids, b, sizes, 0, sizes.size(), s, cen);
    size_t at = b;
    for (size_t sz : sizes) {
        arrange_implicit(ids, at, at + sz, s, cen);
        at += sz;
    }
}

// Alternative arrangement for small clusters: compact PATCHES instead of axis-aligned cells.  Items (triangles, then
// groups of 8, then groups of 64, ...) are peeled off from the outside in: the item farthest from the centre of what is
// left seeds a group, its 7 nearest remaining items join it.  Surfaces (a torus, a sphere) pack tighter this way than
// under planar cuts.  Reorders ids[b, e); the last group of every level is the partial one, as the implicit tree needs.
void arrange_patches(std::vector<int> &ids, size_t b, size_t e, const HostScene &s, const std::vector<Centroid> &cen) {
    struct Item { std::vector<int> tris; double c[3]; };
    std::vector<Item> items;
    for (size_t i = b; i < e; ++i) items.push_back({{ids[i]}, {cen[ids[i]].c[0], cen[ids[i]].c[1], cen[ids[i]].c[2]}});
    const GeoLess less{&s, &cen, 0};
    while (items.size() > 1) {
        std::vector<uint8_t> used(items.size(), 0);
        std::vector<Item> next;
        size_t left = items.size();
        while (left > 0) {
            double m[3] = {0, 0, 0};
            for (size_t i = 0; i < items.size(); ++i)
                if (!used[i]) for (int k = 0; k < 3; ++k) m[k] += items[i].c[k] / static_cast<double>(left);
            auto d2 = [](const double *p, const double *q) { return (p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]) + (p[2] - q[2]) * (p[2] - q[2]); };
            // the seed: farthest from the centre of the remaining items (geometry breaks ties, not the file order)
            size_t seed = items.size();
            for (size_t i = 0; i < items.size(); ++i) {
                if (used[i]) continue;
                if (seed == items.size()) { seed = i; continue; }
                const double a = d2(items[i].c, m), z = d2(items[seed].c, m);
                if (a > z || (a == z && less(items[i].tris[0], items[seed].tris[0]))) seed = i;
            }
            const size_t take = std::min<size_t>(kFan, left);
            // grow the group by the item that enlarges its bounding sphere least (on a curved surface that follows the
            // curvature -- a half ring of a tube fits a smaller sphere than a flat-looking patch of the same area)
            std::vector<size_t> member = {seed};
            used[seed] = 1;
            auto verts_of = [&](const Item &it, std::vector<V3> &out) {
                for (int t : it.tris)
                    for (int v = 0; v < 3; ++v) {
                        const float *p = &s.tri[14 * static_cast<size_t>(t) + 4 + 3 * v];
                        out.push_back({p[0], p[1], p[2]});
                    }
            };
            std::vector<V3> pts;
            verts_of(items[seed], pts);
            V3 gc; double gr;
            bounding_sphere(pts, gc, gr);
            for (size_t k = 1; k < take; ++k) {
                size_t best = items.size();
                double best_r = INFINITY;
                for (size_t i = 0; i < items.size(); ++i) {
                    if (used[i]) continue;
                    std::vector<V3> q;
                    verts_of(items[i], q);
                    double r = gr;   // Ritter-style growth of (gc, gr) over the candidate's vertices
                    V3 c = gc;
                    for (const V3 &p : q) {
                        const V3 d = sub(p, c);
                        const double dist = nrm(d);
                        if (dist > r) {
                            const double nr = (r + dist) / 2, f = (nr - r) / dist;
                            c = {c.x + d.x * f, c.y + d.y * f, c.z + d.z * f};
                            r = nr;
                        }
                    }
                    if (r < best_r || (r == best_r && (best == items.size() || less(items[i].tris[0], items[best].tris[0])))) { best_r = r; best = i; }
                }
                used[best] = 1;
                member.push_back(best);
                verts_of(items[best], pts);
                bounding_sphere(pts, gc, gr);
            }
            left -= take;
            Item g;
            g.c[0] = g.c[1] = g.c[2] = 0;
            std::sort(member.begin(), member.end(), [&](size_t x, size_t y) { return less(items[x].tris[0], items[y].tris[0]); });
            for (size_t i : member) {
                g.tris.insert(g.tris.end(), items[i].tris.begin(), items[i].tris.end());
                for (int k = 0; k < 3; ++k) g.c[k] += items[i].c[k] / static_cast<double>(member.size());
            }
            next.push_back(std::move(g));
        }
        // the partial group (if any) was formed last: it already sits at the end.  But a group of full SUBGROUPS must not
        // follow a partial subgroup inside one parent: partial items can only be the very last item of the level, which
        // holds because only the last-formed group can contain the (single) partial item... unless the peeling picked it
        // earlier: move the item with the fewest triangles to the end of its level.
        size_t small = 0;
        for (size_t i = 1; i < next.size(); ++i) if (next[i].tris.size() < next[small].tris.size()) small = i;
        if (next[small].tris.size() < next.back().tris.size()) std::swap(next[small], next.back());
        items.swap(next);
    }
    std::copy(items[0].tris.begin(), items[0].tris.end(), ids.begin() + b);
}

struct UnionFind {
    std::vector<int> p;
    explicit UnionFind(int n) : p(n) { for (int i = 0; i < n; ++i) p[i] = i; }
    int find(int x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
    void unite(int a, int b) { a = find(a); b = find(b); if (a != b) p[std::max(a, b)] = std::min(a, b); }
};

struct Box {
    double lo[3], hi[3];
    void grow(const Box &b) {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
    }
};
const Box kEmptyBox = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};

// Axis-aligned box of every point Triangle::Intersect can accept for this triangle (DESIGN.md "Culling"): barycentric
// coordinates >= -m_geo (the triangle grown about its centroid: vertex k moves to v_k + m (2 v_k - v_i - v_j)), at most
// h_max off the triangle's own plane (displacement h_max |n_axis| per axis), plus the float rounding of
// P* = o + d t* (eps_line).
Box acceptance_box(const TriGeo &g, double eps_line) {
    Box b = kEmptyBox;
    const double m = g.m_geo;
    for (int k = 0; k < 3; ++k) {
        const V3 &v = g.v[k], &a = g.v[(k + 1) % 3], &c = g.v[(k + 2) % 3];
        const double p[3] = {v.x + m * (2 * v.x - a.x - c.x), v.y + m * (2 * v.y - a.y - c.y), v.z + m * (2 * v.z - a.z - c.z)};
        for (int x = 0; x < 3; ++x) { b.lo[x] = std::min(b.lo[x], p[x]); b.hi[x] = std::max(b.hi[x], p[x]); }
    }
    const V3 nn = crs(sub(g.v[1], g.v[0]), sub(g.v[2], g.v[0]));
    const double len = nrm(nn);
    const double n[3] = {std::fabs(nn.x) / len, std::fabs(nn.y) / len, std::fabs(nn.z) / len};
    for (int x = 0; x < 3; ++x) {
        const double pad = g.h_max * n[x] * (1.0 + 1e-9) + eps_line;
        b.lo[x] -= pad;
        b.hi[x] += pad;
#ifdef PT_TEST_HOOKS
        // mutation testing: scale the box about its centre (1 = as shipped)
        const double c = 0.5 * (b.lo[x] + b.hi[x]), h = 0.5 * (b.hi[x] - b.lo[x]) * g_cull_mutation.box;
        b.lo[x] = c - h;
        b.hi[x] = c + h;
#endif
    }
    return b;
}

#ifndef PT_BVH_MODE
#define PT_BVH_MODE 1   // box-tree builder of big scenes: 0 = uniform depth (build_bvh), 1 = binary SAH collapsed to 8-wide nodes (build_bvh_sah)
#endif

// One node of the box tree from its own box and its children's: the children as 8-bit boxes in the node's frame, rounded outward.
// `base`: an inner node's first child node (its children are consecutive), a leaf's first slot / 8.
void quantise_node(BvhNode &q, const Box &nb, const std::vector<Box> &kids, bool leaf, uint32_t base) {
    std::memset(&q, 0, sizeof q);
    double extent = 0;
    for (int x = 0; x < 3; ++x) {
        float f = static_cast<float>(nb.lo[x]);
        if (static_cast<double>(f) > nb.lo[x]) f = std::nextafterf(f, -INFINITY);
        q.org[x] = f;
        extent = std::max(extent, nb.hi[x] - static_cast<double>(f));
    }
    int e = extent > 0 ? static_cast<int>(std::ceil(std::log2(extent / 255.0))) : -100;
    e = std::max(-100, std::min(100, e));
    while (std::ceil(extent / std::ldexp(1.0, e)) > 255.0) ++e;
    const double step = std::ldexp(1.0, e);
    const size_t k = kids.size();
    for (size_t c = 0; c < k; ++c)
        for (int x = 0; x < 3; ++x) {
            const double lo = std::floor((kids[c].lo[x] - static_cast<double>(q.org[x])) / step);
            const double hi = std::ceil((kids[c].hi[x] - static_cast<double>(q.org[x])) / step);
            q.lo[x][c] = static_cast<uint8_t>(std::max(0.0, std::min(255.0, lo)));
            q.hi[x][c] = static_cast<uint8_t>(std::max(0.0, std::min(255.0, hi)));
        }
    // (base: an inner node's first child or a leaf's ordinal -- the tree's slots are the FIRST slots of the global order, so a
    // leaf's first slot / 8 is its ordinal among the leaves --: below the node count either way, which check_table_limits bounds
    // by 2^20 before any table reaches a device; a tree beyond that is still built, with the field wrapped, and then refused)
    q.meta = static_cast<uint32_t>(e + 127) | (static_cast<uint32_t>(k - 1) << 8) | (leaf ? 1u << 11 : 0u) | ((base & 0xFFFFFu) << 12);
}

// Box tree of a big scene over the (non-degenerate, small) triangles `ids`: uniform depth, up to 8 children per node,
// every node's triangles split into spatially compact children of (nearly) equal size.  Appends the tree's slots
// (8 per leaf, -1 = empty) to `order`, which must be empty: the tree's slots are the first slots.
void build_bvh(const HostScene &s, const std::vector<TriGeo> &geo, const std::vector<Centroid> &cen, std::vector<int> ids,
               double eps_line, CullTables &out, std::vector<int> &order) {
    out.bvh.clear();
    out.bvh_inner = 0;
    out.bvh_depth = 0;
    out.bvh_err = static_cast<float>(5.0e-7 * PT_MUT(box_err));   // first-order worst case 3.6e-7 (tests/test_cull_margins_host.py)
    const size_t n = ids.size();
    if (n == 0) return;
    std::vector<VBox> tbox(geo.size());
    for (int t : ids) {
        tbox[t].reset();
        for (const V3 &v : geo[t].v) {
            const double p[3] = {v.x, v.y, v.z};
            for (int x = 0; x < 3; ++x) { tbox[t].lo[x] = std::min(tbox[t].lo[x], p[x]); tbox[t].hi[x] = std::max(tbox[t].hi[x], p[x]); }
        }
    }
    int top = 1;
    for (size_t cap = kFan; cap < n; cap *= kFan) ++top;
    struct Range { size_t b, e; };
    std::vector<std::vector<Range>> levels(top + 1);       // levels[L] = nodes of level L in BFS order; level 1 = leaves
    std::vector<std::vector<uint32_t>> child_base(top + 1);
    levels[top].push_back({0, n});
    for (int L = top; L >= 2; --L) {
        size_t cap = 1;
        for (int k = 0; k < L - 1; ++k) cap *= kFan;       // capacity of a child
        for (const Range &r : levels[L]) {
            // as many children as keeps them at most half full -- in effect eight wherever the count allows: a uniform-depth
            // tree over n triangles has room for up to 8x n anyway, and nodes with few, full children only add levels
            // whose boxes prune little (x195 replica: 14.1 instead of 17.4 box rounds per wave-segment, +13 %; fill
            // factors from 0.25 to 0.5 build the same trees for the replicas, 0.55 and above lose) --
            // never fewer than the capacity demands
            const size_t cnt = r.e - r.b, k_min = (cnt + cap - 1) / cap;
#ifdef PT_TEST_HOOKS
            const double fill = g_cull_mutation.bvh_fill;
#else
            const double fill = 0.5;
#endif
            const size_t k_want = static_cast<size_t>(std::ceil(static_cast<double>(cnt) / (fill * static_cast<double>(cap))));
            const size_t k = std::min<size_t>(std::min<size_t>(kFan, cnt), std::max(k_min, k_want));
            std::vector<size_t> sizes;
            split_sah(ids, r.b, r.e, k, cap, s, cen, tbox, sizes);
            child_base[L].push_back(static_cast<uint32_t>(levels[L - 1].size()));
            size_t at = r.b;
            for (size_t sz : sizes) {
                levels[L - 1].push_back({at, at + sz});
                at += sz;
            }
        }
    }
    // node indices: level `top` first, leaves last
    std::vector<size_t> level_first(top + 2, 0);
    size_t total = 0;
    for (int L = top; L >= 1; --L) { level_first[L] = total; total += levels[L].size(); }
    out.bvh_inner = static_cast<uint32_t>(level_first[1]);
    out.bvh_depth = static_cast<uint32_t>(top);
    out.bvh.resize(total);
    std::vector<Box> node_box(total, kEmptyBox);
    std::vector<std::vector<Box>> kid_box(total);
    // leaves: slots and triangle boxes
    order.assign(levels[1].size() * kFan, -1);
    for (size_t j = 0; j < levels[1].size(); ++j) {
        const Range &r = levels[1][j];
        const size_t node = level_first[1] + j;
        std::sort(ids.begin() + r.b, ids.begin() + r.e, GeoLess{&s, &cen, 0});   // canonical order inside a leaf
        for (size_t c = 0; c < r.e - r.b; ++c) {
            order[j * kFan + c] = ids[r.b + c];
            kid_box[node].push_back(acceptance_box(geo[ids[r.b + c]], eps_line));
            node_box[node].grow(kid_box[node].back());
        }
    }
    // internal nodes, bottom-up
    for (int L = 2; L <= top; ++L)
        for (size_t j = 0; j < levels[L].size(); ++j) {
            const size_t node = level_first[L] + j;
            const size_t first = level_first[L - 1] + child_base[L][j];
            const size_t k = (j + 1 < levels[L].size() ? child_base[L][j + 1] : levels[L - 1].size()) - child_base[L][j];
            for (size_t c = 0; c < k; ++c) {
                kid_box[node].push_back(node_box[first + c]);
                node_box[node].grow(node_box[first + c]);
            }
        }
    for (size_t node = 0; node < total; ++node) {
        uint32_t base = static_cast<uint32_t>(node - level_first[1]);   // leaf j holds slots 8 j ...
        const bool leaf = node >= level_first[1];
        if (!leaf) {   // internal: index of the first child node
            int L = top;
            while (node >= level_first[L] + levels[L].size()) --L;
            base = static_cast<uint32_t>(level_first[L - 1] + child_base[L][node - level_first[L]]);
        }
        quantise_node(out.bvh[node], node_box[node], kid_box[node], leaf, base);
    }
}

// The same tree WITHOUT the uniform depth: a binary tree built top down with the surface-area heuristic (every cut where
// area(left) n_left + area(right) n_right is smallest over the three axes; binned above 512 triangles), leaves of at most 8
// triangles, then collapsed to nodes of up to 8 children (a node takes a binary node's two children and keeps replacing the
// child of the largest area by its two children) -- the textbook wide-tree construction.  Big empty regions end up high in the
// tree and dense ones get the depth they need: on the x195 replica a ray visits 5 % fewer nodes and 7 % fewer child boxes than in
// the uniform-depth tree (profiles/r03_ab_logs.txt tree02).  Nodes are numbered breadth-first, a node's children are consecutive
// whatever their kind; a leaf's slots follow the order in which leaves are created.
void build_bvh_sah(const HostScene &s, const std::vector<TriGeo> &geo, const std::vector<Centroid> &cen, std::vector<int> ids,
                   double eps_line, CullTables &out, std::vector<int> &order) {
    out.bvh.clear();
    out.bvh_inner = 0;
    out.bvh_depth = 0;
    out.bvh_err = static_cast<float>(5.0e-7 * PT_MUT(box_err));
    const size_t n = ids.size();
    if (n == 0) return;
    std::vector<VBox> tbox(geo.size());
    for (int t : ids) {
        tbox[t].reset();
        for (const V3 &v : geo[t].v) {
            const double p[3] = {v.x, v.y, v.z};
            for (int x = 0; x < 3; ++x) { tbox[t].lo[x] = std::min(tbox[t].lo[x], p[x]); tbox[t].hi[x] = std::max(tbox[t].hi[x], p[x]); }
        }
    }
    std::sort(ids.begin(), ids.end(), GeoLess{&s, &cen, 0});   // a starting order that depends on geometry only
    struct Bin { size_t b, e; int left, right; VBox box; };
    std::vector<Bin> bin;
    bin.reserve(2 * n / 4 + 16);
    // (iterative: a work list instead of recursion, children created in a fixed order)
    {
        Bin root = {0, n, -1, -1, {}};
        root.box.reset();
        for (size_t i = 0; i < n; ++i) root.box.add(tbox[ids[i]]);
        bin.push_back(root);
    }
    std::vector<int> scratch;
    for (size_t at = 0; at < bin.size(); ++at) {
        const size_t b = bin[at].b, e = bin[at].e, cnt = e - b;
        if (cnt <= static_cast<size_t>(kFan)) continue;   // a leaf
        size_t cut = 0;          // triangles of the left child
        int cut_axis = -1;
        double best = INFINITY;
        if (cnt <= 512) {
            // exact sweep: every cut position along each axis
            std::vector<double> right_area(cnt);
            for (int ax = 0; ax < 3; ++ax) {
                std::sort(ids.begin() + b, ids.begin() + e, GeoLess{&s, &cen, ax});
                VBox acc;
                acc.reset();
                for (size_t i = cnt; i-- > 1;) { acc.add(tbox[ids[b + i]]); right_area[i] = acc.area(); }
                acc.reset();
                for (size_t i = 1; i < cnt; ++i) {
                    acc.add(tbox[ids[b + i - 1]]);
                    const double c = acc.area() * static_cast<double>(i) + right_area[i] * static_cast<double>(cnt - i);
                    if (c < best) { best = c; cut = i; cut_axis = ax; }
                }
            }
            if (cut_axis != 2) std::sort(ids.begin() + b, ids.begin() + e, GeoLess{&s, &cen, cut_axis});
        } else {
            // 32 bins per axis over the centroids' range
            constexpr int kBins = 32;
            double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (size_t i = b; i < e; ++i)
                for (int x = 0; x < 3; ++x) { clo[x] = std::min(clo[x], cen[ids[i]].c[x]); chi[x] = std::max(chi[x], cen[ids[i]].c[x]); }
            int best_bin = -1;
            for (int ax = 0; ax < 3; ++ax) {
                if (!(chi[ax] > clo[ax])) continue;
                const double scale = kBins / (chi[ax] - clo[ax]);
                VBox bb[kBins];
                size_t bc[kBins] = {};
                for (auto &x : bb) x.reset();
                for (size_t i = b; i < e; ++i) {
                    const int k = std::min(kBins - 1, static_cast<int>((cen[ids[i]].c[ax] - clo[ax]) * scale));
                    bb[k].add(tbox[ids[i]]);
                    ++bc[k];
                }
                double ra[kBins];
                VBox acc;
                acc.reset();
                for (int k = kBins - 1; k >= 1; --k) { acc.add(bb[k]); ra[k] = acc.area(); }
                acc.reset();
                size_t nl = 0;
                for (int k = 1; k < kBins; ++k) {
                    acc.add(bb[k - 1]);
                    nl += bc[k - 1];
                    if (nl == 0 || nl == cnt) continue;
                    const double c = acc.area() * static_cast<double>(nl) + ra[k] * static_cast<double>(cnt - nl);
                    if (c < best) { best = c; cut = nl; cut_axis = ax; best_bin = k; }
                }
            }
            if (cut_axis >= 0) {
                const double scale = kBins / (chi[cut_axis] - clo[cut_axis]);
                const int ax = cut_axis, kb = best_bin;
                const double lo = clo[ax];
                const auto mid = std::stable_partition(ids.begin() + b, ids.begin() + e, [&](int t) {
                    return std::min(kBins - 1, static_cast<int>((cen[t].c[ax] - lo) * scale)) < kb;
                });
                cut = static_cast<size_t>(mid - (ids.begin() + b));
            }
        }
        if (cut_axis < 0 || cut == 0 || cut >= cnt) {   // all centroids equal (or the heuristic found nothing): halves in canonical order
            std::sort(ids.begin() + b, ids.begin() + e, GeoLess{&s, &cen, 0});
            cut = cnt / 2;
        }
        Bin l = {b, b + cut, -1, -1, {}}, r = {b + cut, e, -1, -1, {}};
        l.box.reset();
        r.box.reset();
        for (size_t i = l.b; i < l.e; ++i) l.box.add(tbox[ids[i]]);
        for (size_t i = r.b; i < r.e; ++i) r.box.add(tbox[ids[i]]);
        bin[at].left = static_cast<int>(bin.size());
        bin.push_back(l);
        bin[at].right = static_cast<int>(bin.size());
        bin.push_back(r);
    }
    // collapse, breadth first: wide node w <-> binary node wide_bin[w]; its children are numbered consecutively
    std::vector<int> wide_bin = {0};
    std::vector<std::vector<int>> wide_kids;   // binary nodes that become the children (empty for a leaf)
    for (size_t w = 0; w < wide_bin.size(); ++w) {
        const Bin &bn = bin[wide_bin[w]];
        std::vector<int> kids;
        if (bn.left >= 0) {
            kids = {bn.left, bn.right};
            while (kids.size() < static_cast<size_t>(kFan)) {
                int pick = -1;
                double pa = -1;
                for (size_t i = 0; i < kids.size(); ++i)
                    if (bin[kids[i]].left >= 0 && bin[kids[i]].box.area() > pa) { pa = bin[kids[i]].box.area(); pick = static_cast<int>(i); }
                if (pick < 0) break;
                const int k = kids[pick];
                kids[pick] = bin[k].left;
                kids.insert(kids.begin() + pick + 1, bin[k].right);
            }
        }
        wide_kids.push_back(kids);
        for (int k : kids) wide_bin.push_back(k);   // (children of w: consecutive node indices, in this order)
    }
    const size_t total = wide_bin.size();
    {   // levels of the wide tree: nodes are in breadth-first order, so a node's level is its parent's + 1 in one forward pass
        std::vector<uint32_t> level(total, 1);
        uint32_t next = 1, deepest = 1;
        for (size_t w = 0; w < total; ++w)
            for (size_t c = 0; c < wide_kids[w].size(); ++c) {
                level[next] = level[w] + 1;
                deepest = std::max(deepest, level[next]);
                ++next;
            }
        out.bvh_depth = deepest;
    }
    // a node's first child: nodes are appended in the order of their parents
    std::vector<uint32_t> first_child(total, 0);
    {
        uint32_t next = 1;
        for (size_t w = 0; w < total; ++w) { first_child[w] = next; next += static_cast<uint32_t>(wide_kids[w].size()); }
    }
    out.bvh.resize(total);
    std::vector<Box> node_box(total, kEmptyBox);
    std::vector<std::vector<Box>> kid_box(total);
    std::vector<uint32_t> leaf_slot(total, 0);
    size_t n_leaves = 0;
    order.clear();
    for (size_t w = 0; w < total; ++w) {
        if (!wide_kids[w].empty()) continue;
        const Bin &bn = bin[wide_bin[w]];
        std::sort(ids.begin() + bn.b, ids.begin() + bn.e, GeoLess{&s, &cen, 0});   // canonical order inside a leaf
        leaf_slot[w] = static_cast<uint32_t>(n_leaves);
        order.resize((n_leaves + 1) * kFan, -1);
        for (size_t c = 0; c < bn.e - bn.b; ++c) {
            order[n_leaves * kFan + c] = ids[bn.b + c];
            kid_box[w].push_back(acceptance_box(geo[ids[bn.b + c]], eps_line));
            node_box[w].grow(kid_box[w].back());
        }
        ++n_leaves;
    }
    for (size_t w = total; w-- > 0;) {   // children have larger indices than their parent: bottom-up in one backward pass
        if (wide_kids[w].empty()) continue;
        for (size_t c = 0; c < wide_kids[w].size(); ++c) {
            kid_box[w].push_back(node_box[first_child[w] + c]);
            node_box[w].grow(node_box[first_child[w] + c]);
        }
    }
    for (size_t w = 0; w < total; ++w) {
        const bool leaf = wide_kids[w].empty();
        quantise_node(out.bvh[w], node_box[w], kid_box[w], leaf, leaf ? leaf_slot[w] : first_child[w]);
    }
    out.bvh_inner = static_cast<uint32_t>(total - n_leaves);
}

}  // namespace

void build_cull_tables(const HostScene &s, float eps_f, CullTables &out) {
    const int T = s.n_tri();
    const double eps = eps_f;
    out = CullTables();
    out.eps = eps_f;
    // sphere trees + small-scene kernels, or one box tree + big-scene kernels (pt_scene.hpp: kBigSceneTriangles)
    int big_threshold = kBigSceneTriangles;
#ifdef PT_TEST_HOOKS
    if (g_cull_mutation.big_threshold >= 0) big_threshold = std::min(g_cull_mutation.big_threshold, kSmallSceneMaxTriangles);
#endif
    const bool big = T > big_threshold;
    out.big = big;

    // ---- scene-wide bounds
    double r_max = 20.0;   // the camera origin (0,0,-20), main.cpp:129
    for (int i = 0; i < T; ++i)
        for (int k = 4; k < 13; ++k) r_max = std::max(r_max, static_cast<double>(std::fabs(s.tri[14 * static_cast<size_t>(i) + k])));
    const double r_org = r_max + 1.0;                       // ray origins sit on surfaces, offset by eps*N
    out.r_org = static_cast<float>(r_org);
    const double d_max = 2.0 * std::sqrt(3.0) * r_org;      // bound on |c - o| for c, o inside the scene box
    // rounding of P* = o + d*t* (per component <= u(2|t| + |o|)) and of the centre-to-origin vector
    const double eps_line = 8.0 * kU * (d_max + r_org);
    // float error of disc = |m|^2 - (m.d)^2 in the kernel, plus |d| != 1 by a few ulp
    const double disc_err = 24.0 * kU * d_max * d_max;

    std::vector<TriGeo> geo(T);
    std::vector<Centroid> cen(T);
    for (int i = 0; i < T; ++i) {
        geo[i] = tri_geometry(&s.tri[14 * static_cast<size_t>(i)], eps);
        const V3 *v = geo[i].v;
        cen[i] = {{(v[0].x + v[1].x + v[2].x) / 3, (v[0].y + v[1].y + v[2].y) / 3, (v[0].z + v[1].z + v[2].z) / 3}};
    }

    // bounding sphere of the acceptance regions of the triangles ids[first, first + count)
    auto sphere_of = [&](const int *ids, int count, SphereRec &rec) {
        std::vector<V3> pts;
        bool inf = false;
        for (int k = 0; k < count; ++k) {
            if (geo[ids[k]].degenerate) inf = true;
            for (const V3 &v : geo[ids[k]].v) pts.push_back(v);
        }
        V3 c; double rad;
        bounding_sphere(pts, c, rad);
        double reff = 0;
        for (int k = 0; k < count && !inf; ++k) {
            const TriGeo &g = geo[ids[k]];
            double dmax = 0;
            for (const V3 &v : g.v) dmax = std::max(dmax, nrm(sub(v, c)));
            // accepted point = sum(lambda_k v_k) + h n, lambda_k >= -m_geo  =>  |P - c| <= (1 + 4 m_geo) dmax + h_max
            reff = std::max(reff, (1.0 + 4.0 * g.m_geo) * dmax + g.h_max + eps_line);
        }
        rec.c[0] = static_cast<float>(c.x); rec.c[1] = static_cast<float>(c.y); rec.c[2] = static_cast<float>(c.z);
        // the centre is rounded to float: grow by that displacement
        const double c_round = nrm(sub(c, V3{rec.c[0], rec.c[1], rec.c[2]}));
        double r2 = (reff + c_round) * (reff + c_round) * (1.0 + 1e-6) + disc_err;
        r2 *= PT_MUT(sphere_r2);
        rec.r2 = (inf || !std::isfinite(r2)) ? INFINITY : static_cast<float>(r2 * (1.0 + 2e-7));
        return inf ? INFINITY : reff;
    };

    // ---- classes.  LARGE: a triangle whose own sphere is a sizeable part of the scene (walls) -- and every triangle
    // that cannot be bounded at all (degenerate) -- is culled by a barycentric record, wave-uniformly.  SMALL: the rest,
    // under a hierarchy of bounding volumes.
    std::vector<uint8_t> large(T);
    std::vector<double> own_radius(T);
    for (int i = 0; i < T; ++i) {
        SphereRec tmp;
        const double reff = sphere_of(&i, 1, tmp);
        own_radius[i] = reff;
        large[i] = !(reff < 0.12 * r_max);
    }
    // A big scene's few emitters (the light of a room) join the large class whatever their size: their records alone then
    // tell which rays of a path's last segment can still contribute (pt_kernels.hip, "last segment").
    if (big) {
        auto emits = [&](int t) {
            const float *m = &s.mat[10 * static_cast<size_t>(s.tri_mat[t])];
            return m[3] != 0.0f || m[4] != 0.0f || m[5] != 0.0f;
        };
        // (not tiny ones: the margins of the barycentric test scale with the inverse size of the smallest record, and
        // every wall would pay for a pinhead of a light)
        int n_emit = 0;
        bool sizeable = true;
        for (int i = 0; i < T; ++i)
            if (emits(i)) {
                ++n_emit;
                sizeable = sizeable && own_radius[i] >= 0.01 * r_max;
            }
        if (n_emit <= 8 && sizeable)
            for (int i = 0; i < T; ++i) if (emits(i)) large[i] = 1;
    }
    // connected groups of small triangles (triangles sharing a vertex position): the objects of the scene
    UnionFind uf(T);
    {
        struct Key { uint32_t b[3]; int tri; };
        std::vector<Key> keys;
        for (int i = 0; i < T; ++i)
            if (!large[i])
                for (int v = 0; v < 3; ++v) {
                    Key k;
                    std::memcpy(k.b, &s.tri[14 * static_cast<size_t>(i) + 4 + 3 * v], 12);
                    k.tri = i;
                    keys.push_back(k);
                }
        std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
            const int m = std::memcmp(a.b, b.b, 12);
            return m != 0 ? m < 0 : a.tri < b.tri;
        });
        for (size_t k = 1; k < keys.size(); ++k)
            if (std::memcmp(keys[k].b, keys[k - 1].b, 12) == 0) uf.unite(keys[k].tri, keys[k - 1].tri);
    }
    std::vector<std::vector<int>> groups;   // small triangles by connected group
    {
        std::vector<int> group_of(T, -1);
        for (int i = 0; i < T; ++i) {
            if (large[i]) continue;
            const int r = uf.find(i);
            if (group_of[r] < 0) { group_of[r] = static_cast<int>(groups.size()); groups.emplace_back(); }
            groups[group_of[r]].push_back(i);
        }
    }
    // A group of one or two small triangles (the light of a room) costs more as a cluster of its own -- descriptor, sphere
    // tests, a publication per segment -- than as one more record of the large class ...
#ifdef PT_TEST_HOOKS
    const bool absorb = !g_cull_mutation.no_absorb;
#else
    const bool absorb = true;
#endif
    bool any_large = false;
    for (int i = 0; i < T; ++i) any_large |= large[i] != 0;
    if (absorb && any_large && !big) {
        for (auto &g : groups) {
            // ... unless they are so small that their barycentric gradients (1 / height) would blow up the margin of
            // the whole class: the test of every large triangle uses the class-wide a_max
            bool fits = g.size() <= 2;
            for (int t : g) fits = fits && !geo[t].degenerate && geo[t].a_max * r_max <= 64.0;
            if (!fits) continue;
            for (int t : g) large[t] = 1;
            g.clear();
        }
        groups.erase(std::remove_if(groups.begin(), groups.end(), [](const std::vector<int> &g) { return g.empty(); }), groups.end());
    }
    // Can a path leave the envelope (origins within r_org) the margins are derived for?  Only through a hit point outside
    // it, i.e. only if some triangle's acceptance region reaches beyond it: a near-degenerate triangle (the reference
    // accepts it for points anywhere along its axis) or a long sliver at the edge of the scene.  Scenes without such
    // triangles (Tor.obj, the replicas) skip the per-segment origin test altogether.
    out.may_leave_envelope = false;
    for (int i = 0; i < T && !out.may_leave_envelope; ++i) {
        if (geo[i].degenerate) { out.may_leave_envelope = true; break; }
        const Box b = acceptance_box(geo[i], eps_line);
        for (int x = 0; x < 3; ++x)
            if (!(b.lo[x] > -(r_org - 0.01)) || !(b.hi[x] < r_org - 0.01)) out.may_leave_envelope = true;
    }

    // ---- slot order of the small class
    std::vector<int> order;                       // slot -> triangle (-1 = padding)
    std::vector<std::pair<int, int>> small_runs;  // small scenes: [first slot, count) of each cluster
    if (big) {
        std::vector<int> ids;
        for (const auto &g : groups) ids.insert(ids.end(), g.begin(), g.end());
        int mode = PT_BVH_MODE;
#ifdef PT_TEST_HOOKS
        if (g_cull_mutation.bvh_mode >= 0) mode = g_cull_mutation.bvh_mode;
#endif
        if (mode == 1) build_bvh_sah(s, geo, cen, ids, eps_line, out, order);
        else build_bvh(s, geo, cen, ids, eps_line, out, order);
        // The SAH tree's depth follows the geometry (nested shells of geometrically growing triangles: 12 levels for 12 000
        // triangles) and the walk's stack slack bounds it (kMaxBvhDepth): such a scene gets the uniform-depth tree instead.
#ifdef PT_TEST_HOOKS
        const uint32_t depth_cap = static_cast<uint32_t>(g_cull_mutation.bvh_depth_cap);
#else
        const uint32_t depth_cap = static_cast<uint32_t>(kMaxBvhDepth);
#endif
        if (mode == 1 && out.bvh_depth > depth_cap) {
            order.clear();
            build_bvh(s, geo, cen, ids, eps_line, out, order);
        }
    } else if (!groups.empty()) {
        int max_clusters = kMaxClusters;
#ifdef PT_TEST_HOOKS
        if (g_cull_mutation.max_clusters >= 0) max_clusters = g_cull_mutation.max_clusters;
#endif
        if (static_cast<int>(groups.size()) > max_clusters) {   // a cloud of loose triangles: one tree over all of them
            std::vector<int> all;
            for (const auto &g : groups) all.insert(all.end(), g.begin(), g.end());
            groups.assign(1, all);
        }
        // clusters in an order that depends on geometry only
        std::vector<std::pair<Centroid, size_t>> keyed;
        for (size_t g = 0; g < groups.size(); ++g) {
            Centroid m = {{0, 0, 0}};
            for (int t : groups[g]) for (int k = 0; k < 3; ++k) m.c[k] += cen[t].c[k] / static_cast<double>(groups[g].size());
            keyed.push_back({m, g});
        }
        std::sort(keyed.begin(), keyed.end(), [](const std::pair<Centroid, size_t> &a, const std::pair<Centroid, size_t> &b) {
            for (int k = 0; k < 3; ++k) if (a.first.c[k] != b.first.c[k]) return a.first.c[k] < b.first.c[k];
            return a.second < b.second;
        });
        // cost of laying the implicit 8-ary sphere tree over an order: sum of r^2 over its nodes (a ray meets a sphere
        // with probability ~ r^2)
        auto order_cost = [&](const std::vector<int> &ids) {
            double cost = 0;
            const long long n = static_cast<long long>(ids.size());
            for (long long span = kFan; span * kFan < n * kFan && span < n; span *= kFan)   // levels below the top one
                for (long long f = 0; f < n; f += span) {
                    SphereRec sr;
                    const double reff = sphere_of(&ids[f], static_cast<int>(std::min<long long>(span, n - f)), sr);
                    cost += std::isfinite(reff) ? reff * reff : 0.0;
                }
            return cost;
        };
        for (const auto &kg : keyed) {
            // the cheaper of two spatial arrangements
            // (two geometry-only arrangements are tried and the cheaper kept; results do not depend on the choice)
            std::vector<int> ids = groups[kg.second], patches = ids;
            arrange_implicit(ids, 0, ids.size(), s, cen);          // axis-aligned cells
            if (patches.size() <= 4096) {                          // compact patches (quadratic in the group size)
                arrange_patches(patches, 0, patches.size(), s, cen);
#ifdef PT_TEST_HOOKS
                if (g_cull_mutation.order_mode == 1) ids = groups[kg.second];   // as filed
                else if (g_cull_mutation.order_mode == 3) ids = patches;
                else if (g_cull_mutation.order_mode == 0 && order_cost(patches) < order_cost(ids)) ids = patches;
#else
                if (order_cost(patches) < order_cost(ids)) ids = patches;
#endif
            }
            small_runs.push_back({static_cast<int>(order.size()), static_cast<int>(ids.size())});
            order.insert(order.end(), ids.begin(), ids.end());
        }
    }
    const int n_small_slots = static_cast<int>(order.size());

    // ---- slot order of the large class: coplanar pairs that share an edge first (they become quad records), then singles
    {
        std::vector<int> lg;
        for (int i = 0; i < T; ++i) if (large[i]) lg.push_back(i);
        const GeoLess less{&s, &cen, 0};
        std::sort(lg.begin(), lg.end(), less);
        std::vector<uint8_t> used(lg.size(), 0);
        std::vector<int> pairs, singles;
        for (size_t a = 0; a < lg.size(); ++a) {
            if (used[a]) continue;
            const float *ra = &s.tri[14 * static_cast<size_t>(lg[a])];
            int mate = -1;
            for (size_t b = a + 1; b < lg.size() && mate < 0 && !geo[lg[a]].degenerate; ++b) {
                if (used[b] || geo[lg[b]].degenerate) continue;
                const float *rb = &s.tri[14 * static_cast<size_t>(lg[b])];
                if (std::memcmp(ra, rb, 16) != 0) continue;   // ONE stored plane
                int shared = 0;
                for (int x = 0; x < 3; ++x)
                    for (int y = 0; y < 3; ++y) shared += std::memcmp(ra + 4 + 3 * x, rb + 4 + 3 * y, 12) == 0;
                if (shared == 2) mate = static_cast<int>(b);
            }
            if (mate >= 0) {
                used[a] = used[mate] = 1;
                pairs.push_back(lg[a]);
                pairs.push_back(lg[mate]);
            } else {
                used[a] = 1;
                singles.push_back(lg[a]);
            }
        }
        order.insert(order.end(), pairs.begin(), pairs.end());
        order.insert(order.end(), singles.begin(), singles.end());
    }
    const int n_slots = static_cast<int>(order.size());
    const int n_large = n_slots - n_small_slots;

    // ---- tables in slot order
    out.slot_tri.resize(n_slots);
    out.exact_slot.resize(n_slots);
    DeviceTables dev_tables;
    build_device_tables(s, dev_tables);
    for (int k = 0; k < n_slots; ++k) {
        out.slot_tri[k] = order[k] < 0 ? kNoTriangle : static_cast<uint32_t>(order[k]);
        if (order[k] >= 0) {
            out.exact_slot[k] = dev_tables.exact[order[k]];
        } else {
            ExactRec e;
            std::memset(&e, 0, sizeof e);
            e.plane[0] = e.plane[1] = e.plane[2] = e.plane[3] = NAN;   // a padding slot can never be accepted
            e.orig = -1;
            out.exact_slot[k] = e;
        }
    }

    double a_max = 0, inv_2s_max = 0, diam2_2s_max = 0, quad_slack = 0;
    const SphereRec never = {{0, 0, 0}, -1.0e30f};
    // ---- small-scene clusters: an implicit 8-ary tree of bounding spheres over each run of slots
    for (const auto &run : small_runs) {
        const int i = run.first, n = run.second;
        ClusterDesc cd;
        std::memset(&cd, 0, sizeof cd);
        SphereRec cs;
        sphere_of(&order[i], n, cs);
        cd.c[0] = cs.c[0]; cd.c[1] = cs.c[1]; cd.c[2] = cs.c[2]; cd.r2 = cs.r2;
        cd.first_tri = static_cast<uint32_t>(i);
        cd.n_tri = static_cast<uint32_t>(n);
        cd.kind = 0u;
        cd.data_off = static_cast<uint32_t>(out.spheres.size());
        long long span = 1;   // triangles per node of the current level
        for (int level = 0;; ++level, span *= kFan) {
            const long long count = (n + span - 1) / span;
            if (level > 0) cd.level_off[level - 1] = static_cast<uint32_t>(out.spheres.size() - cd.data_off);
            for (long long j = 0; j < (count + kFan - 1) / kFan * kFan; ++j) {
                SphereRec sr = never;
                if (j < count) {
                    const long long f = j * span;
                    sphere_of(&order[i + static_cast<int>(f)], static_cast<int>(std::min<long long>(span, n - f)), sr);
                }
                out.spheres.push_back(sr);
            }
            if (count <= kFan || level == kMaxLevels - 1) {
                cd.n_levels = static_cast<uint32_t>(level + 1);
                break;
            }
        }
        out.clusters.push_back(cd);
    }
    // ---- the large class: one cluster of barycentric records
    if (n_large > 0) {
        const int i = n_small_slots, n = n_large;
        ClusterDesc cd;
        std::memset(&cd, 0, sizeof cd);
        SphereRec cs;
        sphere_of(&order[i], n, cs);
        cd.c[0] = cs.c[0]; cd.c[1] = cs.c[1]; cd.c[2] = cs.c[2]; cd.r2 = cs.r2;
        cd.first_tri = static_cast<uint32_t>(i);
        cd.n_tri = static_cast<uint32_t>(n);
        cd.kind = 1u;
        const int n_words = (n + kChunk - 1) / kChunk;
        {
            cd.data_off = static_cast<uint32_t>(out.bary.size());
            for (int k = 0; k < n_words * kChunk; ++k) {
                CullRec c;
                std::memset(&c, 0, sizeof c);
                if (k < n) {
                    const float *r = &s.tri[14 * static_cast<size_t>(order[i + k])];
                    const TriGeo &g = geo[order[i + k]];
                    const V3 e1 = sub(g.v[1], g.v[0]), e2 = sub(g.v[2], g.v[0]);
                    const V3 nn = crs(e1, e2);
                    const double s2 = dt(nn, nn);
                    const V3 au = {crs(e2, nn).x / s2, crs(e2, nn).y / s2, crs(e2, nn).z / s2};
                    const V3 av = {crs(nn, e1).x / s2, crs(nn, e1).y / s2, crs(nn, e1).z / s2};
                    c.n[0] = r[0]; c.n[1] = r[1]; c.n[2] = r[2]; c.w = r[3];
                    c.au[0] = static_cast<float>(au.x); c.au[1] = static_cast<float>(au.y); c.au[2] = static_cast<float>(au.z);
                    c.av[0] = static_cast<float>(av.x); c.av[1] = static_cast<float>(av.y); c.av[2] = static_cast<float>(av.z);
                    c.cu = static_cast<float>(-dt(au, g.v[0]));
                    c.cv = static_cast<float>(-dt(av, g.v[0]));
                    if (g.degenerate) {
                        // NaN coefficients make every comparison of the cull test false: the triangle is always kept
                        c.au[0] = c.au[1] = c.au[2] = c.av[0] = c.av[1] = c.av[2] = c.cu = c.cv = NAN;
                    } else {
                        a_max = std::max(a_max, g.a_max);
                        inv_2s_max = std::max(inv_2s_max, 1.0 / (2.0 * g.area2));
                        diam2_2s_max = std::max(diam2_2s_max, g.diam * g.diam / (2.0 * g.area2));
                    }
                }
                out.bary.push_back(c);
            }
            // Quads: two consecutive large triangles (even slot first) that lie in ONE stored plane and share an edge
            // are the two halves of a (near-)parallelogram s0, a, s1, b with diagonal s0-s1.  With P = s0 + alpha*(a-s0) +
            // beta*(b-s0), half A = (s0, s1, a) has barycentrics (beta, alpha-beta, 1-alpha) and half B = (s0, s1, b) has
            // (alpha, beta-alpha, 1-beta), so ONE plane evaluation and two affine rows cull both.  The identity is exact
            // only for an exact parallelogram; the deviation of each half's own barycentrics from it is measured below
            // over the region alpha, beta in [-1, 2] and added to the margin (quad_slack); pairs that deviate by more
            // than 1 % are left as two triangles.  (Outside that region the derived minimum is <= -1 and the deviation
            // grows at most linearly, so a far point can never be kept by one test and rejected by the other.)
            for (int w = 0; w < n_words && w < kMaxLevels - 1; ++w) {
                uint32_t qmask = 0;
                for (int k = w * kChunk; k + 1 < std::min(n, (w + 1) * kChunk); k += 2) {
                    const float *ra = &s.tri[14 * static_cast<size_t>(order[i + k])], *rb = &s.tri[14 * static_cast<size_t>(order[i + k + 1])];
                    if (std::memcmp(ra, rb, 16) != 0 || geo[order[i + k]].degenerate || geo[order[i + k + 1]].degenerate) continue;
                    // shared vertices (bitwise) and the two apexes
                    int sa[2], sb[2], ns = 0, apex_a = -1, apex_b = -1;
                    bool used_b[3] = {false, false, false};
                    for (int x = 0; x < 3; ++x) {
                        int match = -1;
                        for (int y = 0; y < 3; ++y)
                            if (!used_b[y] && std::memcmp(ra + 4 + 3 * x, rb + 4 + 3 * y, 12) == 0) { match = y; break; }
                        if (match >= 0 && ns < 2) { sa[ns] = x; sb[ns] = match; used_b[match] = true; ++ns; }
                        else apex_a = x;
                    }
                    if (ns != 2 || apex_a < 0) continue;
                    for (int y = 0; y < 3; ++y) if (!used_b[y]) apex_b = y;
                    const TriGeo &ga = geo[order[i + k]], &gb = geo[order[i + k + 1]];
                    const V3 s0 = ga.v[sa[0]], pa = ga.v[apex_a], pb = gb.v[apex_b];
                    const V3 ea = sub(pa, s0), eb = sub(pb, s0);
                    const V3 nn = crs(ea, eb);
                    const double s2 = dt(nn, nn);
                    if (!(s2 > 0)) continue;
                    const V3 ral = {crs(eb, nn).x / s2, crs(eb, nn).y / s2, crs(eb, nn).z / s2};   // alpha row: dual of ea
                    const V3 rbe = {crs(nn, ea).x / s2, crs(nn, ea).y / s2, crs(nn, ea).z / s2};   // beta row: dual of eb
                    const double cal = -dt(ral, s0), cbe = -dt(rbe, s0);
                    // true barycentrics of each half (orthogonal projection onto its own plane), as affine functions
                    auto bary_rows = [&](const TriGeo &g, int i0, int i1, int i2, V3 rows[3], double cst[3]) {
                        const V3 e1 = sub(g.v[i1], g.v[i0]), e2 = sub(g.v[i2], g.v[i0]);
                        const V3 m = crs(e1, e2);
                        const double q = dt(m, m);
                        rows[1] = {crs(e2, m).x / q, crs(e2, m).y / q, crs(e2, m).z / q};   // weight of i1
                        rows[2] = {crs(m, e1).x / q, crs(m, e1).y / q, crs(m, e1).z / q};   // weight of i2
                        rows[0] = {-(rows[1].x + rows[2].x), -(rows[1].y + rows[2].y), -(rows[1].z + rows[2].z)};
                        cst[1] = -dt(rows[1], g.v[i0]); cst[2] = -dt(rows[2], g.v[i0]); cst[0] = 1.0 - cst[1] - cst[2];
                    };
                    V3 rowa[3], rowb[3];
                    double ca[3], cb[3];
                    bary_rows(ga, sa[0], sa[1], apex_a, rowa, ca);   // weights of s0, s1, a
                    bary_rows(gb, sb[0], sb[1], apex_b, rowb, cb);   // weights of s0, s1, b
                    const V3 nu = {nn.x / std::sqrt(s2), nn.y / std::sqrt(s2), nn.z / std::sqrt(s2)};
                    const double hh = 0.05 * std::max(ga.diam, gb.diam);
                    double dev = 0;
                    for (int c8 = 0; c8 < 8; ++c8) {
                        const double al = (c8 & 1) ? 2.0 : -1.0, be = (c8 & 2) ? 2.0 : -1.0, h = (c8 & 4) ? hh : -hh;
                        const V3 P = {s0.x + al * ea.x + be * eb.x + h * nu.x, s0.y + al * ea.y + be * eb.y + h * nu.y, s0.z + al * ea.z + be * eb.z + h * nu.z};
                        const double a_ = dt(ral, P) + cal, b_ = dt(rbe, P) + cbe;
                        const double da[3] = {1 - a_, b_, a_ - b_}, db[3] = {1 - b_, a_, b_ - a_};   // derived weights of (s0, s1, apex)
                        for (int t3 = 0; t3 < 3; ++t3) {
                            dev = std::max(dev, std::fabs(dt(rowa[t3], P) + ca[t3] - da[t3]));
                            dev = std::max(dev, std::fabs(dt(rowb[t3], P) + cb[t3] - db[t3]));
                        }
                    }
                    if (!(dev < 0.01)) continue;
                    quad_slack = std::max(quad_slack, dev);
                    a_max = std::max(a_max, std::max(nrm(ral), std::max(nrm(rbe), nrm(sub(ral, rbe)))));
                    CullRec &q = out.bary[cd.data_off + k];
                    q.au[0] = static_cast<float>(ral.x); q.au[1] = static_cast<float>(ral.y); q.au[2] = static_cast<float>(ral.z);
                    q.av[0] = static_cast<float>(rbe.x); q.av[1] = static_cast<float>(rbe.y); q.av[2] = static_cast<float>(rbe.z);
                    q.cu = static_cast<float>(cal);
                    q.cv = static_cast<float>(cbe);
                    std::memset(&out.bary[cd.data_off + k + 1], 0, sizeof(CullRec));
                    qmask |= 1u << (k - w * kChunk);
                }
                cd.level_off[w] = qmask;   // large clusters have no sphere tree: the slots carry the quad masks
            }
        }
        out.clusters.push_back(cd);
    }
    // keep the tables non-empty and padded so that speculative wide scalar loads stay inside the allocation
    // (the root round of the kernel reads up to 8 x 64 records from a level's start without a bounds test and masks afterwards)
    for (int k = 0; k < 16 + 8 * 64; ++k) out.spheres.push_back(never);
    for (int k = 0; k < 4; ++k) { CullRec c; std::memset(&c, 0, sizeof c); out.bary.push_back(c); }

    // ---- big scenes: a barycentric record for every triangle, used to thin the (ray, triangle) pairs before the exact test
    double a_max_all = 0, inv_2s_max_all = 0, diam2_2s_max_all = 0;
    if (big) {
        out.bary_all.resize(static_cast<size_t>(n_slots) + 4);
        for (auto &c : out.bary_all) std::memset(&c, 0, sizeof c);
        for (int k = 0; k < n_slots; ++k) {
            if (order[k] < 0) continue;   // padding slots are never candidates
            CullRec &c = out.bary_all[k];
            const float *r = &s.tri[14 * static_cast<size_t>(order[k])];
            const TriGeo &g = geo[order[k]];
            const V3 e1 = sub(g.v[1], g.v[0]), e2 = sub(g.v[2], g.v[0]);
            const V3 nn = crs(e1, e2);
            const double s2 = dt(nn, nn);
            const V3 au = {crs(e2, nn).x / s2, crs(e2, nn).y / s2, crs(e2, nn).z / s2};
            const V3 av = {crs(nn, e1).x / s2, crs(nn, e1).y / s2, crs(nn, e1).z / s2};
            c.n[0] = r[0]; c.n[1] = r[1]; c.n[2] = r[2]; c.w = r[3];
            c.au[0] = static_cast<float>(au.x); c.au[1] = static_cast<float>(au.y); c.au[2] = static_cast<float>(au.z);
            c.av[0] = static_cast<float>(av.x); c.av[1] = static_cast<float>(av.y); c.av[2] = static_cast<float>(av.z);
            c.cu = static_cast<float>(-dt(au, g.v[0]));
            c.cv = static_cast<float>(-dt(av, g.v[0]));
            if (g.degenerate) {
                c.au[0] = c.au[1] = c.au[2] = c.av[0] = c.av[1] = c.av[2] = c.cu = c.cv = NAN;   // always kept
            } else {
                a_max_all = std::max(a_max_all, g.a_max);
                inv_2s_max_all = std::max(inv_2s_max_all, 1.0 / (2.0 * g.area2));
                diam2_2s_max_all = std::max(diam2_2s_max_all, g.diam * g.diam / (2.0 * g.area2));
            }
        }
    }

    // ---- margins of the barycentric test (large triangles)
    const double m_abs = 2.0 * std::sqrt(3.0) * r_org;   // bound on |o.n| + |w|
    CullConstants &cc = out.cc;
    cc.k2 = static_cast<float>(PT_MUT(k12) * (12.0 * kU * m_abs + 8.0 * kU * r_org));
    cc.k1 = static_cast<float>(PT_MUT(k12) * 40.0 * kU);
    cc.a_max = static_cast<float>(PT_MUT(a_max) * a_max * (1.0 + 1e-6));
    cc.m0 = static_cast<float>(PT_MUT(m0) * (std::fabs(eps) * inv_2s_max * 1.01 + 48.0 * kU * diam2_2s_max
                                             + 16.0 * kU * a_max * r_org * std::sqrt(3.0) + 1e-6));
    double tg = 4096.0 * r_org;
    if (a_max > 0) tg = std::min(tg, 1.0e6 / a_max);   // keep the reference's own area arithmetic meaningful
    cc.t_guard = static_cast<float>(tg);
    cc.m0_quad = static_cast<float>(static_cast<double>(cc.m0) + PT_MUT(quad_slack) * (quad_slack * 1.01 + 8.0 * kU * a_max * r_org));
    // the same margins over ALL triangles (pair pre-filter of big scenes)
    out.cc_all = cc;
    out.cc_all.a_max = static_cast<float>(PT_MUT(a_max) * a_max_all * (1.0 + 1e-6));
    out.cc_all.m0 = static_cast<float>(PT_MUT(m0) * (std::fabs(eps) * inv_2s_max_all * 1.01 + 48.0 * kU * diam2_2s_max_all
                                                     + 16.0 * kU * a_max_all * r_org * std::sqrt(3.0) + 1e-6));
    double tga = 4096.0 * r_org;
    if (a_max_all > 0) tga = std::min(tga, 1.0e6 / a_max_all);
    out.cc_all.t_guard = static_cast<float>(tga);

    // ---- where the emitters are (a material with Ke != 0 is the emissive lobe alone, material.h:58-106)
    auto emits = [&](uint32_t slot) {
        const uint32_t t = slot < out.slot_tri.size() ? out.slot_tri[slot] : kNoTriangle;
        if (t == kNoTriangle) return false;
        const float *m = &s.mat[10 * static_cast<size_t>(s.tri_mat[t])];
        return m[3] != 0.0f || m[4] != 0.0f || m[5] != 0.0f;
    };
    out.emis_clusters = 0;
    out.emis_large_w0 = 0xFFFFFFFFu;
    for (size_t c = 0; c < out.clusters.size(); ++c) {
        const ClusterDesc &cd = out.clusters[c];
        bool any = false;
        for (uint32_t k = 0; k < cd.n_tri; ++k) any = any || emits(cd.first_tri + k);
        if (c >= 32 || any) out.emis_clusters |= c < 32 ? (1u << c) : 0u;
        if (cd.kind == 1u && cd.n_tri <= static_cast<uint32_t>(kChunk)) {
            out.emis_large_w0 = 0;
            for (uint32_t k = 0; k < cd.n_tri; ++k) out.emis_large_w0 |= emits(cd.first_tri + k) ? (1u << k) : 0u;
        }
    }
    if (out.clusters.size() > 32) out.emis_clusters = 0xFFFFFFFFu;   // (small scenes have at most 10 clusters)
#ifdef PT_TEST_HOOKS
    // negative control of the shipped-path verification: forget one emitter of the large class
    if (g_cull_mutation.emis_drop && out.emis_large_w0 != 0xFFFFFFFFu) out.emis_large_w0 &= out.emis_large_w0 - 1u;
#endif
    out.emis_bvh = false;
    if (!out.bvh.empty())
        for (uint32_t k = 0; k < static_cast<uint32_t>(n_small_slots); ++k) out.emis_bvh = out.emis_bvh || emits(k);
}

}  // namespace pt
