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
    out.cull.clear();
    out.geo = CullGeometry();
    out.geo.r_max = 20.0;   // the camera origin (0,0,-20), main.cpp:129
    for (int i = 0; i < T; ++i) {
        const float *r = &s.tri[14 * static_cast<size_t>(i)];
        ExactRec &e = out.exact[i];
        std::memcpy(e.plane, r, 16);
        std::memcpy(e.v0, r + 4, 12);
        e.square = r[13];
        std::memcpy(e.v1, r + 7, 12);
        e.material = s.tri_mat[i];
        std::memcpy(e.v2, r + 10, 12);
        e.pad = 0.0f;

        // Barycentric functions of the orthogonal projection onto the triangle's own plane, in double.
        const double v0[3] = {r[4], r[5], r[6]}, v1[3] = {r[7], r[8], r[9]}, v2[3] = {r[10], r[11], r[12]};
        const double e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
        const double e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
        const double e3[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
        const double nn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double s2 = nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2];
        const double area2 = std::sqrt(s2);   // parallelogram area S
        double au[3], av[3];
        au[0] = (e2[1] * nn[2] - e2[2] * nn[1]) / s2;
        au[1] = (e2[2] * nn[0] - e2[0] * nn[2]) / s2;
        au[2] = (e2[0] * nn[1] - e2[1] * nn[0]) / s2;
        av[0] = (nn[1] * e1[2] - nn[2] * e1[1]) / s2;
        av[1] = (nn[2] * e1[0] - nn[0] * e1[2]) / s2;
        av[2] = (nn[0] * e1[1] - nn[1] * e1[0]) / s2;
        CullRec c;
        c.n[0] = r[0]; c.n[1] = r[1]; c.n[2] = r[2]; c.w = r[3];
        for (int k = 0; k < 3; ++k) { c.au[k] = static_cast<float>(au[k]); c.av[k] = static_cast<float>(av[k]); }
        c.cu = static_cast<float>(-(au[0] * v0[0] + au[1] * v0[1] + au[2] * v0[2]));
        c.cv = static_cast<float>(-(av[0] * v0[0] + av[1] * v0[1] + av[2] * v0[2]));
        out.cull.push_back(c);

        auto len = [](const double *a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); };
        const double aw[3] = {au[0] + av[0], au[1] + av[1], au[2] + av[2]};
        const double amax = std::max(len(au), std::max(len(av), len(aw)));
        const double diam = std::max(len(e1), std::max(len(e2), len(e3)));
        if (std::isfinite(amax)) out.geo.a_max = std::max(out.geo.a_max, amax);
        if (area2 > 0) {
            out.geo.inv_2s_max = std::max(out.geo.inv_2s_max, 1.0 / (2.0 * area2));
            out.geo.diam2_2s_max = std::max(out.geo.diam2_2s_max, diam * diam / (2.0 * area2));
        }
        for (int k = 4; k < 13; ++k) out.geo.r_max = std::max(out.geo.r_max, static_cast<double>(std::fabs(r[k])));
    }
    // Pad to whole mask words; the kernel masks the padding bits off, the records only have to be readable.
    while (out.cull.size() % kChunk) {
        CullRec c;
        std::memset(&c, 0, sizeof c);
        out.cull.push_back(c);
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

CullConstants cull_constants(const CullGeometry &g, float eps) {
    // u = unit roundoff of binary32.  Derivation in DESIGN.md ("Culling: why it cannot reject a hit").
    const double u = 5.9604644775390625e-08;
    const double r = g.r_max + 1.0;                 // ray origins sit on surfaces, offset by eps*N
    const double m_abs = 2.0 * std::sqrt(3.0) * r;  // bound on |o.n| + |w|
    CullConstants c;
    c.k2 = static_cast<float>(12.0 * u * m_abs + 8.0 * u * r);
    c.k1 = static_cast<float>(40.0 * u);
    c.a_max = static_cast<float>(g.a_max * (1.0 + 1e-6));
    const double e_fp = 40.0 * u * g.diam2_2s_max;   // float error of the reference's area sum, already /(2S)
    c.m0 = static_cast<float>(std::fabs(static_cast<double>(eps)) * g.inv_2s_max * 1.01 + e_fp
                              + 16.0 * u * g.a_max * r * std::sqrt(3.0) + 1e-6);
    c.t_guard = static_cast<float>(4096.0 * r);
    return c;
}

}  // namespace pt
