// Host-side scene code (OBJ/MTL ingestion, device tables, culling tables) under AddressSanitizer + UBSan.
// Built and run by tests/test_host_sanitizers.py; argv[1] = models directory (with trailing slash), argv[2] = scratch directory.
#include "pt_scene.hpp"
#include <cstdio>
#include <fstream>
#include <random>
#include <string>
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const std::string models = argv[1], scratch = argv[2];
    pt::HostScene s; std::string err; bool io = false;
    if (!pt::load_obj(models, "Tor.obj", s, err, io)) { std::printf("load failed %s\n", err.c_str()); return 1; }
    pt::DeviceTables dt; pt::build_device_tables(s, dt);
    for (float eps : {1e-4f, 1e-2f, 0.0f, 1e-7f}) { pt::CullTables ct; pt::build_cull_tables(s, eps, ct); std::printf("eps %g: %zu clusters %zu spheres %zu bary\n", eps, ct.clusters.size(), ct.spheres.size(), ct.bary.size()); }
    // random scenes incl. degenerate triangles, > 2048 triangles (bary_all path)
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-9, 9), sz(-1, 1);
    for (int n : {0, 1, 7, 9, 65, 513, 3000}) {
        pt::HostScene h; h.mat.assign(10, 0.5f);
        for (int i = 0; i < n; ++i) {
            float p[3] = {u(rng), u(rng), u(rng)}, q[3], r[3];
            const float k = (i % 13 == 0) ? 8.0f : (i % 7 == 0 ? 0.0f : 0.4f);
            for (int c = 0; c < 3; ++c) { q[c] = p[c] + k * sz(rng); r[c] = p[c] + k * sz(rng); }
            pt::append_triangle(h, p, q, r, nullptr, 0);
        }
        pt::CullTables ct; pt::build_cull_tables(h, 1e-4f, ct);
        pt::DeviceTables d2; pt::build_device_tables(h, d2);
        std::printf("n %d: %zu clusters %zu spheres %zu bary %zu bary_all %zu slots %zu bvh\n", n, ct.clusters.size(), ct.spheres.size(), ct.bary.size(), ct.bary_all.size(), ct.slot_tri.size(), ct.bvh.size());
    }
    // malformed OBJ files
    const char *bad[] = {"f 1 2 3\n", "mtllib nope.mtl\nv 0 0 0\n", "v 0 0\nf 1//1 2 3\n", "usemtl x\nv 1 2 3\nv 1 2 4\nv 2 2 2\nf 1 2 9\n", ""};
    for (const char *b : bad) { std::ofstream(scratch + "b.obj") << b; pt::HostScene h; bool io2; std::string e; const bool ok = pt::load_obj(scratch, "b.obj", h, e, io2); std::printf("bad obj -> %d %s\n", ok, e.c_str()); }
    return 0;
}
