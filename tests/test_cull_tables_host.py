"""CPU checks of the culling hierarchy the host builds (pt_scene.cpp: build_cull_tables): structure, containment, and
the property everything rests on -- a triangle the reference ACCEPTS for a ray is never culled for that ray.

The hierarchy lists the triangles in its own SLOT order (spatial grouping, independent of the file order); cluster
ranges, sphere-tree leaves and box-tree leaves are slots, `cull_layout()["slot_triangle"]` maps them back."""
import importlib
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

pt = importlib.import_module("path-tracing_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ray_sphere_keep(c, r2, o, d):
    """float64 model of sphere_keep() in pt_kernels.hip (the float32 rounding slack is part of r2)."""
    m = c - o
    b = np.maximum((m * d).sum(-1), 0.0)
    return ~((m * m).sum(-1) - b * b > r2)


def _layout(t):
    """slot -> (cluster, [sphere index of its ancestor at every level, top first ... level 0 last])."""
    out = {}
    for ci in range(len(t["kind"])):
        if t["kind"][ci] != 0:
            continue
        for k in range(t["n_tri"][ci]):
            chain = [t["data_off"][ci] + t["level_off"][ci][lv] + (k >> (3 * lv)) for lv in range(t["n_levels"][ci] - 1, -1, -1)]
            out[t["first_tri"][ci] + k] = (ci, chain)
    return out


@pytest.fixture(scope="module")
def tor(models_dir):
    return pt.Scene.load_obj(models_dir, "Tor.obj", device=-1)


def test_structure_tor(tor):
    t = tor.cull_tables()
    # torus: sphere tree.  The rest is ONE barycentric run: the file lists 2 wall triangles, the light's 2 triangles,
    # then 10 wall triangles, and a run of <= 2 small triangles between two large runs is absorbed into them.
    assert list(t["first_tri"]) == [0, 256] and list(t["n_tri"]) == [256, 14]
    assert list(t["kind"]) == [0, 1]
    assert t["n_levels"][0] == 3                    # 256 -> 32 -> 4 nodes
    assert list(t["level_off"][0][:3]) == [0, 256, 288]
    assert t["n_large"] == 14
    tri, _ = tor.triangles()
    st = tor.cull_layout()["slot_triangle"]
    assert sorted(st) == list(range(270))                  # a permutation of the triangles, no padding in a small scene
    assert sorted(st[:256]) == list(range(256))            # the torus (file triangles 0-255) is the sphere-tree cluster
    v = tri[st, 4:13].reshape(-1, 3, 3).astype(np.float64)   # vertices by SLOT
    lay = _layout(t)
    assert sorted(lay) == list(range(256))
    for i, (ci, chain) in lay.items():
        for sph in [t["cluster_sphere"][ci]] + [t["spheres"][k] for k in chain]:
            assert (((v[i] - sph[:3]) ** 2).sum(1) <= sph[3]).all()      # vertices inside every enclosing sphere
    # padding records can never keep a finite ray
    pads = t["spheres"][t["spheres"][:, 3] < 0]
    assert len(pads) > 0 and (pads[:, 3] <= -1e29).all()
    # the triangle spheres really are small compared with the octet and cluster spheres
    r_tri = np.sqrt([t["spheres"][chain[-1]][3] for _, chain in lay.values()])
    assert np.median(r_tri) < 1.0 and np.sqrt(t["cluster_sphere"][0][3]) < 4.5


def test_accepted_triangles_are_never_culled(tor, oracle_scene):
    """Every (ray, triangle) pair the reference's Triangle::Intersect accepts passes all three sphere levels."""
    import ctypes as C
    t = tor.cull_tables()
    lay = _layout(t)
    tri, _ = tor.triangles()
    st = tor.cull_layout()["slot_triangle"]
    slot_of = np.argsort(st)
    rng = np.random.default_rng(17)
    v = tri[:, 4:13].reshape(-1, 3, 3).astype(np.float64)
    # rays aimed at random points of small triangles (inside, on edges, just outside), from random origins
    small = np.array(sorted(st[k] for k in lay))
    n = 40000
    a = small[rng.integers(0, len(small), n)]
    w = rng.dirichlet([0.6, 0.6, 0.6], n) * rng.choice([1.0, 1.0, 1.001, 1.01], n)[:, None]
    target = (v[a] * w[:, :, None]).sum(1)
    org = rng.uniform([-9.5, -9.5, -20.5], [9.5, 9.5, 9.5], (n, 3))
    d = (target - org).astype(np.float32)
    inv = np.float32(1) / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32)
    d = d * inv[:, None]
    o32 = org.astype(np.float32)
    accepted = 0
    best = C.c_float()
    L = O.lib()
    for i in range(n):
        stage = L.orc_probe_intersect(oracle_scene.h, int(a[i]), o32[i].ctypes.data_as(C.POINTER(C.c_float)),
                                      d[i].ctypes.data_as(C.POINTER(C.c_float)), 1e-4, np.float32(np.inf), C.byref(best))
        if stage != 4:
            continue
        accepted += 1
        ci, chain = lay[int(slot_of[a[i]])]
        o64, d64 = o32[i].astype(np.float64), d[i].astype(np.float64)
        for sph in [t["cluster_sphere"][ci]] + [t["spheres"][k] for k in chain]:
            assert _ray_sphere_keep(sph[:3].astype(np.float64), float(sph[3]), o64, d64)
    assert accepted > 0.5 * n


def test_degenerate_and_tiny_triangles_are_always_kept(tmp_path):
    (tmp_path / "d.mtl").write_text("newmtl 0\nKd 1 1 1\n")
    (tmp_path / "d.obj").write_text("mtllib d.mtl\nv 0 0 0\nv 1 0 0\nv 2 0 0\nv 0 0.001 0\nv 0.001 0 0\nv 5 5 5\nv 5.1 5 5\nv 5 5.1 5\n"
                                    "usemtl 0\nf 1 2 3\nf 1 4 5\nf 6 7 8\n")
    s = pt.Scene.load_obj(str(tmp_path) + "/", "d.obj", device=-1)
    t = s.cull_tables()
    lay = _layout(t)
    # collinear triangle 0 and the 1e-6-area triangle 1 (area below a few eps) cannot be bounded: they land in the
    # barycentric class with NaN coefficients, which makes every cull comparison false (always kept)
    assert list(t["kind"]) == [0, 1] and list(t["n_tri"]) == [1, 2]      # the sphere-tree cluster first, the large class last
    st = s.cull_layout()["slot_triangle"]
    assert st[0] == 2 and sorted(st[1:]) == [0, 1]
    assert np.isnan(t["bary"][0, 4:]).all() and np.isnan(t["bary"][1, 4:]).all()
    assert np.isinf(t["cluster_sphere"][1][3])
    assert sorted(lay) == [0] and np.isfinite(t["spheres"][lay[0][1][-1]][3])


def test_light_as_its_own_cluster_without_absorption(models_dir):
    """The test build's `no_absorb` hook (a tuning knob of the table builder) keeps the light (a connected group of two
    small triangles) as a sphere cluster of its own instead of two more records of the large class."""
    H = pt.load_library(pt.TESTHOOKS_LIB_PATH)
    H.pt_test_set_mutation(b"no_absorb", 1.0)
    try:
        t = pt.Scene.load_obj(models_dir, "Tor.obj", device=-1, library=H).cull_tables()
    finally:
        H.pt_test_set_mutation(b"reset", 0.0)
    assert list(t["kind"]) == [0, 0, 1] and sorted(t["n_tri"][:2]) == [2, 256] and t["n_tri"][2] == 12
    assert sorted(t["n_levels"][:2]) == [1, 3] and t["n_large"] == 12


def test_tiny_run_between_walls_keeps_its_own_cluster(tmp_path):
    """Absorption is refused when the small triangles' barycentric gradients would inflate the class-wide margin."""
    (tmp_path / "w.mtl").write_text("newmtl 0\nKd 1 1 1\n")
    quad = lambda z: [f"v -9 -9 {z}", f"v 9 -9 {z}", f"v 9 9 {z}", f"v -9 9 {z}"]
    lines = ["mtllib w.mtl", "usemtl 0"] + quad(-9) + ["f 1 2 3", "f 1 3 4"]
    lines += ["v 0 0 0", "v 0.05 0 0", "v 0 0.05 0", "f 5 6 7"]                       # a 5-centimetre triangle (smaller ones count as degenerate)
    lines += quad(9) + ["f 8 9 10", "f 8 10 11"]
    lines += ["v 0 0 1", "v 3 0 1", "v 0 3 1", "f 12 13 14"]                            # a 3-unit triangle: absorbed
    lines += quad(5) + ["f 15 16 17", "f 15 17 18"]
    (tmp_path / "w.obj").write_text("\n".join(lines) + "\n")
    t = pt.Scene.load_obj(str(tmp_path) + "/", "w.obj", device=-1).cull_tables()
    assert list(t["kind"]) == [0, 1] and list(t["n_tri"]) == [1, 7]     # the 5-cm triangle keeps its cluster, the 3-unit one is absorbed


def test_tables_depend_on_eps(tor):
    a, b = tor.cull_tables(1e-4), tor.cull_tables(1e-2)
    assert (b["spheres"][:, 3] >= a["spheres"][:, 3]).all()
    assert b["constants"]["m0"] > a["constants"]["m0"]


def test_quads_of_the_room_are_fused_and_never_cull_an_accepted_hit(tor, oracle_scene):
    """Large class: consecutive coplanar triangle pairs share one quad record (plane, alpha row, beta row).  Every
    (ray, wall triangle) pair the reference accepts must satisfy the quad test with the margins the kernel uses."""
    import ctypes as C
    t = tor.cull_tables()
    large = np.flatnonzero(t["kind"] == 1)
    assert [int(t["level_off"][c][1]) for c in large] == [0b1010101010101]     # quad mask of word 0: walls and light, every pair fused
    k1, k2, a_max, m0 = (t["constants"][k] for k in ("k1", "k2", "a_max", "m0"))
    tri, _ = tor.triangles()
    slot_tri = tor.cull_layout()["slot_triangle"]
    v = tri[:, 4:13].reshape(-1, 3, 3).astype(np.float64)
    rng = np.random.default_rng(23)
    L = O.lib()
    best = C.c_float()
    checked = 0
    for c in large:
        first, n, off, mask = t["first_tri"][c], t["n_tri"][c], t["data_off"][c], int(t["level_off"][c][1])
        for k in range(0, n, 2):
            assert (mask >> k) & 1
            rec = t["bary"][off + k].astype(np.float64)
            for half in (0, 1):
                ti = int(slot_tri[first + k + half])      # slot -> original triangle
                w = rng.dirichlet([0.5, 0.5, 0.5], 3000) * rng.choice([1.0, 1.0, 1.0005], 3000)[:, None]   # incl. edges / just outside
                target = (v[ti] * w[:, :, None]).sum(1)
                org = rng.uniform([-9.5, -9.5, -20.5], [9.5, 9.5, 9.5], (3000, 3))
                d = (target - org).astype(np.float32)
                inv = np.float32(1) / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32)
                d = d * inv[:, None]
                o32 = org.astype(np.float32)
                for i in range(len(o32)):
                    st = L.orc_probe_intersect(oracle_scene.h, int(ti), o32[i].ctypes.data_as(C.POINTER(C.c_float)),
                                               d[i].ctypes.data_as(C.POINTER(C.c_float)), 1e-4, np.float32(np.inf), C.byref(best))
                    if st != 4:
                        continue
                    checked += 1
                    o64, d64 = o32[i].astype(np.float64), d[i].astype(np.float64)
                    num, den = rec[0:3] @ o64 + rec[3], rec[0:3] @ d64
                    tt = -num / den
                    P = o64 + tt * d64
                    al, be = rec[4:7] @ P + rec[7], rec[8:11] @ P + rec[11]
                    e = min(be, al - be, 1 - al) if half == 0 else min(al, be - al, 1 - be)
                    et = (k1 * abs(tt) + k2) / abs(den)
                    assert e >= -(a_max * et + m0) and tt >= -et, (ti, e, tt)
    assert checked > 10000


def _shuffled_copy(models_dir, tmp, seed):
    """Tor.obj with its faces in random order (each face keeps its material: usemtl is re-emitted per face)."""
    lines = open(models_dir + "Tor.obj").read().split("\n")
    head, faces, mtl = [], [], None
    for l in lines:
        if l.startswith("usemtl"):
            mtl = l
        elif l.startswith("f "):
            faces.append((mtl, l))
        else:
            head.append(l)
    perm = np.random.default_rng(seed).permutation(len(faces))
    out = head + [x for k in perm for x in faces[k]]
    open(tmp + "shuffled.obj", "w").write("\n".join(out) + "\n")
    open(tmp + "Tor.mtl", "w").write(open(models_dir + "Tor.mtl").read())
    return perm


def test_hierarchy_does_not_depend_on_the_file_order(tmp_path, models_dir, tor):
    d = str(tmp_path) + "/"
    perm = _shuffled_copy(models_dir, d, 4)
    sh = pt.Scene.load_obj(d, "shuffled.obj", device=-1)
    a, b = tor.cull_tables(), sh.cull_tables()
    for k in ("kind", "first_tri", "n_tri", "n_levels"):
        assert list(a[k]) == list(b[k])
    assert np.array_equal(a["spheres"].view(np.uint32), b["spheres"].view(np.uint32))       # the same tree, bit for bit
    assert np.array_equal(a["bary"].view(np.uint32), b["bary"].view(np.uint32))
    sa, sb = tor.cull_layout()["slot_triangle"], sh.cull_layout()["slot_triangle"]
    ta, tb = tor.triangles()[0], sh.triangles()[0]
    assert np.array_equal(ta[sa].view(np.uint32), tb[sb].view(np.uint32))                  # slot k holds the same geometry
    assert np.array_equal(perm[sb], sa)                                                       # ... i.e. the same triangle


@pytest.mark.parametrize("bvh_mode", [0, 1], ids=["uniform-depth", "sah-collapsed"])
def test_box_tree_never_drops_the_chain_above_a_hit(tmp_path, bvh_mode, request):
    """Big scenes: the chain of box-tree nodes above the triangle the reference hits survives the kernel's slab test
    (numpy restatement in float32, tests/bvh_emulation.py) for the tightest t_best the walk can hold: the hit's own t.
    Both tree builders (pt_scene.cpp: build_bvh, build_bvh_sah), chosen through the test-hook build."""
    import bvh_emulation as B
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_replicated_scene as M
    d = str(tmp_path) + "/"
    n_tri = M.generate(os.path.join(ROOT, "models"), d, "x9.obj", 9)
    assert n_tri > pt.BIG_SCENE_TRIANGLES
    hooks = pt.load_library(pt.TESTHOOKS_LIB_PATH)
    hooks.pt_test_set_mutation(b"reset", 0.0)
    hooks.pt_test_set_mutation(b"bvh_mode", float(bvh_mode))     # (the test-hook build rebuilds the hierarchy at every call:
    request.addfinalizer(lambda: hooks.pt_test_set_mutation(b"reset", 0.0))   # the knob stays set until the test is over)
    g = pt.Scene.load_obj(d, "x9.obj", device=-1, library=hooks)
    o = O.Scene.load(d, "x9.obj")
    lay = g.cull_layout()
    t, fl = B.decode(lay["bvh"]), lay["bvh_inner_nodes"]
    st = lay["slot_triangle"]
    assert len(lay["bvh"]) > fl > 0 and sorted(st[st >= 0]) == list(range(n_tri))
    assert (len(lay["bvh"]), fl) == ((491, 65) if bvh_mode == 0 else (516, 110))     # (the two builders' trees for this scene)
    n_tree_slots = (len(lay["bvh"]) - fl) * 8
    assert (st[n_tree_slots:] >= 0).all()                          # padding only inside the tree's leaves
    par, pos = B.parents(t, fl)
    assert (par[1:] >= 0).all() and par[0] == -1                   # one root, every other node has a parent
    slot_of = np.full(n_tri, -1)
    slot_of[st[st >= 0]] = np.flatnonzero(st >= 0)
    tri, _ = o.triangles()
    rng = np.random.default_rng(8)
    n = 60000
    v = tri[:, 4:13].reshape(-1, 3, 3).astype(np.float64)
    a = rng.integers(0, n_tri, n)
    w = rng.dirichlet([1, 1, 1], n)
    src = ((v[a] * w[:, :, None]).sum(1) + tri[a, 0:3] * 1e-4).astype(np.float32)
    dd = rng.normal(size=(n, 3)).astype(np.float32)
    dd[::7, 0] = 0                                                  # some axis-parallel components (the 1e-30 substitution)
    inv = np.float32(1) / np.sqrt((dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2], dtype=np.float32)
    dd = (dd * inv[:, None]).astype(np.float32)
    hi, ht, nan = o.closest_hits(src, dd)
    ok = (hi >= 0) & ~nan
    ok &= slot_of[np.maximum(hi, 0)] < n_tree_slots                 # hits on triangles of the tree (not the walls)
    ro, rd, tb, sl = src[ok], dd[ok], ht[ok], slot_of[hi[ok]]
    assert len(sl) > 10000
    leaf_of_group = np.full(n_tree_slots // 8, -1)                  # leaf node that holds slots 8 g ... 8 g + 7
    leaf_of_group[t["base"][t["leaf"]]] = np.flatnonzero(t["leaf"])
    assert (leaf_of_group >= 0).all() and t["leaf"].sum() == len(lay["bvh"]) - fl
    node, child = leaf_of_group[sl // 8], sl % 8
    levels = 0
    while len(node):
        kept = B.children_kept(t, node, ro, rd, tb, 5e-7)
        assert kept[np.arange(len(node)), child].all(), f"a node of level {levels} above a hit was dropped"
        # the packed half-precision form of the test (pt_kernels.hip: box_children_kept_h, built with -DPT_BOX_F16=1) on the same items
        kept_h = B.children_kept_f16(t, node, ro, rd, tb)
        assert kept_h[np.arange(len(node)), child].all(), f"half precision: a node of level {levels} above a hit was dropped"
        child, node = pos[node], par[node]
        live = node >= 0
        node, child, ro, rd, tb = node[live], child[live], ro[live], rd[live], tb[live]
        levels += 1
    assert levels >= 3


def test_big_scene_keeps_its_few_emitters_in_the_large_class(tmp_path, models_dir):
    """pt_scene.cpp: a big scene's emitters (<= 8 triangles: the light of a room) join the large class whatever their size,
    so that the kernel's last-segment test has their records; many emissive triangles stay under the box tree."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_replicated_scene as M
    d = str(tmp_path) + "/"
    n_tri = M.generate(models_dir, d, "x9.obj", 9)
    def large_class(scene):
        lay = scene.cull_layout()
        st = lay["slot_triangle"]
        return set(int(t) for t in st[(len(lay["bvh"]) - lay["bvh_inner_nodes"]) * 8:] if t >= 0)
    light = large_class(pt.Scene.load_obj(d, "x9.obj", device=-1))
    assert len(light) == 14 and all(t >= n_tri - 14 for t in light)          # the room: 12 wall triangles and the light's 2
    # the same scene with an emissive torus material: 2 + 9 * 256 emitters, none of them moved
    text = open(d + "Tor.mtl").read().replace("newmtl 4\n", "newmtl 4\nKe 0.8 0.6 0.2\n")
    lines, cur, seen = [], None, False
    for line in text.split("\n"):
        tok = line.split()
        if tok and tok[0] == "newmtl":
            cur, seen = tok[1], False
        if tok and tok[0] == "Ke" and cur == "4":
            if seen:
                continue                                                     # keep the first Ke line of the material only
            seen = True
        lines.append(line)
    open(d + "Tor.mtl", "w").write("\n".join(lines))
    many = large_class(pt.Scene.load_obj(d, "x9.obj", device=-1))
    assert len(many) == 12 and many < light
