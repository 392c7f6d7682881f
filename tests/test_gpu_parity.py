"""GPU parity: the HIP integrator (through the C ABI) against the CPU oracle on identical seeds.

Bar: BIT-EXACT accumulators (sum, sum2 as raw float bits, count as ints) and identical segment counts, for every
configuration below.  The arithmetic is float32, but every operation that decides a result is IEEE and unfused on both
sides, so no tolerance is needed; the 8-bit image is then identical by construction.
"""
import importlib
import os

import numpy as np
import pytest

import oracle_lib as O

pt = importlib.import_module("path-tracing_amd")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_scene(models_dir):
    assert pt.device_count() >= 1, "no HIP device: the integrator has no CPU fallback"
    return pt.Scene.load_obj(models_dir, "Tor.obj", device=0)


def _same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def _compare(gpu_scene, oracle_scene, W, H, spp, mrr, **kw):
    s, s2, c, st = gpu_scene.render_host(W, H, spp, mrr, **kw)
    rs, rs2, rc, rst = O.render(oracle_scene, W, H, spp, mrr, rng=O.RNG_COUNTER, trig=O.TRIG_PORTABLE, **kw)
    assert st["samples_traced"] == rst["samples_traced"]
    assert st["segments"] == rst["segments"]
    assert st["contributing"] == rst["contributing"]
    assert st["misses"] == rst["misses"]
    assert np.array_equal(c, rc)
    bad = np.flatnonzero((s.view(np.uint32) != rs.view(np.uint32)).any(axis=1))
    assert bad.size == 0, f"{bad.size} pixels differ, first {bad[:5]}"
    assert _same(s2, rs2)
    return st, rst


CASES = [
    (64, 64, 4, 3, {}),                       # reference config "64x64 -RPP 4 -MRR 3" geometry
    (96, 64, 6, 8, {}),
    (37, 19, 9, 8, {}),                       # ragged tiles in both directions
    (8, 8, 32, 8, {"seed": 7}),
    (16, 16, 8, 1, {}),                       # MRR 1: primary segment only
    (40, 24, 20, 8, {"error": 0.001}),        # adaptive sampling on (main.cpp:118-125)
    (40, 24, 20, 8, {"error": 0.5}),          # adaptive skip fires for nearly every lit pixel
    (33, 17, 5, 8, {"eps": 1e-3}),
    (256, 256, 4, 3, {"error": 0.001}),       # BASELINE config 1 geometry
]


@pytest.mark.parametrize("W,H,spp,mrr,kw", CASES, ids=[f"{c[0]}x{c[1]}x{c[2]}m{c[3]}{c[4]}" for c in CASES])
def test_frame_bit_exact(gpu_scene, oracle_scene, W, H, spp, mrr, kw):
    st, rst = _compare(gpu_scene, oracle_scene, W, H, spp, mrr, **kw)
    assert st["exact_tests"] < 0.05 * st["segments"] * 270   # the cull really culls


# The launches a caller WITHOUT pt_render_stats gets run another instantiation: two pixels per lane (16 x 8 tiles, 128 rays per
# wave-segment, 7-bit ray ids in the work queues), accumulators read-modify-written in memory, the emitter-first last segment.
# Same bar: the oracle's bits.  Widths around the 16-pixel tile, one-pixel images, adaptive sampling, and passes added in slices
# (a later slice finds the earlier one's sums in memory).
SHIPPED_CASES = [(1, 1, 40, 8, {}), (7, 3, 30, 8, {}), (8, 9, 20, 8, {}), (9, 8, 20, 3, {}), (15, 5, 16, 8, {}), (16, 8, 16, 8, {}),
                 (17, 33, 8, 8, {"error": 0.001}), (31, 7, 12, 2, {}), (37, 19, 9, 8, {}), (96, 64, 6, 8, {}),
                 (40, 24, 24, 8, {"error": 0.5}), (104, 50, 12, 8, {"error": 0.001, "seed": 7}), (250, 130, 5, 8, {"eps": 1e-3}),
                 # adaptive sampling deep into the passes where it bites: the two-pixel kernel runs batches of the tile's pixels
                 (48, 40, 40, 8, {"error": 0.02}), (33, 17, 64, 4, {"error": 0.005, "seed": 3}), (64, 16, 30, 8, {"error": 0.5, "seed": 9}),
                 (16, 8, 50, 8, {"error": 0.01})]


@pytest.fixture(scope="module")
def hooks_lib():
    L = pt.load_library(pt.TESTHOOKS_LIB_PATH)      # the product's kernels; its C API can pin the tile width (test hook)
    L.pt_test_set_mutation(b"reset", 0.0)
    yield L
    L.pt_test_set_mutation(b"reset", 0.0)


@pytest.mark.parametrize("W,H,spp,mrr,kw", SHIPPED_CASES, ids=[f"{c[0]}x{c[1]}x{c[2]}m{c[3]}{c[4]}" for c in SHIPPED_CASES])
def test_statistics_free_instantiation_bit_exact(gpu_scene, oracle_scene, hooks_lib, models_dir, W, H, spp, mrr, kw):
    rs, rs2, rc, _ = O.render(oracle_scene, W, H, spp, mrr, rng=O.RNG_COUNTER, trig=O.TRIG_PORTABLE, **kw)
    # The library picks 16 x 8 tiles (two pixels per lane) only when they fill the chip, so frames of this size would run the
    # 8 x 8 variant: the variants are pinned in turn through the test-hook build (same kernels; 3 = 16 x 8, and 32 x 8 for the
    # batches of adaptive launches), then the product's own choice.
    for width_mode in (3.0, 2.0, 1.0):
        hooks_lib.pt_test_set_mutation(b"tile_width", width_mode)
        try:
            h = pt.Scene.load_obj(models_dir, "Tor.obj", device=0, library=hooks_lib)
            s, s2, c, _ = h.render_host(W, H, spp, mrr, want_stats=False, **kw)
        finally:
            hooks_lib.pt_test_set_mutation(b"reset", 0.0)
        assert np.array_equal(c, rc) and _same(s, rs) and _same(s2, rs2), width_mode
    hooks_lib.pt_test_set_mutation(b"tile_width", 2.0)      # the slices and the band below: the wide variant again
    s, s2, c, _ = gpu_scene.render_host(W, H, spp, mrr, want_stats=False, **kw)
    assert np.array_equal(c, rc) and _same(s, rs) and _same(s2, rs2)
    gpu_scene = pt.Scene.load_obj(models_dir, "Tor.obj", device=0, library=hooks_lib)
    # the same frame in three pass slices on a device-resident session (odd slice lengths, a one-pass slice)
    ses = pt.Session(gpu_scene, W, H)
    cuts = [0, min(1, spp), min(1 + (spp - 1) // 3, spp), spp]
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            ses.render(a, b - a, mrr, **kw)
    s, s2, c = ses.read()
    ses.close()
    assert np.array_equal(c, rc) and _same(s, rs) and _same(s2, rs2)
    # and as a row band of a taller image (the band's first row is not a multiple of the tile height)
    if H >= 5:
        r0, r1 = 2, H - 1
        bs, bs2, bc, _ = gpu_scene.render_host(W, H, spp, mrr, rows=(r0, r1), want_stats=False, **kw)
        assert np.array_equal(bc, rc[r0 * W:r1 * W]) and _same(bs, rs[r0 * W:r1 * W]) and _same(bs2, rs2[r0 * W:r1 * W])
    hooks_lib.pt_test_set_mutation(b"reset", 0.0)


def test_empty_and_degenerate_calls(gpu_scene):
    s, s2, c, st = gpu_scene.render_host(16, 16, 0, 8)
    assert st["segments"] == 0 and not s.any() and not c.any()
    s, s2, c, st = gpu_scene.render_host(16, 16, 4, 8, rows=(5, 5))
    assert s.shape == (0, 3)
    s, s2, c, st = gpu_scene.render_host(16, 16, 4, 0)      # MRR 0: no ray is ever valid (ray.h:52-54)
    assert st["segments"] == 0 and st["samples_traced"] == 16 * 16 * 4


def test_row_bands_equal_full_frame(gpu_scene):
    W, H, spp, mrr = 48, 40, 6, 8
    s, s2, c, _ = gpu_scene.render_host(W, H, spp, mrr)
    for r0, r1 in [(0, 13), (13, 14), (14, 40)]:
        bs, bs2, bc, _ = gpu_scene.render_host(W, H, spp, mrr, rows=(r0, r1))
        assert _same(bs, s[r0 * W:r1 * W]) and _same(bs2, s2[r0 * W:r1 * W]) and np.array_equal(bc, c[r0 * W:r1 * W])


def test_pass_slices_equal_one_call(gpu_scene):
    W, H, mrr = 32, 24, 8
    s, s2, c, _ = gpu_scene.render_host(W, H, 24, mrr, error=0.001)
    acc = None
    for p0, n in [(0, 5), (5, 7), (12, 12)]:
        a = gpu_scene.render_host(W, H, n, mrr, error=0.001, pass_begin=p0, accum=acc)
        acc = a[:3]
    assert _same(acc[0], s) and _same(acc[1], s2) and np.array_equal(acc[2], c)


def test_synthetic_scene_with_all_material_kinds(gpu_scene):
    # one emitter, a pure mirror (Ns=1000 -> glossy only), a pure diffuse (Ns=0), a black absorber (no Ke, Kd=0),
    # and a two-lobe material; closed box so paths keep bouncing
    rng = np.random.default_rng(3)
    import tempfile
    d = tempfile.mkdtemp() + "/"
    mtl = ["newmtl 0\nKe 1 1 1\nKd 0.9 0.8 0.7\n", "newmtl 1\nNs 1000\nKs 0.9 0.9 0.9\nKd 0 0 0\n",
           "newmtl 2\nNs 0\nKd 0.7 0.7 0.7\n", "newmtl 3\nNs 500\nKs 0.5 0.6 0.7\nKd 0.3 0.9 0.3\n",
           "newmtl 4\nNs 0\nKd 0 0 0\n"]
    open(d + "b.mtl", "w").write("".join(mtl))
    L = 8.0
    corners = [(-L, -L, -25), (L, -L, -25), (L, L, -25), (-L, L, -25), (-L, -L, 6), (L, -L, 6), (L, L, 6), (-L, L, 6)]
    faces = [((0, 1, 2, 3), 2), ((4, 5, 6, 7), 3), ((0, 1, 5, 4), 1), ((3, 2, 6, 7), 0), ((0, 3, 7, 4), 2), ((1, 2, 6, 5), 4)]
    lines = ["mtllib b.mtl"] + ["v %f %f %f" % c for c in corners]
    for (a, b, c_, e), m in faces:
        lines += [f"usemtl {m}", f"f {a+1} {b+1} {c_+1}", f"f {a+1} {c_+1} {e+1}"]
    # a few random small triangles floating inside
    for k in range(20):
        p = rng.uniform(-4, 4, 3) + [0, 0, -2]
        q = p + rng.uniform(-1.5, 1.5, 3)
        r = p + rng.uniform(-1.5, 1.5, 3)
        n = len(corners) + 3 * k
        lines += ["v %f %f %f" % tuple(p), "v %f %f %f" % tuple(q), "v %f %f %f" % tuple(r), f"usemtl {k % 5}",
                  f"f {n+1} {n+2} {n+3}"]
    open(d + "b.obj", "w").write("\n".join(lines) + "\n")
    g = pt.Scene.load_obj(d, "b.obj", device=0)
    o = O.Scene.load(d, "b.obj")
    st, rst = _compare(g, o, 48, 32, 12, 8)
    assert st["contributing"] > 100


# ---- large scenes: many clusters, several flushes of the candidate slots, several batches of the work queues ----
def _replicated(tmp, instances):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_replicated_scene as M
    d = str(tmp) + "/"
    name = f"TorX{instances}.obj"
    n = M.generate(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models"), d, name, instances)
    return d, name, n


@pytest.mark.parametrize("instances,W,H,spp", [(8, 64, 48, 6), (64, 56, 40, 3)])
def test_replicated_scene_bit_exact(tmp_path, instances, W, H, spp):
    d, name, n = _replicated(tmp_path, instances)      # BASELINE config 5 geometry (x64 -> 16 398 triangles)
    g = pt.Scene.load_obj(d, name, device=0)
    o = O.Scene.load(d, name)
    assert g.counts()[0] == n == o.n_tri
    st, rst = _compare(g, o, W, H, spp, 8)
    assert st["exact_tests"] < 0.01 * st["segments"] * n


@pytest.mark.parametrize("instances,W,H,spp,kw", [(8, 48, 32, 100, {"error": 0.05}), (64, 40, 16, 40, {"error": 0.2, "seed": 3})], ids=["x8", "x64"])
def test_box_tree_batches_of_adaptive_sampling(tmp_path, hooks_lib, instances, W, H, spp, kw):
    """Adaptive sampling on a scene under the box tree: the statistics-free kernel runs batches of 64 of its tile's 128 / 256 pixels,
    each at its own next pass (pt_kernels.hip "Batches"; one ray slot per lane here) -- the oracle's bits for every tile width (1 = the
    plain 8 x 8 kernel that sits passes out)."""
    d, name, n = _replicated(tmp_path, instances)
    o = O.Scene.load(d, name)
    rs, rs2, rc, rst = O.render(o, W, H, spp, 8, rng=O.RNG_COUNTER, trig=O.TRIG_PORTABLE, **kw)
    assert rst["samples_traced"] < 0.95 * W * H * spp and rc.sum() > 0     # the adaptive skip bites (pixels without a contribution never sit out)
    for width_mode in (3.0, 2.0, 1.0):
        hooks_lib.pt_test_set_mutation(b"tile_width", width_mode)
        try:
            h = pt.Scene.load_obj(d, name, device=0, library=hooks_lib)
            assert len(h.cull_layout()["bvh"]) > 0
            s, s2, c, _ = h.render_host(W, H, spp, 8, want_stats=False, **kw)
        finally:
            hooks_lib.pt_test_set_mutation(b"reset", 0.0)
        assert np.array_equal(c, rc) and _same(s, rs) and _same(s2, rs2), width_mode


def test_many_candidates_per_ray(tmp_path):
    # a deep stack of large coplanar-ish sheets in front of the camera: every ray has dozens of candidate triangles,
    # so the (ray, triangle) queue needs several batches and the slots several flushes
    lines = ["mtllib s.mtl"]
    open(str(tmp_path / "s.mtl"), "w").write("newmtl 0\nKe 1 1 1\nKd 1 1 1\nnewmtl 1\nNs 200\nKs 0.8 0.8 0.8\nKd 0.6 0.6 0.6\n")
    nv = 0
    k = 0
    for layer in range(90):
        z = -5.0 + 0.11 * layer
        s = 9.0
        lines += [f"v {-s} {-s} {z}", f"v {s} {-s} {z}", f"v {s} {s} {z}", f"v {-s} {s} {z}"]
        lines += [f"usemtl {1 if layer % 7 else 0}", f"f {nv+1} {nv+2} {nv+3}", f"f {nv+1} {nv+3} {nv+4}"]
        nv += 4
    # plus a cloud of small triangles so that sphere clusters exist too
    rng = np.random.default_rng(11)
    for t in range(700):
        p = rng.uniform(-6, 6, 3) + [0, 0, -9]
        q = p + rng.uniform(-0.4, 0.4, 3)
        r = p + rng.uniform(-0.4, 0.4, 3)
        lines += ["v %f %f %f" % tuple(p), "v %f %f %f" % tuple(q), "v %f %f %f" % tuple(r), f"usemtl {t % 2}",
                  f"f {nv+1} {nv+2} {nv+3}"]
        nv += 3
    open(str(tmp_path / "s.obj"), "w").write("\n".join(lines) + "\n")
    d = str(tmp_path) + "/"
    g = pt.Scene.load_obj(d, "s.obj", device=0)
    o = O.Scene.load(d, "s.obj")
    st, rst = _compare(g, o, 40, 32, 8, 8)
    assert st["exact_tests"] > 20 * st["segments"]      # really many candidates per ray


def test_tiny_and_empty_scenes(tmp_path):
    d = str(tmp_path) + "/"
    open(d + "m.mtl", "w").write("newmtl 0\nKe 1 1 1\nKd 0.5 0.25 1\nnewmtl 1\nNs 0\nKd 0.8 0.8 0.8\n")
    # no triangle at all: every ray misses
    open(d + "empty.obj", "w").write("mtllib m.mtl\nv 0 0 0\n")
    g = pt.Scene.load_obj(d, "empty.obj", device=0)
    s, s2, c, st = g.render_host(24, 16, 3, 8)
    assert st["segments"] == st["misses"] == 24 * 16 * 3 and not c.any()
    i, t = g.trace_rays(np.zeros((5, 3), np.float32), np.tile(np.array([0, 0, 1], np.float32), (5, 1)))
    assert (i == -1).all() and np.isinf(t).all()
    # one emissive triangle in front of the camera; one diffuse triangle behind it
    open(d + "one.obj", "w").write("mtllib m.mtl\nv -3 -3 0\nv 3 -3 0\nv 0 4 0\nusemtl 0\nf 1 3 2\n")
    open(d + "two.obj", "w").write("mtllib m.mtl\nv -3 -3 0\nv 3 -3 0\nv 0 4 0\nv -9 -9 5\nv 9 -9 5\nv 0 9 5\nusemtl 0\nf 1 3 2\nusemtl 1\nf 4 6 5\n")
    for name in ("one.obj", "two.obj"):
        g = pt.Scene.load_obj(d, name, device=0)
        o = O.Scene.load(d, name)
        st, rst = _compare(g, o, 40, 40, 5, 8)
        assert st["contributing"] > 0


def test_node_stack_overflow_paths(tmp_path, hooks_lib, request):
    """1 800 small triangles packed into a ball of radius 1: every ray through it passes nearly every sphere of the
    tree, so a walk round produces far more children than the 96-entry node stack and the 144-entry pair queue of the
    small-scene kernel hold -- the partial-commit and forced-commit paths run, and must not change any result.
    (1 800 triangles are beyond the shipped small / big switch since round 4: the test hook keeps them on the sphere-tree path.)"""
    hooks_lib.pt_test_set_mutation(b"big_threshold", 16384.0)
    request.addfinalizer(lambda: hooks_lib.pt_test_set_mutation(b"reset", 0.0))
    rng = np.random.default_rng(99)
    d = str(tmp_path) + "/"
    open(d + "m.mtl", "w").write("newmtl 0\nKe 1 1 1\nKd 1 1 1\nnewmtl 1\nNs 100\nKs 0.9 0.9 0.9\nKd 0.7 0.7 0.7\n")
    lines = ["mtllib m.mtl"]
    nv = 0
    for k in range(1800):
        c = rng.normal(size=3) * 0.35 + [0, 0, -8]
        p, q, r = (c + rng.normal(size=3) * 0.25 for _ in range(3))
        lines += ["v %f %f %f" % tuple(p), "v %f %f %f" % tuple(q), "v %f %f %f" % tuple(r), f"usemtl {1 if k % 9 else 0}",
                  f"f {nv+1} {nv+2} {nv+3}"]
        nv += 3
    open(d + "ball.obj", "w").write("\n".join(lines) + "\n")
    g = pt.Scene.load_obj(d, "ball.obj", device=0, library=hooks_lib)
    o = O.Scene.load(d, "ball.obj")
    t = g.cull_tables()
    assert list(t["kind"]) == [0] and t["n_levels"][0] == 4           # one small-triangle cluster, 1800 -> 225 -> 29 -> 4
    # rays aimed at the ball from all around, 64 per wave all passing through the dense core
    n = 64 * 300
    org = (rng.normal(size=(n, 3)) * 4 + [0, 0, -8]).astype(np.float32)
    tgt = (rng.normal(size=(n, 3)) * 0.2 + [0, 0, -8]).astype(np.float32)
    dirs = tgt - org
    inv = np.float32(1) / np.sqrt((dirs[:, 0] * dirs[:, 0] + dirs[:, 1] * dirs[:, 1]) + dirs[:, 2] * dirs[:, 2], dtype=np.float32)
    dirs = (dirs * inv[:, None]).astype(np.float32)
    gi, gt = g.trace_rays(org, dirs)
    ri, rt, nan_seen = o.closest_hits(org, dirs)
    assert not nan_seen.any()
    assert np.array_equal(gi, ri) and np.array_equal(gt.view(np.uint32), rt.view(np.uint32))
    assert (ri >= 0).mean() > 0.9
    st, rst = _compare(g, o, 64, 48, 3, 8)                             # camera looks straight at the ball
    assert st["partial_commit_rounds"] > 100                          # the overflow paths really ran
    assert st["exact_tests"] > 30 * st["segments"] * 0.05              # dozens of candidates for the rays that reach it


def test_device_pointer_path_on_two_streams(gpu_scene):
    """pt_render_device with caller-owned device buffers (the bench / integration path), launched back to back on two
    different streams without a host synchronisation in between: a scene's launches share its scheduler words, so the
    library has to order them on the device; both frames must equal the pt_render_host frame bit for bit."""
    import torch
    W, H, spp, mrr = 320, 200, 12, 8
    n = W * H
    dev = torch.device("cuda", 0)
    p = pt.RenderParams(W, H, 0, H, 0, spp, mrr, 1e-4, -1.0, 42)
    bufs = [torch.zeros(7 * n, dtype=torch.float32, device=dev) for _ in range(3)]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    torch.cuda.synchronize(dev)
    for k in range(3):   # third launch: back on the first stream
        b = bufs[k]
        gpu_scene.render_device(p, b.data_ptr(), b.data_ptr() + 12 * n, b.data_ptr() + 24 * n,
                                stream=streams[k % 2].cuda_stream)
    torch.cuda.synchronize(dev)
    s, s2, c, _ = gpu_scene.render_host(W, H, spp, mrr)
    for b in bufs:
        h = b.cpu().numpy()
        assert _same(h[:3 * n], s.ravel()) and _same(h[3 * n:6 * n], s2.ravel())
        assert np.array_equal(h[6 * n:].view(np.int32), c)


def test_sessions_keep_the_band_on_the_device(gpu_scene, oracle_scene):
    """pt_session_*: pass slices added to device-resident accumulators equal one pt_render_host call (and the oracle) bit
    for bit; a session can be read at any point, cleared and reused; mismatching parameters are refused."""
    W, H, mrr = 72, 40, 8
    s, s2, c, _ = gpu_scene.render_host(W, H, 30, mrr, error=0.001)
    ses = pt.Session(gpu_scene, W, H)
    for p0, n in [(0, 4), (4, 11), (15, 15)]:
        ses.render(p0, n, mrr, error=0.001)
    a, a2, ac = ses.read()
    assert _same(a, s) and _same(a2, s2) and np.array_equal(ac, c)
    rs, rs2, rc, _ = O.render(oracle_scene, W, H, 30, mrr, error=0.001)
    assert _same(a, rs) and np.array_equal(ac, rc)
    b, b2, bc = ses.read()                                   # reading does not disturb the accumulators
    assert _same(b, s) and np.array_equal(bc, c)
    st = ses.render(30, 2, mrr, error=0.001, want_stats=True)   # statistics of one slice
    assert st["samples_traced"] <= W * H * 2 and st["segments"] > 0
    ses.clear()
    z, z2, zc = ses.read()
    assert not z.any() and not z2.any() and not zc.any()
    ses.render(0, 30, mrr, error=0.001)
    a, a2, ac = ses.read()
    assert _same(a, s) and np.array_equal(ac, c)
    # a row band of a larger image
    band = pt.Session(gpu_scene, W, H, rows=(8, 24))
    band.render(0, 30, mrr, error=0.001)
    d, d2, dc = band.read()
    assert _same(d, s[8 * W:24 * W]) and np.array_equal(dc, c[8 * W:24 * W])
    # parameters that describe another band are refused
    p = pt.RenderParams(W, H + 1, 0, H + 1, 0, 1, mrr, 1e-4, -1.0, 42, 0)
    assert pt.lib().pt_session_render(ses._h, __import__("ctypes").byref(p), None) == 1
    assert b"band" in pt.lib().pt_last_error()
    ses.close(); band.close()


def test_pinned_host_buffers_through_the_abi(gpu_scene):
    """pt_host_alloc: page-locked accumulators passed to pt_render_host give the same frame as pageable ones."""
    import ctypes as C
    W, H, spp, mrr = 64, 48, 6, 8
    s, s2, c, _ = gpu_scene.render_host(W, H, spp, mrr)
    n = W * H
    L = pt.lib()
    ptrs = [L.pt_host_alloc(12 * n), L.pt_host_alloc(12 * n), L.pt_host_alloc(4 * n)]
    assert all(ptrs)
    try:
        arrs = [np.ctypeslib.as_array(C.cast(ptrs[0], C.POINTER(C.c_float)), (n, 3)), np.ctypeslib.as_array(C.cast(ptrs[1], C.POINTER(C.c_float)), (n, 3)),
                np.ctypeslib.as_array(C.cast(ptrs[2], C.POINTER(C.c_int32)), (n,))]
        for a in arrs:
            a[...] = 0
        gpu_scene.render_host(W, H, spp, mrr, accum=tuple(arrs), want_stats=False)
        assert _same(arrs[0], s) and _same(arrs[1], s2) and np.array_equal(arrs[2], c)
    finally:
        for p in ptrs:
            L.pt_host_free(p)
    L.pt_host_free(None)


@pytest.mark.parametrize("W,H,kw", [(960, 540, {}), (960, 540, {"error": 0.001}), (1366, 768, {"error": 0.001}), (1920, 1080, {"error": 0.001})],
                         ids=["960x540", "960x540-adaptive", "1366x768-adaptive", "1080p-adaptive"])
def test_the_schedulers_chunk_schemes_render_the_same_frame(hooks_lib, models_dir, W, H, kw):
    """Between about one and two pixel tiles per wave slot a launch's passes are cut into EQUAL chunks (pt_capi.cpp:
    enqueue_render), elsewhere into 3/4 - of - the - rest chunks; a tile's chunks run in order on whichever wave takes them, each
    starting from the accumulators the previous one published.  The frames here fall in that range on a 256-CU device (8 x 8 tiles at
    960 x 540, 16 x 8 at 1366 x 768, the 32 x 8 batch tiles of an adaptive 1080p launch): the library's own choice, the 3/4 scheme
    alone (hook < 0), and equal chunks of other sizes must all give the same bits -- the statistics instantiation's too, whose report
    shows that the choice really was equal chunks."""
    spp = 32
    frames, chunks = {}, {}
    try:
        for ips in (0.0, -1.0, 8.0, 40.0):
            hooks_lib.pt_test_set_mutation(b"reset", 0.0)
            hooks_lib.pt_test_set_mutation(b"items_per_slot", ips)
            h = pt.Scene.load_obj(models_dir, "Tor.obj", device=0, library=hooks_lib)
            s, s2, c, _ = h.render_host(W, H, spp, 8, want_stats=False, **kw)
            frames[ips] = (s.copy(), s2.copy(), c.copy())
            if ips <= 0.0:
                t, t2, tc, st = h.render_host(W, H, spp, 8, want_stats=True, **kw)
                chunks[ips] = st["n_chunks"]
                assert np.array_equal(tc, c) and _same(t, s) and _same(t2, s2)
            h.close()
    finally:
        hooks_lib.pt_test_set_mutation(b"reset", 0.0)
    for ips, f in frames.items():
        assert np.array_equal(f[2], frames[0.0][2]) and _same(f[0], frames[0.0][0]) and _same(f[1], frames[0.0][1]), ips
    assert frames[0.0][2].sum() > 0
    assert chunks[-1.0] == 2, chunks                      # 32 passes, a launch with statistics: 24 + 8
    if (W, H) == (960, 540):
        assert chunks[0.0] >= 4, chunks                   # 8 160 tiles of 8 x 8 on 5 120 slots: equal chunks of a few passes


@pytest.mark.parametrize("instances,W,H,spp,kw", [(3, 48, 32, 200, {"error": 0.05}), (6, 32, 24, 160, {"error": 0.2, "seed": 5})],
                         ids=["x3", "x6"])
def test_compacted_adaptive_passes_on_scenes_with_several_trees(tmp_path, hooks_lib, instances, W, H, spp, kw):
    """Adaptive sampling, two pixels per lane, on small scenes with several sphere-tree clusters (Tor.obj's torus 3 and 6 times
    in the room: 782 / 1 550 triangles): the tile's pixels run in batches, each pixel at its own next pass (pt_kernels.hip
    "Batches"), a lane then traces another lane's pixel through every cluster loop, root round and tree walk -- the
    oracle's bits all the same, for every tile width."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_replicated_scene as M
    d = str(tmp_path) + "/"
    M.generate(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models"), d, "r.obj", instances)
    o = O.Scene.load(d, "r.obj")
    rs, rs2, rc, rst = O.render(o, W, H, spp, 8, rng=O.RNG_COUNTER, trig=O.TRIG_PORTABLE, **kw)
    assert rst["samples_traced"] < 0.8 * W * H * spp        # the adaptive skip really bites (about 60 % of the pixels end up skipping)
    for width_mode in (3.0, 2.0, 1.0):
        hooks_lib.pt_test_set_mutation(b"tile_width", width_mode)
        hooks_lib.pt_test_set_mutation(b"big_threshold", 16384.0)      # (1 550 triangles are beyond the shipped switch: keep the sphere trees)
        try:
            h = pt.Scene.load_obj(d, "r.obj", device=0, library=hooks_lib)
            assert h.counts()[0] == o.n_tri and len(h.cull_layout()["bvh"]) == 0
            s, s2, c, _ = h.render_host(W, H, spp, 8, want_stats=False, **kw)
        finally:
            hooks_lib.pt_test_set_mutation(b"reset", 0.0)
        assert np.array_equal(c, rc) and _same(s, rs) and _same(s2, rs2), width_mode


def _tor_with_materials(tmp, n_mats):
    """Tor.obj with `n_mats` materials: copies of its five (material i is a tinted copy of material i % 5, so the light stays a
    light), the faces of every `usemtl` run dealt out over the copies."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    blocks = open(os.path.join(root, "models", "Tor.mtl")).read().split("newmtl ")[1:]
    base = {}
    for b in blocks:
        lines = b.strip().split("\n")
        base[int(lines[0])] = lines[1:]
    mtl = []
    for i in range(n_mats):
        mtl.append(f"newmtl {i}")
        for ln in base[i % 5]:
            tok = ln.split()
            if tok and tok[0] == "Kd":
                tint = 1.0 - 0.03 * (i // 5)
                ln = "Kd " + " ".join(f"{float(v) * tint:.6f}" for v in tok[1:])
            mtl.append(ln)
        mtl.append("")
    d = str(tmp) + "/"
    open(d + "m.mtl", "w").write("\n".join(mtl))
    out, cur, k = [], 0, 0
    for ln in open(os.path.join(root, "models", "Tor.obj")).read().split("\n"):
        tok = ln.split()
        if tok and tok[0] == "mtllib":
            out.append("mtllib m.mtl")
        elif tok and tok[0] == "usemtl":
            cur = int(tok[1])
        elif tok and tok[0] == "f":
            copies = [i for i in range(n_mats) if i % 5 == cur]
            out.append(f"usemtl {copies[k % len(copies)]}")
            out.append(ln)
            k += 1
        else:
            out.append(ln)
    open(d + "m.obj", "w").write("\n".join(out))
    return d


@pytest.mark.parametrize("n_mats", [16, 17, 40])
def test_material_counts_around_the_lds_copy(tmp_path, hooks_lib, n_mats):
    """The two-pixel kernels read a scene's materials from a copy in LDS if it has at most 16 of them, from memory otherwise:
    16 (the last count that fits), 17 and 40 materials, both tile widths, adaptive sampling on and off -- the oracle's bits."""
    d = _tor_with_materials(tmp_path, n_mats)
    o = O.Scene.load(d, "m.obj")
    assert len(o.materials()) == n_mats
    for kw in ({}, {"error": 0.02}):
        W, H, spp = 48, 24, 40
        rs, rs2, rc, _ = O.render(o, W, H, spp, 8, rng=O.RNG_COUNTER, trig=O.TRIG_PORTABLE, **kw)
        assert rc.sum() > 0
        for width_mode in (3.0, 2.0, 1.0):
            hooks_lib.pt_test_set_mutation(b"tile_width", width_mode)
            try:
                h = pt.Scene.load_obj(d, "m.obj", device=0, library=hooks_lib)
                s, s2, c, _ = h.render_host(W, H, spp, 8, want_stats=False, **kw)
            finally:
                hooks_lib.pt_test_set_mutation(b"reset", 0.0)
            assert np.array_equal(c, rc) and _same(s, rs) and _same(s2, rs2), (kw, width_mode)
