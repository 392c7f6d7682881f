"""Closest-hit parity on explicit rays: the GPU's cull hierarchy + exact test against the oracle's brute-force loop
(scene.cpp:114-120), bit for bit, on rays that ordinary path sampling rarely produces: grazing, edge-on, vertex-on,
axis-aligned, starting on surfaces, pointing away."""
import importlib

import numpy as np
import pytest

import oracle_lib as O

pt = importlib.import_module("path-tracing_amd")
pytestmark = pytest.mark.gpu


def _normalise(d):
    """Ray's constructor (ray.h:23): v * (1 / sqrt((x*x + y*y) + z*z)) in float32."""
    d = np.ascontiguousarray(d, np.float32)
    inv = np.float32(1.0) / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32)
    return (d * inv[:, None]).astype(np.float32)


def _adversarial_rays(tri, rng, n):
    T = len(tri)
    v = tri[:, 4:13].reshape(T, 3, 3).astype(np.float64)
    nrm = tri[:, 0:3].astype(np.float64)
    O_, D_ = [], []
    # (1) random interior origins, random directions
    o = rng.uniform([-9.9, -9.9, -20.9], [9.9, 9.9, 9.9], (n, 3))
    d = rng.normal(size=(n, 3))
    O_.append(o); D_.append(d)
    # (2) from a surface point (+eps*N like the lobes do) towards a point on another triangle's EDGE or VERTEX
    a, b = rng.integers(0, T, n), rng.integers(0, T, n)
    w = rng.dirichlet([1, 1, 1], n)
    src = (v[a] * w[:, :, None]).sum(1) + nrm[a] * 1e-4
    e = rng.random((n, 1))
    kind = rng.integers(0, 3, n)
    tgt = np.where((kind == 0)[:, None], v[b, 0], np.where((kind == 1)[:, None], v[b, 0] * e + v[b, 1] * (1 - e),
                                                           v[b, 1] * e + v[b, 2] * (1 - e)))
    O_.append(src); D_.append(tgt - src)
    # (3) grazing: direction in the plane of a triangle, tilted by a tiny angle, from just above that plane
    a = rng.integers(0, T, n)
    tang = v[a, 1] - v[a, 0]
    tang /= np.linalg.norm(tang, axis=1, keepdims=True) + 1e-30
    tilt = rng.choice([0.0, 1e-7, -1e-7, 1e-5, -1e-5, 1e-3, -1e-3], n)[:, None]
    src = (v[a] * w[:, :, None]).sum(1) + nrm[a] * rng.choice([0.0, 1e-4, -1e-4, 1e-2], n)[:, None] - tang * rng.uniform(0, 5, (n, 1))
    O_.append(src); D_.append(tang + tilt * nrm[a])
    # (4) axis-aligned directions and origins on lattice points (exact zeros in products)
    o = rng.integers(-9, 10, (n, 3)).astype(np.float64)
    d = np.zeros((n, 3)); d[np.arange(n), rng.integers(0, 3, n)] = rng.choice([-1.0, 1.0], n)
    O_.append(o); D_.append(d)
    # (5) towards the centroid of a random triangle, from far and from very near
    a = rng.integers(0, T, n)
    cen = v[a].mean(1)
    src = np.where(rng.random((n, 1)) < 0.5, rng.uniform(-9, 9, (n, 3)), cen + rng.normal(size=(n, 3)) * 1e-3)
    O_.append(src); D_.append(cen - src + 1e-12)
    o = np.concatenate(O_).astype(np.float32)
    d = _normalise(np.concatenate(D_).astype(np.float32))
    ok = np.isfinite(d).all(1)
    return o[ok], d[ok]


def _check(g, o_scene, o, d, eps=1e-4):
    gi, gt = g.trace_rays(o, d, eps)
    ri, rt, nan_seen = o_scene.closest_hits(o, d, eps)
    # Rays that lie EXACTLY in some triangle's stored plane (0/0 in PlaneIntersect) are outside the contract: the
    # reference then accepts that triangle wherever it is, which no geometric cull can follow (deviations).
    # They cannot come out of the integrator (no direction component is ever exactly zero there); only the lattice
    # family below produces them.
    assert nan_seen.mean() < 0.01
    bad = np.flatnonzero(((gi != ri) | (gt.view(np.uint32) != rt.view(np.uint32))) & ~nan_seen)
    assert bad.size == 0, (f"{bad.size} of {len(o)} rays differ; first: ray {bad[0]} o={o[bad[0]]} d={d[bad[0]]} "
                           f"gpu=({gi[bad[0]]},{gt[bad[0]]}) oracle=({ri[bad[0]]},{rt[bad[0]]})")
    return gi


def test_tor_scene_rays(models_dir, oracle_scene):
    g = pt.Scene.load_obj(models_dir, "Tor.obj", device=0)
    tri, _ = oracle_scene.triangles()
    o, d = _adversarial_rays(tri, np.random.default_rng(5), 120_000)
    hits = _check(g, oracle_scene, o, d)
    assert (hits >= 0).mean() > 0.5 and (hits < 0).sum() > 100      # both hits and misses are exercised
    _check(g, oracle_scene, o[:50_000], d[:50_000], eps=1e-3)          # tables are rebuilt for another eps


def test_empty_batch_and_device_rule(models_dir):
    g = pt.Scene.load_obj(models_dir, "Tor.obj", device=0)
    i, t = g.trace_rays(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert len(i) == 0
    h = pt.Scene.load_obj(models_dir, "Tor.obj", device=-1)
    with pytest.raises(pt.PtError) as e:
        h.trace_rays(np.zeros((1, 3), np.float32), np.array([[0, 0, 1]], np.float32))
    assert e.value.status == 4


def test_replicated_scene_rays(tmp_path):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_replicated_scene as M
    d_ = str(tmp_path) + "/"
    M.generate(os.path.join(root, "models"), d_, "x27.obj", 27)
    g = pt.Scene.load_obj(d_, "x27.obj", device=0)
    o_scene = O.Scene.load(d_, "x27.obj")
    tri, _ = o_scene.triangles()
    o, d = _adversarial_rays(tri, np.random.default_rng(6), 12_000)
    _check(g, o_scene, o, d)


@pytest.mark.parametrize("n_tri,n_rays", [(512, 1), (512, 4), (512, 16), (512, 17), (64, 1), (64, 4), (64, 16), (40, 3), (9, 2)])
def test_root_round_and_its_fallbacks(tmp_path, n_tri, n_rays):
    """The root round of the tree walk (few rays near a small tree: the level below the top is tested in one
    lane-spread round) and the cases it must hand back to the general path: a 'rod' of small triangles stacked along z
    and rays fired down its axis, so that EVERY node of every level is kept for every ray -- with 4 rays and 64 nodes
    below the top that is 256 survivors, more than either queue holds; 17 rays are one more than the round takes;
    64 / 40 / 9 triangles make the level below the top the triangles themselves (pairs instead of nodes)."""
    (tmp_path / "r.mtl").write_text("newmtl 0\nKd 0.5 0.5 0.5\n")
    lines = ["mtllib r.mtl", "usemtl 0"]
    for k in range(n_tri):
        z = -10.0 + 0.02 * k
        lines += [f"v -0.2 -0.15 {z:.6f}", f"v 0.2 -0.15 {z:.6f}", f"v 0.0 0.25 {z:.6f}", f"f {3 * k + 1} {3 * k + 2} {3 * k + 3}"]
    (tmp_path / "r.obj").write_text("\n".join(lines) + "\n")
    d = str(tmp_path) + "/"
    g = pt.Scene.load_obj(d, "r.obj", device=0)
    o = O.Scene.load(d, "r.obj")
    t = g.cull_tables()
    assert list(t["kind"]) == [0] and t["n_tri"][0] == n_tri
    org = np.zeros((n_rays, 3), np.float32)
    org[:, 0] = np.linspace(-0.05, 0.05, n_rays) if n_rays > 1 else 0.0
    org[:, 2] = -15.0
    dirs = _normalise(np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (n_rays, 1)) +
                      np.linspace(0, 1e-3, n_rays, dtype=np.float32)[:, None] * np.array([[1.0, 0.5, 0.0]], np.float32))
    gi, gt = g.trace_rays(org, dirs)
    ri, rt, nan_seen = o.closest_hits(org, dirs)
    assert not nan_seen.any()
    assert np.array_equal(gi, ri) and np.array_equal(gt.view(np.uint32), rt.view(np.uint32))
    assert (ri == 0).all()          # the first triangle of the rod is the closest hit of every ray
    # and from the far end, where the LAST triangle is the answer (every candidate must have been examined)
    org2 = org.copy(); org2[:, 2] = 5.0
    gi, gt = g.trace_rays(org2, -dirs)
    ri, rt, _ = o.closest_hits(org2, -dirs)
    assert np.array_equal(gi, ri) and np.array_equal(gt.view(np.uint32), rt.view(np.uint32)) and (ri == n_tri - 1).all()


def test_rays_outside_the_envelope_get_the_reference_answer(models_dir, oracle_scene, tmp_path):
    """pt_trace_rays_host accepts any ray.  The culling margins are derived for unit directions and origins within
    r_org = max(20, largest |coordinate|) + 1; rays outside that envelope -- far origins (|o| up to 1e5, where the
    sphere test's |m|^2 - (m.d)^2 would cancel catastrophically), unnormalised directions, non-finite origins -- are
    answered by the reference's all-triangles loop on the device, so the answer is still the reference's, bit for bit.
    Both the small-scene kernel (Tor.obj) and the box-tree kernel (a replicated scene)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_replicated_scene as M
    d_ = str(tmp_path) + "/"
    M.generate(os.path.join(root, "models"), d_, "x9.obj", 9)
    rng = np.random.default_rng(12)
    for g, o_scene in ((pt.Scene.load_obj(models_dir, "Tor.obj", device=0), oracle_scene),
                       (pt.Scene.load_obj(d_, "x9.obj", device=0), O.Scene.load(d_, "x9.obj"))):
        n = 6000
        tgt = rng.uniform(-8, 8, (n, 3))
        far = rng.normal(size=(n, 3))
        far = far / np.linalg.norm(far, axis=1, keepdims=True) * rng.choice([25.0, 1e3, 1e4, 1e5], n)[:, None]
        o = far.astype(np.float32)
        d = _normalise((tgt - far).astype(np.float32))
        # a third of them with directions that are NOT unit length (scaled by 0.5 ... 3): t scales by the inverse
        scale = np.where(np.arange(n) % 3 == 0, rng.uniform(0.5, 3.0, n), 1.0).astype(np.float32)
        d = (d * scale[:, None]).astype(np.float32)
        # and inside origins mixed in, so that one wave holds both kinds
        inside = np.arange(n) % 4 == 1
        o[inside] = rng.uniform(-9, 9, (inside.sum(), 3)).astype(np.float32)
        hits = _check(g, o_scene, o, d)
        assert (hits >= 0).mean() > 0.3
    g = pt.Scene.load_obj(models_dir, "Tor.obj", device=0)
    o = np.array([[np.inf, 0, 0], [np.nan, 0, 0], [0, 0, -20]], np.float32)
    d = np.array([[0, 0, 1], [0, 0, 1], [0, 0, 1]], np.float32)
    gi, gt = g.trace_rays(o, d)
    ri, rt, _ = oracle_scene.closest_hits(o, d)
    assert np.array_equal(gi[2:], ri[2:]) and gi[2] >= 0            # the finite ray is answered
    # non-finite origins make every distance NaN: the reference then "accepts" by failing every comparison (the known NaN
    # deviation of pt_hip.h); here such rays simply miss -- and must not crash or disturb the other lanes of their wave
    assert gi[0] == -1 and gi[1] == -1


def test_negative_eps_means_every_ray_misses(models_dir, oracle_scene):
    """eps < 0: the last test of Triangle::Intersect, abs(..) > eps (triangles.h:68), rejects every triangle -- in the
    reference and here alike, for explicit rays and for frames."""
    g = pt.Scene.load_obj(models_dir, "Tor.obj", device=0)
    tri, _ = oracle_scene.triangles()
    o, d = _adversarial_rays(tri, np.random.default_rng(9), 4000)
    gi, gt = g.trace_rays(o, d, eps=-1e-4)
    ri, rt, nan_seen = oracle_scene.closest_hits(o, d, -1e-4)
    ok = ~nan_seen
    assert (gi[ok] == -1).all() and (ri[ok] == -1).all() and np.isinf(gt[ok]).all()
    s, s2, c, st = g.render_host(32, 24, 4, 8, eps=-1e-4)
    rs, rs2, rc, rst = O.render(oracle_scene, 32, 24, 4, 8, eps=-1e-4)
    assert st["segments"] == rst["segments"] == st["misses"] == 32 * 24 * 4 and not c.any() and not rc.any()
