"""Double entry.  First for the restatements no recorded reference output touches, then
for the core of the path, whose only anchors are md5s recorded from a stand-in build (rows A1-A15): see the second half of the file.

tests/reference_restatements.py -- numpy, written from /root/reference/scene.cpp:126-149 and main.cpp:11-33, 49-80 on their own --
against oracle/pt_oracle.c, bit for bit, on random images and directions that exercise the quirks: the reversed mix weight
(1 - x + x1), the `% width` / `% height` wrap of the second texel, /256, clamp-to-edge taps, round-half-away, and element
w * w / 2 of the (2 w + 1)^2 window."""
import os

import numpy as np
import pytest

import oracle_lib as O
import reference_restatements as N


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _unit(v):
    v = np.asarray(v, np.float64)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("sw,sh,seed", [(64, 32, 1), (5, 3, 2), (1, 1, 3), (7, 200, 4)])
def test_skybox_lookup_two_restatements_agree(tmp_path, models_dir, sw, sh, seed):
    rng = np.random.default_rng(seed)
    sky = str(tmp_path / "sky.bmp")
    bgr = rng.integers(0, 256, (sh, sw, 3)).astype(np.uint8)
    O.write_bmp(sky, bgr)
    img = N.load_bmp_top_down(sky)                       # the numpy side reads the file itself ...
    assert np.array_equal(img, bgr)                      # ... and finds the rows where the writer put them (top row first in memory)
    sc = O.Scene.load(models_dir, "Tor.obj")
    sc.set_skybox(sky)
    # directions all over the sphere, plus the ones where the lookup wraps: phi just below 1 (x1 = width - 1, x2 = 0: atan2(z, -x)
    # near +pi, i.e. x > 0 and z slightly positive) and theta just below 1 (looking straight down: y2 wraps to row 0)
    d = _unit(rng.normal(size=(6000, 3)))
    seam = _unit(np.stack([np.abs(rng.normal(size=1500)) + 0.05, rng.normal(size=1500), rng.uniform(1e-7, 2e-2, 1500)], 1))
    down = _unit(np.stack([rng.normal(size=1500) * 0.02, -np.ones(1500), rng.normal(size=1500) * 0.02], 1))
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    dirs = np.concatenate([d, seam, down, axes])
    mine, defined = N.sky_lookup(img, dirs)
    theirs = sc.sky_lookup(dirs, trig=O.TRIG_LIBM)
    assert defined.sum() > 0.99 * len(dirs)              # (undefined = the reference reads outside the bitmap: phi or theta == 1)
    assert np.array_equal(_bits(mine[defined]), _bits(theirs[defined]))
    # the quirks really were exercised: second texel wrapped around, and weights on both sides of 1/2
    theta = np.arccos(np.clip(dirs[:, 1].astype(np.float64), -1, 1)) / 3.141593
    phi = np.arctan2(dirs[:, 2].astype(np.float64), -dirs[:, 0].astype(np.float64)) / 3.141593 / 2 + 0.5
    if sw > 1:
        assert ((phi * sw).astype(int) == sw - 1).sum() > 100
    if sh > 1:
        assert ((theta * sh).astype(int) == sh - 1).sum() > 100
    # and the reversed weight is what it is: a direction a quarter of a texel past a texel's left edge takes 3/4 of the RIGHT texel
    if (sw, sh) == (64, 32):
        x1, frac, y1 = 20, 0.25, 15
        ang = ((x1 + frac) / sw - 0.5) * 2 * np.pi
        th = (y1 + 0.5) / sh * 3.141593                                      # half-way between rows 15 and 16
        one = _unit([[-np.cos(ang) * np.sin(th), np.cos(th), np.sin(ang) * np.sin(th)]])
        got = sc.sky_lookup(one, trig=O.TRIG_LIBM)[0].astype(np.float64)
        row = lambda y: 0.25 * img[y, x1, ::-1].astype(np.float64) + 0.75 * img[y, x1 + 1, ::-1]      # NOT 0.75 / 0.25
        assert np.allclose(got, 0.5 * (row(y1) + row(y1 + 1)) / 256, atol=2e-3), got


@pytest.mark.parametrize("W,H,r", [(23, 17, 1), (31, 9, 2), (12, 40, 3), (40, 33, 0.7), (9, 9, 5)])
def test_gauss_blur_two_restatements_agree(W, H, r):
    rng = np.random.default_rng(int(r * 10) + W)
    img = (rng.random((H, W, 3)) * 255).astype(np.float32)
    img[rng.random((H, W)) < 0.3] = 0                    # black pixels (no samples) among lit ones, as in a real frame
    assert np.array_equal(_bits(N.gauss_blur(img, r)), _bits(O.gauss_blur(img, r)))


@pytest.mark.parametrize("W,H,ws", [(23, 17, 1), (31, 9, 2), (12, 40, 3), (8, 8, 4), (5, 3, 6)])
def test_median_filter_two_restatements_agree(W, H, ws):
    rng = np.random.default_rng(ws + W)
    img = (rng.random((H, W, 3)) * 255).astype(np.float32)
    img[rng.random((H, W)) < 0.3] = 0
    mine = N.median_filter(img, ws)
    assert np.array_equal(_bits(mine), _bits(O.median_filter(img, ws)))
    # (the element taken is window_size^2 / 2 of (2 window_size + 1)^2 sorted values: far below the middle -- not a median)
    n = (2 * ws + 1) ** 2
    assert ws * ws // 2 < n // 2
    if ws >= 2:
        true_median = np.sort(np.stack([img[np.clip(np.arange(H)[:, None] + dy, 0, H - 1), np.clip(np.arange(W)[None, :] + dx, 0, W - 1)]
                                        for dx in range(-ws, ws + 1) for dy in range(-ws, ws + 1)], 0), axis=0)[n // 2]
        assert (mine <= true_median).all() and (mine < true_median).any()


# ---- the core of the path: loader geometry, Triangle setup, Triangle::Intersect, the TraceRay loop ------------------------------
def test_triangle_tables_two_restatements_agree(models_dir):
    """Scene::LoadModel's triangles (plane from the FIRST vertex's vn where the file has one -- scene.cpp:101-103 --, square from the
    geometric normal) restated in numpy from the reference's lines against the oracle's loader: the same bits in every field."""
    planes, verts, squares, mats = N.load_obj_triangles(os.path.join(models_dir, "Tor.obj"))
    o = O.Scene.load(models_dir, "Tor.obj")
    t14, tm = o.triangles()
    assert len(planes) == o.n_tri == 270
    assert np.array_equal(_bits(planes), _bits(t14[:, :4]))
    assert np.array_equal(_bits(verts.reshape(-1, 9)), _bits(t14[:, 4:13]))
    assert np.array_equal(_bits(squares), _bits(t14[:, 13]))
    assert np.array_equal(mats, tm)
    # the quirk is live in this file: the planes carry the file's six-digit vn, not the normal through the vertices -- a few 1e-5
    # apart for most triangles, and the OPPOSITE normal where the face winds against its vn
    geo = N._normalize(N._cross((verts[:, 1] - verts[:, 0]).astype(np.float32), (verts[:, 2] - verts[:, 0]).astype(np.float32)))
    differs = (_bits(geo) != _bits(planes[:, :3])).any(1)
    along = (geo * planes[:, :3]).sum(1)
    assert differs.sum() > 200 and (along < -0.99).sum() > 0 and (np.abs(along) > 0.999).all()


@pytest.mark.parametrize("seed,eps", [(1, 1e-4), (2, 1e-4), (3, 1e-3), (4, 1e-6)])
def test_trace_ray_two_restatements_agree(models_dir, seed, eps):
    """Triangle::Intersect and the TraceRay loop, numpy from triangles.h:10-17, 48-73 and scene.cpp:114-120 against the oracle: the
    same triangle and the same distance bits for camera rays, rays from inside the room in every direction, rays starting ON
    surfaces (the paths' own: origin = a hit point pushed out by eps) and rays grazing the torus."""
    rng = np.random.default_rng(seed)
    planes, verts, squares, _ = N.load_obj_triangles(os.path.join(models_dir, "Tor.obj"))
    o = O.Scene.load(models_dir, "Tor.obj")
    n = 6000
    cam = np.tile(np.array([0, 0, -20], np.float32), (n, 1))
    cd = _unit(np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n), np.ones(n)], 1))
    inside = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    anyd = _unit(rng.normal(size=(n, 3)))
    # second-generation rays: from where the first ones land, pushed off the surface along its normal by eps (material.h: Reflect)
    i1, t1, _ = o.closest_hits(cam, cd, eps)
    hit = i1 >= 0
    p = (cam[hit] + cd[hit] * t1[hit, None]).astype(np.float32) + planes[i1[hit], :3] * np.float32(eps)
    sd = _unit(rng.normal(size=(hit.sum(), 3)))
    # grazing: aimed at torus vertices from the camera and from random points
    tv = verts[:192].reshape(-1, 3)
    tgt = tv[rng.integers(0, len(tv), n)]
    go = np.where(rng.random((n, 1)) < 0.5, cam, inside)
    gd = _unit(tgt - go)
    origins = np.concatenate([cam, inside, p.astype(np.float32), go]).astype(np.float32)
    dirs = np.concatenate([cd, anyd, sd, gd]).astype(np.float32)
    idx, t = N.trace_rays(planes, verts, squares, origins, dirs, eps)
    oi, ot, nan_seen = o.closest_hits(origins, dirs, eps)
    assert not nan_seen.any()
    assert np.array_equal(idx, oi.astype(np.int64)), (np.flatnonzero(idx != oi)[:5], idx[idx != oi][:5], oi[idx != oi][:5])
    both = idx >= 0
    assert np.array_equal(_bits(t[both]), _bits(ot[both]))
    assert both.mean() > (0.9 if eps >= 1e-4 else 0.3)      # (with eps = 1e-6 the area test's own rounding rejects every other hit)
    for k in rng.integers(0, len(origins), 300):           # the all-triangles-at-once form the per-ray restatements below use
        i1, t1 = N.trace_one(planes, verts, squares, origins[k], dirs[k], eps)
        assert i1 == idx[k] and (i1 < 0 or _bits(t1) == _bits(t[k])), k
    # the ties and near-ties the `>=` of triangles.h:51 decides: several triangles accepted per ray on the torus's seams
    assert len(np.unique(idx[both])) > 150


def test_materials_two_restatements_agree(models_dir):
    mats = N.load_mtl(os.path.join(models_dir, "Tor.mtl"))
    o = O.Scene.load(models_dir, "Tor.obj")
    assert np.array_equal(_bits(mats), _bits(o.materials()))
    kinds = [tuple(k for k, _ in N.lobes_of(m)) for m in mats]
    assert (0,) in kinds and (1, 2) in kinds                         # the light; everything else in this file is glossy + diffuse (Ns > 0)


_OTHER_MTL = """# the lobes Tor.mtl does not have: diffuse alone (Ns 0; Ks 0), glossy alone (Ns 1000), no lobe's worth of colour (Kd 0)
newmtl 0
Ke 3 2 1
Kd 0.5 0.25 1
newmtl 1
Ns 0
Kd 0.7 0.6 0.5
Ks 0.5 0.5 0.5
newmtl 2
Ns 1000
Kd 1 1 1
Ks 0.9 0.8 0.7
newmtl 3
Ns 400
Kd 0.2 0.9 0.4
Ks 0 0 0
newmtl 4
Ns 250.5
Kd 0 0 0
Ks 0.3 0.3 0.3
"""


@pytest.mark.parametrize("seed,mtl", [(11, None), (12, "other")], ids=["Tor.mtl", "other-lobes"])
def test_the_hit_branch_two_restatements_agree(tmp_path, models_dir, seed, mtl):
    """What Scene::TraceRay does with a hit -- Material::Process, the emissive / glossy / diffuse lobes, Ray::Reflect, MakeInvalid --
    restated in numpy from material.h:36-100, ray.h:45-56 and scene.cpp:121-124, against the oracle's segment (its counter-policy
    random words converted to the floats Random() would have returned; std::cos / std::sin = the C library's): the ray afterwards,
    the contribution and the depth, bit for bit, for rays that meet every material of the scene from both sides -- Tor.obj with its
    own materials, and with a material file that has the lobe combinations Tor.mtl lacks."""
    rng = np.random.default_rng(seed)
    d = models_dir
    if mtl:
        d = str(tmp_path) + "/"
        open(d + "Tor.obj", "w").write(open(os.path.join(models_dir, "Tor.obj")).read())
        open(d + "Tor.mtl", "w").write(_OTHER_MTL)
    planes, verts, squares, tri_mat = N.load_obj_triangles(os.path.join(d, "Tor.obj"))
    mats = N.load_mtl(os.path.join(d, "Tor.mtl"))
    o = O.Scene.load(d, "Tor.obj")
    assert np.array_equal(_bits(mats), _bits(o.materials()))
    if mtl:
        assert [tuple(k for k, _ in N.lobes_of(m)) for m in mats] == [(0,), (2,), (1,), (2,), (1, 2)]
    n, eps, mrr = 2500, 1e-4, 8
    origins = np.where(rng.random((n, 1)) < 0.3, np.array([[0, 0, -20]], np.float32), rng.uniform(-4, 4, (n, 3)).astype(np.float32)).astype(np.float32)
    tv = verts.reshape(-1, 3)
    aim = tv[rng.integers(0, len(tv), n)] + rng.normal(scale=0.3, size=(n, 3))
    light = verts[tri_mat == 0].mean(1)                                  # a fifth of the rays go for the emitter, from either side
    to_light = rng.random(n) < 0.2
    aim[to_light] = light[rng.integers(0, len(light), to_light.sum())] + rng.normal(scale=0.05, size=(to_light.sum(), 3))
    origins[to_light & (rng.random(n) < 0.3)] += np.array([0, 40, 0], np.float32)      # some from behind it, outside the room
    dirs = _unit(np.where((rng.random((n, 1)) < 0.6) | to_light[:, None], aim - origins, rng.normal(size=(n, 3))))
    colors = rng.uniform(0.05, 1.0, (n, 3)).astype(np.float32)
    depths = rng.integers(0, mrr, n).astype(np.int32)
    words = rng.integers(0, 2 ** 32, (n, 3), dtype=np.uint64).astype(np.uint32)
    unit = np.array([[O.lib().orc_probe_unit_float(int(w)) for w in row] for row in words], np.float32)
    oo, od, oc, odep, ocontrib, odid = o.segments(origins, dirs, colors, depths, words, eps=eps, mrr=mrr, trig=O.TRIG_LIBM)
    seen, compared = set(), 0
    for i in range(n):
        ro, rd, rc, rdep, contrib, defined = N.trace_segment(planes, verts, squares, tri_mat, mats, origins[i], dirs[i], colors[i], int(depths[i]), unit[i], eps, mrr)
        if not defined:
            continue
        compared += 1
        assert rdep == odep[i], i
        assert (contrib is not None) == bool(odid[i]), i
        if contrib is not None:
            assert np.array_equal(_bits(contrib), _bits(ocontrib[i])), i
            seen.add("emitted")
        if rdep < mrr or rdep == depths[i] + 1:        # a reflected ray: its begin, direction and throughput
            assert np.array_equal(_bits(ro), _bits(oo[i])) and np.array_equal(_bits(rd), _bits(od[i])) and np.array_equal(_bits(rc), _bits(oc[i])), i
            seen.add("reflected")
        else:
            seen.add("ended")
    assert compared > 0.99 * n and seen == {"emitted", "reflected", "ended"}


def test_primary_ray_and_adaptive_skip_two_restatements_agree():
    """main.cpp:118-129 + ray.h:21-25 in numpy against the oracle: camera directions for every pixel position class and jitter, and the
    adaptive-sampling answer over accumulators that sit on, just under and just over the threshold."""
    rng = np.random.default_rng(21)
    n = 20000
    W, H = 1920, 1080
    x, y = rng.integers(0, W, n), rng.integers(0, H, n)
    jx, jy = rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n)
    jx[:100], jy[:100] = -0.5, np.nextafter(0.5, 0)                      # the ends of the distribution's range
    assert np.array_equal(_bits(N.primary_direction(x, y, jx, jy, W, H)), _bits(O.primary_directions(x, y, jx, jy, W, H)))
    for w, h in ((1, 1), (7, 3), (3840, 2160)):
        xs, ys = rng.integers(0, w, 500), rng.integers(0, h, 500)
        assert np.array_equal(_bits(N.primary_direction(xs, ys, jx[:500], jy[:500], w, h)), _bits(O.primary_directions(xs, ys, jx[:500], jy[:500], w, h)))
    cnt = rng.integers(0, 6, n)
    contrib = rng.uniform(0, 2, (n, 5, 3)).astype(np.float32) * (np.arange(5)[None, :, None] < cnt[:, None, None])
    s = contrib.sum(1, dtype=np.float32)
    s2 = (contrib * contrib).sum(1, dtype=np.float32)
    s2[::7] = (s[::7] * s[::7] / np.maximum(cnt[::7, None], 1)).astype(np.float32)      # variance near zero, either side of it
    passes = rng.integers(0, 40, n)
    for error in (0.001, 0.0, 0.3, -1.0):
        mine = N.adaptive_skip(passes, s, s2, cnt, error)
        assert np.array_equal(mine, O.adaptive_skip(passes, s, s2, cnt, error)), error
        assert error <= 0 or mine.any()
    assert not N.adaptive_skip(passes, s, s2, cnt, -1.0).any()


def test_resolve_two_restatements_agree():
    """main.cpp:162-185 (mean, gamma, x 255; dispersion statistics with their float running sum in loop order) in numpy against the
    oracle, on accumulators with unsampled pixels, single samples and bright outliers."""
    rng = np.random.default_rng(31)
    H, W = 40, 56
    cnt = rng.integers(0, 5, (H, W)).astype(np.int32)
    contrib = rng.uniform(0, 0.9, (H, W, 4, 3)).astype(np.float32) * (np.arange(4)[None, None, :, None] < cnt[..., None, None])
    contrib[3, 5] *= 40
    s = contrib.sum(2, dtype=np.float32)
    s2 = (contrib * contrib).sum(2, dtype=np.float32)
    for gamma in (np.float32(1 / np.float32(2.2)), np.float32(1.0), np.float32(0.3)):
        rgb, disp = N.resolve(s, s2, cnt, gamma)
        orgb, odisp = O.resolve_float(W, H, s.reshape(-1, 3), s2.reshape(-1, 3), cnt.ravel(), gamma)
        assert np.array_equal(_bits(rgb), _bits(orgb)), gamma
        assert np.array_equal(_bits(disp), _bits(odisp)), (gamma, disp, odisp)
        # the bytes bitmap_image::set_pixel receives (float -> unsigned char: defined below 256)
        obgr, _ = O.resolve(W, H, s.reshape(-1, 3), s2.reshape(-1, 3), cnt.ravel(), gamma)
        ok = (rgb < 256).all(2) & (cnt > 0)
        assert ok.sum() > 0.5 * H * W and np.array_equal(obgr[ok][:, ::-1], rgb[ok].astype(np.uint8))
        assert (obgr[cnt == 0] == 0).all()


@pytest.mark.parametrize("W,H,spp,mrr,error,seed", [(24, 16, 4, 8, -1.0, 42), (12, 8, 24, 5, 0.05, 7), (9, 7, 16, 8, 0.001, 1234)],
                         ids=["plain", "adaptive", "adaptive-default-threshold"])
def test_the_whole_program_two_restatements_agree(models_dir, W, H, spp, mrr, error, seed):
    """main.cpp:91-140 on one thread with the reference's own two minstd_rand0 streams (libstdc++'s generate_canonical written out
    from its published source), every piece above composed -- camera jitter drawn y first, a path's draws taken where the code reaches
    them, contributions added in path order, the adaptive skip looking at the sums so far -- against the oracle in the mode that
    reproduces the reference's recorded frames (sequential streams, libm trig, one thread): the same accumulator bits."""
    planes, verts, squares, tri_mat = N.load_obj_triangles(os.path.join(models_dir, "Tor.obj"))
    mats = N.load_mtl(os.path.join(models_dir, "Tor.mtl"))
    color, color2, samples = N.render_sequential(planes, verts, squares, tri_mat, mats, W, H, spp, mrr, 1e-4, error, seed)
    o = O.Scene.load(models_dir, "Tor.obj")
    s, s2, c, st = O.render(o, W, H, spp, mrr, eps=1e-4, error=error, seed=seed, rng=O.RNG_SEQUENTIAL, trig=O.TRIG_LIBM, threads=1)
    assert samples.sum() > 0
    assert np.array_equal(samples.ravel(), c)
    assert np.array_equal(_bits(color.reshape(-1, 3)), _bits(s)) and np.array_equal(_bits(color2.reshape(-1, 3)), _bits(s2))
    if error >= 0:
        assert st["samples_traced"] < W * H * spp          # the skip happened
