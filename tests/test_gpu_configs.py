"""BASELINE.json configurations the round-1 suite did not reach, and every kernel instantiation against the oracle.

configs[2]  Tor.obj 1920x1080 x 1024 spp: pass indices >= 64, long chunk schedules, 1024-term float sums.  The oracle
            renders a small frame with all 1024 passes bit for bit; the full-size frame is checked through properties.
configs[3]  Tor.obj 3840x2160 x 256 spp: a 3840-wide frame (global pixel indices above 2^22, 129 600 tiles).  Sampled
            row pairs of a reduced-spp frame against the oracle, properties of the full frame.
integrate_kernel<SKY,BIG,STATS>: all eight instantiations render oracle-identical frames.
"""
import hashlib
import importlib
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

pt = importlib.import_module("path-tracing_amd")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _digest(s, s2, c):
    return hashlib.sha256(s.tobytes() + s2.tobytes() + c.tobytes()).hexdigest()


@pytest.fixture(scope="module")
def tor(models_dir):
    assert pt.device_count() >= 1, "no HIP device: the integrator has no CPU fallback"
    return pt.Scene.load_obj(models_dir, "Tor.obj", device=0)


# ---- configs[2]: 1024 passes -------------------------------------------------------------------------------------
@pytest.mark.parametrize("error", [-1.0, 0.001])
def test_1024_passes_bit_exact(tor, oracle_scene, error):
    W, H, spp, mrr = 64, 48, 1024, 8
    s, s2, c, st = tor.render_host(W, H, spp, mrr, error=error)
    rs, rs2, rc, rst = O.render(oracle_scene, W, H, spp, mrr, error=error)
    assert st["samples_traced"] == rst["samples_traced"] and st["segments"] == rst["segments"]
    assert st["contributing"] == rst["contributing"] and st["misses"] == rst["misses"]
    if error < 0:
        assert st["samples_traced"] == W * H * spp
    else:
        assert st["samples_traced"] < 0.9 * W * H * spp          # the adaptive skip really fired (main.cpp:118-125)
    assert np.array_equal(c, rc)
    assert np.array_equal(_bits(s), _bits(rs)) and np.array_equal(_bits(s2), _bits(rs2))
    # the statistics-free instantiation (the one bench.py times) on the same long pass range
    q = tor.render_host(W, H, spp, mrr, error=error, want_stats=False)
    assert np.array_equal(_bits(q[0]), _bits(rs)) and np.array_equal(_bits(q[1]), _bits(rs2)) and np.array_equal(q[2], rc)


def test_1024_passes_pass_window_late_in_the_frame(tor, oracle_scene):
    """Passes [960, 1024) alone, added to the oracle's accumulators of passes [0, 960): pass indices far above 64 enter
    the RNG counter and the adaptive test sees sums of 960 terms."""
    W, H, mrr = 48, 32, 8
    acc = O.render(oracle_scene, W, H, 960, mrr, error=0.001)[:3]
    ref = O.render(oracle_scene, W, H, 64, mrr, error=0.001, pass_begin=960, accum=tuple(a.copy() for a in acc))[:3]
    got = tor.render_host(W, H, 64, mrr, error=0.001, pass_begin=960, accum=tuple(a.copy() for a in acc))[:3]
    assert np.array_equal(got[2], ref[2])
    assert np.array_equal(_bits(got[0]), _bits(ref[0])) and np.array_equal(_bits(got[1]), _bits(ref[1]))


def test_config2_full_size_properties(tor):
    W, H, spp, mrr = 1920, 1080, 1024, 8
    s, s2, c, st = tor.render_host(W, H, spp, mrr, error=-1.0)
    assert st["samples_traced"] == W * H * spp
    assert 7.5 * st["samples_traced"] < st["segments"] <= mrr * st["samples_traced"]
    assert st["contributing"] == int(c.sum(dtype=np.int64))
    assert 0.005 < st["contributing"] / st["samples_traced"] < 0.02
    assert (s >= 0).all() and (s2 <= s + 1e-2).all()                  # every contribution is <= 1 per channel
    assert int(c.max()) <= spp
    d = _digest(s, s2, c)
    acc = None
    for p0, n in [(0, 100), (100, 900), (1000, 24)]:                   # the same frame in three pass slices
        acc = tor.render_host(W, H, n, mrr, error=-1.0, pass_begin=p0, accum=acc, want_stats=False)[:3]
    assert _digest(*acc) == d


# ---- configs[3]: the 3840 x 2160 frame ---------------------------------------------------------------------------
def test_config3_sampled_rows_match_the_oracle(tor, oracle_scene):
    W, H, spp, mrr = 3840, 2160, 8, 8
    s, s2, c, st = tor.render_host(W, H, spp, mrr, error=-1.0)
    assert st["samples_traced"] == W * H * spp
    for r0 in (0, 822, 1079, 2158):          # top edge, torus, the seam of a two-band split, bottom edge (gpix > 2^23)
        rs, rs2, rc, _ = O.render(oracle_scene, W, H, spp, mrr, rows=(r0, r0 + 2), error=-1.0)
        sl = slice(r0 * W, (r0 + 2) * W)
        assert np.array_equal(c[sl], rc)
        assert np.array_equal(_bits(s[sl]), _bits(rs)) and np.array_equal(_bits(s2[sl]), _bits(rs2))


def test_config3_full_size_properties(tor):
    W, H, spp, mrr = 3840, 2160, 256, 8
    s, s2, c, st = tor.render_host(W, H, spp, mrr, error=-1.0)
    assert st["samples_traced"] == W * H * spp == 2123366400
    assert st["contributing"] == int(c.sum(dtype=np.int64))
    assert 0.005 < st["contributing"] / st["samples_traced"] < 0.02
    d = _digest(s, s2, c)
    # eight row bands of 270 rows (the 8-GPU split of configs[3]) assemble to the same frame
    parts = [tor.render_host(W, H, spp, mrr, error=-1.0, rows=(270 * k, 270 * (k + 1)), want_stats=False)[:3] for k in range(8)]
    assert _digest(*(np.concatenate([p[k] for p in parts]) for k in range(3))) == d


# ---- every instantiation of integrate_kernel<SKY, BIG, STATS> ----------------------------------------------------
def _replica(tmp, instances):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_replicated_scene as M
    d = str(tmp) + "/"
    name = f"TorX{instances}.obj"
    n = M.generate(os.path.join(ROOT, "models"), d, name, instances)
    return d, name, n


def _write_sky(path, w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    bgr = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), rng.integers(0, 256, (h, w))], -1).astype(np.uint8)
    O.write_bmp(path, bgr)


@pytest.mark.parametrize("sky", [False, True], ids=["nosky", "sky"])
@pytest.mark.parametrize("big", [False, True], ids=["small", "big"])
def test_all_instantiations_match_the_oracle(tmp_path, models_dir, sky, big):
    d = str(tmp_path) + "/"
    if big:
        d, name, n = _replica(tmp_path, 9)            # 9 x 256 + 14 = 2318 triangles: above the deep-queue threshold
        assert n > pt.BIG_SCENE_TRIANGLES
    else:
        name = "Tor.obj"
        for f in ("Tor.obj", "Tor.mtl"):
            open(d + f, "w").write(open(models_dir + f).read())
    if sky:
        # drop the back wall so that rays can reach the sky: keep every face but the last four of the file
        lines = open(d + name).read().split("\n")
        faces = [i for i, l in enumerate(lines) if l.startswith("f ")]
        for i in faces[-4:]:
            lines[i] = ""
        open(d + name, "w").write("\n".join(lines))
        _write_sky(d + "sky.bmp", 31, 17, seed=5)
    g = pt.Scene.load_obj(d, name, device=0)
    o = O.Scene.load(d, name)
    if sky:
        g.set_skybox(d + "sky.bmp")
        o.set_skybox(d + "sky.bmp")
    W, H, spp, mrr = 56, 40, 5, 8
    for err in (-1.0, 0.01):
        rs, rs2, rc, rst = O.render(o, W, H, spp + 14, mrr, error=err)
        if sky:
            assert rst["misses"] > 100
        for want_stats in (True, False):             # STATS and statistics-free instantiations
            s, s2, c, st = g.render_host(W, H, spp + 14, mrr, error=err, want_stats=want_stats)
            assert np.array_equal(c, rc), (sky, big, want_stats, err)
            assert np.array_equal(_bits(s), _bits(rs)) and np.array_equal(_bits(s2), _bits(rs2)), (sky, big, want_stats, err)
            if want_stats:
                assert st["segments"] == rst["segments"] and st["misses"] == rst["misses"]
                assert st["n_triangles"] == o.n_tri


# ---- both sides of the switch between the sphere-tree path and the box-tree path (pt_scene.hpp: kBigSceneTriangles) ------
@pytest.mark.parametrize("instances", [3, 4], ids=["x3_782_triangles", "x4_1038_triangles"])
def test_both_sides_of_the_small_big_switch_match_the_oracle(tmp_path, instances):
    """The torus 3 times in the room (782 triangles: sphere trees, small-scene kernels) and 4 times (1 038: one box tree,
    big-scene kernels) -- the neighbours of the switch, moved from 2 048 to 1 024 triangles in round 4 (profiles/r04_t_sweep.jsonl)
    -- through the shipped library, and each of them forced through the OTHER path by the test hook: the oracle's bits every time."""
    d, name, n = _replica(tmp_path, instances)
    assert (n > pt.BIG_SCENE_TRIANGLES) == (instances == 4)
    o = O.Scene.load(d, name)
    W, H, spp, mrr = 64, 40, 12, 8
    H_ = pt.load_library(os.path.join(os.path.dirname(pt.LIB_PATH), "libpt_testhooks.so"))
    for err in (-1.0, 0.01):
        rs, rs2, rc, rst = O.render(o, W, H, spp, mrr, error=err)
        for lib, thr in ((None, None), (H_, 0.0), (H_, 16384.0)):
            if thr is not None:
                lib.pt_test_set_mutation(b"reset", 0.0)
                lib.pt_test_set_mutation(b"big_threshold", thr)
            try:
                g = pt.Scene.load_obj(d, name, device=0, library=lib)
                has_tree = len(g.cull_layout()["bvh"]) > 0
                assert has_tree == ((n > pt.BIG_SCENE_TRIANGLES) if thr is None else (thr == 0.0))
                for want_stats in (True, False):
                    s, s2, c, st = g.render_host(W, H, spp, mrr, error=err, want_stats=want_stats)
                    assert np.array_equal(c, rc), (instances, thr, want_stats, err)
                    assert np.array_equal(_bits(s), _bits(rs)) and np.array_equal(_bits(s2), _bits(rs2)), (instances, thr, want_stats, err)
                    if want_stats:
                        assert st["segments"] == rst["segments"] and st["misses"] == rst["misses"]
            finally:
                if thr is not None:
                    lib.pt_test_set_mutation(b"reset", 0.0)


# ---- open scenes with a skybox: path regeneration (pt_kernels.hip: REGEN) -------------------------------------------------
@pytest.mark.parametrize("big", [False, True], ids=["small", "big"])
def test_open_scene_with_skybox_matches_the_oracle_for_every_path_length(tmp_path, big):
    """Tor.obj without its back wall under a sky bitmap (tools/make_open_scene.py): most paths end on their first or second
    segment, and the skybox instantiations let a lane whose path has ended start its pixel's next pass at once.  A lane is then
    in a pass of its own -- its RNG counters, its adaptive-sampling skips (which jump to the next multiple of four) and the
    order of its pixel's contributions must still be the reference's: the oracle's bits at -MRR 1 / 3 / 8, adaptive sampling
    off and on (60 passes, so that pixels sit passes out), a ragged tile width, with and without statistics, and in two pass
    slices (the second slice starts at a pass that is no multiple of four)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_open_scene as MO
    d = str(tmp_path) + "/"
    src, src_dir = "Tor.obj", None
    if big:
        src_dir, src, n = _replica(tmp_path, 5)
        assert n > pt.BIG_SCENE_TRIANGLES
    MO.generate(os.path.join(ROOT, "models"), d, name="Open.obj", source=src, source_dir=src_dir)
    g = pt.Scene.load_obj(d, "Open.obj", device=0)
    o = O.Scene.load(d, "Open.obj")
    g.set_skybox(d + "sky.bmp")
    o.set_skybox(d + "sky.bmp")
    W, H, spp = 52, 36, 60
    for mrr in (1, 3, 8):
        for err in (-1.0, 0.02):
            rs, rs2, rc, rst = O.render(o, W, H, spp, mrr, error=err)
            assert rst["misses"] > W * H * spp // 50
            if err > 0:
                assert rst["samples_traced"] < 0.97 * W * H * spp           # adaptive sampling really skips
            for want_stats in (True, False):
                s, s2, c, st = g.render_host(W, H, spp, mrr, error=err, want_stats=want_stats)
                assert np.array_equal(c, rc), (big, mrr, err, want_stats)
                assert np.array_equal(_bits(s), _bits(rs)) and np.array_equal(_bits(s2), _bits(rs2)), (big, mrr, err, want_stats)
                if want_stats:
                    assert (st["samples_traced"], st["segments"], st["misses"], st["contributing"]) == \
                           (rst["samples_traced"], rst["segments"], rst["misses"], rst["contributing"])
            # two slices through a session: passes [0, 17) and [17, 60)
            ses = pt.Session(g, W, H)
            ses.render(0, 17, mrr, error=err)
            ses.render(17, spp - 17, mrr, error=err)
            s, s2, c = ses.read()
            ses.close()
            assert np.array_equal(c, rc) and np.array_equal(_bits(s), _bits(rs)) and np.array_equal(_bits(s2), _bits(rs2)), (big, mrr, err, "slices")


def test_regeneration_setting_does_not_change_the_frame(tmp_path):
    """The library starts new paths once four slots wait from -MRR 5 up and keeps the wave's passes in step below
    (regen_min_dead_for); the test hook forces other settings: 1, 7, 33 and 64 waiting slots at -MRR 3 and 8, adaptive sampling
    on -- the same bits."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_open_scene as MO
    d = str(tmp_path) + "/"
    MO.generate(os.path.join(ROOT, "models"), d)
    o = O.Scene.load(d, "TorOpen.obj")
    o.set_skybox(d + "sky.bmp")
    H_ = pt.load_library(os.path.join(os.path.dirname(pt.LIB_PATH), "libpt_testhooks.so"))
    W, H, spp = 72, 40, 36
    for mrr in (3, 8):
        rs, rs2, rc, rst = O.render(o, W, H, spp, mrr, error=0.02)
        for setting in (1, 7, 33, 64):
            H_.pt_test_set_mutation(b"reset", 0.0)
            H_.pt_test_set_mutation(b"regen_min_dead", float(setting))
            try:
                g = pt.Scene.load_obj(d, "TorOpen.obj", device=0, library=H_)
                g.set_skybox(d + "sky.bmp")
                for want_stats in (True, False):
                    s, s2, c, st = g.render_host(W, H, spp, mrr, error=0.02, want_stats=want_stats)
                    assert np.array_equal(c, rc) and np.array_equal(_bits(s), _bits(rs)) and np.array_equal(_bits(s2), _bits(rs2)), (mrr, setting, want_stats)
                    if want_stats:
                        assert st["segments"] == rst["segments"] and st["samples_traced"] == rst["samples_traced"]
            finally:
                H_.pt_test_set_mutation(b"reset", 0.0)


def test_regeneration_keeps_the_lanes_busy(tmp_path):
    """What regeneration is for: on the open scene a wave-segment carries most of its 64 rays instead of half of them (the same
    launch with the passes kept in step, through the test hook, for comparison; what is left are the tails of the pass chunks)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_open_scene as MO
    d = str(tmp_path) + "/"
    MO.generate(os.path.join(ROOT, "models"), d)
    H_ = pt.load_library(os.path.join(os.path.dirname(pt.LIB_PATH), "libpt_testhooks.so"))
    live = {}
    for setting in (1, 64):
        H_.pt_test_set_mutation(b"reset", 0.0)
        H_.pt_test_set_mutation(b"regen_min_dead", float(setting))
        H_.pt_test_set_mutation(b"items_per_slot", -1.0)      # (the scheduler's long chunks: a frame of this size would get short ones, each with a tail)
        try:
            g = pt.Scene.load_obj(d, "TorOpen.obj", device=0, library=H_)
            g.set_skybox(d + "sky.bmp")
            st = g.render_host(960, 540, 128, 8, error=-1.0)[3]
        finally:
            H_.pt_test_set_mutation(b"reset", 0.0)
        live[setting] = st["segments"] / st["wave_segments"]
    assert live[64] < 40 and live[1] > 50 and live[1] > live[64] + 15, live
    # the library's own choice: once four slots wait from -MRR 5 up, in step below
    g = pt.Scene.load_obj(d, "TorOpen.obj", device=0)
    g.set_skybox(d + "sky.bmp")
    for mrr, busy in ((8, True), (5, True), (4, False), (2, False)):
        hook = {}
        for setting in (4, 64):
            H_.pt_test_set_mutation(b"reset", 0.0)
            H_.pt_test_set_mutation(b"regen_min_dead", float(setting))
            try:
                h = pt.Scene.load_obj(d, "TorOpen.obj", device=0, library=H_)
                h.set_skybox(d + "sky.bmp")
                hook[setting] = h.render_host(320, 200, 48, mrr, error=-1.0)[3]["wave_segments"]
            finally:
                H_.pt_test_set_mutation(b"reset", 0.0)
        assert not busy or hook[4] < hook[64], (mrr, hook)      # (fewer wave-segments for the same paths: fuller waves)
        assert g.render_host(320, 200, 48, mrr, error=-1.0)[3]["wave_segments"] == hook[4 if busy else 64], (mrr, hook)
