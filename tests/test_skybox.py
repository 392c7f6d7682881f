"""Skybox miss shader (scene.cpp:16-23,126-154): BMP ingestion on the CPU, bit-exact GPU-vs-oracle frames on an open
scene where most rays leave, and the portable acos/atan2 the lookup uses."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

import oracle_lib as O

pt = importlib.import_module("path-tracing_amd")


def _write_sky(path, w, h, seed=0):
    """A 24-bit BMP with smooth gradients + noise, written with the oracle's writer (reference byte layout)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    bgr = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), rng.integers(0, 256, (h, w))], -1).astype(np.uint8)
    O.write_bmp(path, bgr)
    return bgr


def _open_scene(tmp, models_dir):
    """The torus and the light of Tor.obj without the walls: almost every path ends in the sky."""
    src = open(models_dir + "Tor.obj").read().split("\n")
    out = []
    for line in src:
        out.append(line)
        if line.startswith("usemtl 1"):      # white walls start here (models/Tor.obj:497)
            break
    out = [l for l in out if not l.startswith("usemtl 1")]
    # drop the two green-wall faces (the first two faces after "usemtl 2")
    res, skip = [], 0
    for l in out:
        if l.startswith("usemtl 2"):
            skip = 2
            continue
        if skip and l.startswith("f "):
            skip -= 1
            continue
        res.append(l)
    open(tmp + "open.obj", "w").write("\n".join(res) + "\n")
    open(tmp + "Tor.mtl", "w").write(open(models_dir + "Tor.mtl").read())


def test_portable_acos_atan2_are_correctly_rounded():
    rng = np.random.default_rng(2)
    n = 400_000
    v = np.concatenate([rng.uniform(-1, 1, n - 6), [1, -1, 0, 0.99999994, -0.99999994, 1e-30]]).astype(np.float32)
    y = rng.normal(size=n).astype(np.float32)
    x = rng.normal(size=n).astype(np.float32)
    y[:4] = [0, 0, 1, -1]
    x[:4] = [1, -1, 0, 0]
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    ac, at = np.zeros(n, np.float32), np.zeros(n, np.float32)
    O.lib().orc_probe_acos_atan2(fp(v), fp(y), fp(x), n, O.TRIG_PORTABLE, fp(ac), fp(at))
    for got, ref in ((ac, np.arccos(v.astype(np.float64))), (at, np.arctan2(y.astype(np.float64), x.astype(np.float64)))):
        ulp = np.spacing(np.abs(ref).astype(np.float32) + np.float32(1e-37)).astype(np.float64)
        assert np.max(np.abs(got.astype(np.float64) - ref) / ulp) <= 0.5000001
    assert at[0] == 0 and at[1] == np.float32(np.pi) and at[2] == np.float32(np.pi / 2) and at[3] == np.float32(-np.pi / 2)


def test_skybox_loader_checks(tmp_path, models_dir):
    s = pt.Scene.load_obj(models_dir, "Tor.obj", device=-1)
    good = str(tmp_path / "sky.bmp")
    _write_sky(good, 37, 21)                       # row padding exercised: 3*37 = 111 -> 1 pad byte
    s.set_skybox(good)
    s.set_skybox(None)
    with pytest.raises(pt.PtError) as e:
        s.set_skybox(str(tmp_path / "missing.bmp"))
    assert e.value.status == 2
    data = bytearray(open(good, "rb").read())
    for at, val, what in ((0, b"XX", "type"), (28, bytes([32, 0]), "bit depth"), (14, bytes([12, 0, 0, 0]), "BIH size")):
        bad = bytearray(data)
        bad[at:at + len(val)] = val
        p = str(tmp_path / "bad.bmp")
        open(p, "wb").write(bad)
        with pytest.raises(pt.PtError) as e:
            s.set_skybox(p)
        assert e.value.status == 3 and what in str(e.value)
    open(str(tmp_path / "short.bmp"), "wb").write(data[:-5])
    with pytest.raises(pt.PtError) as e:
        s.set_skybox(str(tmp_path / "short.bmp"))
    assert e.value.status == 3 and "logical" in str(e.value)
    o = O.Scene.load(models_dir, "Tor.obj")
    o.set_skybox(good)
    with pytest.raises(RuntimeError):
        o.set_skybox(str(tmp_path / "short.bmp"))


def test_oracle_skybox_accumulates_without_throughput(tmp_path, models_dir):
    d = str(tmp_path) + "/"
    _open_scene(d, models_dir)
    o = O.Scene.load(d, "open.obj")
    assert o.n_tri == 258
    sky = _write_sky(d + "sky.bmp", 64, 32, seed=3)
    s0, _, c0, st0 = O.render(o, 32, 24, 4, 8)
    o.set_skybox(d + "sky.bmp")
    s1, s2, c1, st1 = O.render(o, 32, 24, 4, 8)
    assert st0["segments"] == st1["segments"] and st0["misses"] == st1["misses"] > 1000
    assert st1["contributing"] == st0["contributing"] + st1["misses"]      # every miss now counts (scene.cpp:151-153)
    assert s1.max() <= 1.0 * 4 and (s1 >= s0).all() and (c1 >= c0).all()      # texel/256 < 1 per sample


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,spp,sw,sh", [(48, 32, 6, 64, 32), (40, 24, 12, 5, 3), (33, 17, 5, 1, 1)])
def test_gpu_skybox_bit_exact(tmp_path, models_dir, W, H, spp, sw, sh):
    d = str(tmp_path) + "/"
    _open_scene(d, models_dir)
    _write_sky(d + "sky.bmp", sw, sh, seed=sw)
    g = pt.Scene.load_obj(d, "open.obj", device=0)
    o = O.Scene.load(d, "open.obj")
    g.set_skybox(d + "sky.bmp")
    o.set_skybox(d + "sky.bmp")
    for err in (-1.0, 0.02):
        s, s2, c, st = g.render_host(W, H, spp, 8, error=err)
        rs, rs2, rc, rst = O.render(o, W, H, spp, 8, error=err)
        assert st["segments"] == rst["segments"] and st["misses"] == rst["misses"] and st["contributing"] == rst["contributing"]
        assert np.array_equal(c, rc)
        assert np.array_equal(s.view(np.uint32), rs.view(np.uint32)) and np.array_equal(s2.view(np.uint32), rs2.view(np.uint32))
    # removing the skybox restores the no-skybox kernel and its results
    g.set_skybox(None)
    o.set_skybox(None)
    s, s2, c, st = g.render_host(W, H, spp, 8)
    rs, rs2, rc, rst = O.render(o, W, H, spp, 8)
    assert np.array_equal(s.view(np.uint32), rs.view(np.uint32)) and np.array_equal(c, rc)


@pytest.mark.gpu
def test_gpu_device_copies_keep_the_skybox_they_were_made_with(tmp_path, models_dir):
    """pt_scene_clone_to_device shares the host side of a scene; a copy inherits the skybox set before it was made and keeps
    it -- texels AND size -- when the original gets another one afterwards."""
    d = str(tmp_path) + "/"
    _open_scene(d, models_dir)
    _write_sky(d + "a.bmp", 64, 32, seed=1)
    _write_sky(d + "b.bmp", 5, 3, seed=2)
    g = pt.Scene.load_obj(d, "open.obj", device=0)
    g.set_skybox(d + "a.bmp")
    copy = g.clone_to_device(0)               # inherits a.bmp
    g.set_skybox(d + "b.bmp")                 # the original moves on to a smaller image
    late = g.clone_to_device(0)               # inherits b.bmp
    o = O.Scene.load(d, "open.obj")
    for scene, sky in ((copy, "a.bmp"), (g, "b.bmp"), (late, "b.bmp")):
        o.set_skybox(d + sky)
        s, s2, c, st = scene.render_host(40, 24, 6, 8)
        rs, rs2, rc, rst = O.render(o, 40, 24, 6, 8)
        assert st["misses"] == rst["misses"] > 0 and np.array_equal(c, rc), sky
        assert np.array_equal(s.view(np.uint32), rs.view(np.uint32)) and np.array_equal(s2.view(np.uint32), rs2.view(np.uint32)), sky
