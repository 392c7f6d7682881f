"""Interleaved bands (pt_render_params::row_stride): device k of n renders every n-th tile row of 8 image rows, packed in its buffers.
The counter RNG is keyed by the GLOBAL pixel and a pixel's samples do not depend on who renders its neighbours, so the rows of an
interleaved band must be, bit for bit, those rows of the whole frame -- for every instantiation (8 x 8, 16 x 8 and 32 x 8 tiles, plain
and adaptive batches, the box tree, the regenerating sky kernels, with statistics and without), for frames whose height is not a
multiple of 8, for more bands than tile rows, through sessions in two pass slices and through the host-buffer call."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
pt = importlib.import_module("path-tracing_amd")

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _check_bands(sc, W, H, spp, mrr, n, full, want_stats, **kw):
    fs, fs2, fc = (full[0].reshape(H, W, 3), full[1].reshape(H, W, 3), full[2].reshape(H, W))
    seen = np.zeros(H, bool)
    for k in range(n):
        if 8 * k > H:
            continue
        s, s2, c, _ = sc.render_host(W, H, spp, mrr, rows=(8 * k, H), row_stride=n, want_stats=want_stats, **kw)
        rows = pt.interleaved_rows(H, 8 * k, n)
        assert len(rows) * W == len(c), (k, n, len(rows), len(c))
        s, s2, c = s.reshape(len(rows), W, 3), s2.reshape(len(rows), W, 3), c.reshape(len(rows), W)
        inside = rows >= 0
        assert not seen[rows[inside]].any()
        seen[rows[inside]] = True
        assert np.array_equal(c[inside], fc[rows[inside]]), (k, n)
        assert np.array_equal(_bits(s[inside]), _bits(fs[rows[inside]])) and np.array_equal(_bits(s2[inside]), _bits(fs2[rows[inside]])), (k, n)
        assert (c[~inside] == 0).all() and (s[~inside] == 0).all()          # buffer rows beyond the image are left alone
    assert seen.all()


@pytest.mark.parametrize("width_hook", [1, 2, 3], ids=["8x8", "16x8", "32x8"])
@pytest.mark.parametrize("error", [-1.0, 0.02], ids=["plain", "adaptive"])
def test_tor_bands_are_rows_of_the_frame(width_hook, error):
    L = pt.load_library(pt.TESTHOOKS_LIB_PATH)
    models = os.path.join(ROOT, "models") + "/"
    W, H, spp, mrr = 203, 131, 40, 8
    try:
        L.pt_test_set_mutation(b"reset", 0.0)
        L.pt_test_set_mutation(b"tile_width", float(width_hook))
        sc = pt.Scene.load_obj(models, "Tor.obj", device=0, library=L)
        full = sc.render_host(W, H, spp, mrr, error=error, want_stats=False)
        assert full[2].sum() > 0
        for n in (2, 3, 8, 20):
            _check_bands(sc, W, H, spp, mrr, n, full, False, error=error)
        if width_hook == 1:
            _check_bands(sc, W, H, spp, mrr, 3, full, True, error=error)     # the statistics instantiation
    finally:
        L.pt_test_set_mutation(b"reset", 0.0)


@pytest.mark.parametrize("kind", ["x9", "x9-adaptive", "open-sky", "open-sky-adaptive"])
def test_box_tree_and_sky_bands_are_rows_of_the_frame(tmp_path, kind):
    import make_open_scene as MO
    import make_replicated_scene as MR
    d = str(tmp_path) + "/"
    error = 0.02 if "adaptive" in kind else -1.0
    if kind.startswith("x9"):
        MR.generate(os.path.join(ROOT, "models"), d, "s.obj", 9)
        sc = pt.Scene.load_obj(d, "s.obj", device=0)
    else:
        MO.generate(os.path.join(ROOT, "models"), d)
        sc = pt.Scene.load_obj(d, "TorOpen.obj", device=0)
        sc.set_skybox(d + "sky.bmp")
    W, H, spp, mrr = 150, 100, 24, 8
    full = sc.render_host(W, H, spp, mrr, error=error, want_stats=False)
    assert full[2].sum() > 0
    for n in (2, 5):
        _check_bands(sc, W, H, spp, mrr, n, full, False, error=error)


def test_full_size_interleaved_band_and_session_slices():
    """One band of four of the 1080p frame (the launch a device of a four-GPU node gets: 16 x 8 tiles, and 32 x 8 with adaptive sampling)
    on a session in two pass slices, against those rows of the whole frame."""
    models = os.path.join(ROOT, "models") + "/"
    sc = pt.Scene.load_obj(models, "Tor.obj", device=0)
    W, H, spp = 1920, 1080, 24
    for error in (-1.0, 0.001):
        full = sc.render_host(W, H, spp, 8, error=error, want_stats=False)
        fs, fc = full[0].reshape(H, W, 3), full[2].reshape(H, W)
        for k in (0, 3):
            ses = pt.Session(sc, W, H, rows=(8 * k, H), row_stride=4)
            ses.render(0, 15, 8, error=error)
            ses.render(15, spp - 15, 8, error=error)
            s, s2, c = ses.read()
            ses.close()
            rows = pt.interleaved_rows(H, 8 * k, 4)
            inside = rows >= 0
            assert np.array_equal(c.reshape(len(rows), W)[inside], fc[rows[inside]])
            assert np.array_equal(_bits(s.reshape(len(rows), W, 3)[inside]), _bits(fs[rows[inside]]))


def test_bad_strides_are_refused():
    models = os.path.join(ROOT, "models") + "/"
    sc = pt.Scene.load_obj(models, "Tor.obj", device=0)
    for rows, stride in (((4, 64), 2), ((0, 64), -1)):
        with pytest.raises(pt.PtError):
            sc.render_host(64, 64, 2, 3, rows=rows, row_stride=stride)
    p = pt.RenderParams(64, 61, 8, 61, 0, 1, 1, 1e-4, -1.0, 1, 0, 3)
    assert pt.band_rows(p) == 24 and pt.band_rows(pt.RenderParams(64, 61, 8, 61, 0, 1, 1, 1e-4, -1.0, 1, 0, 0)) == 53
