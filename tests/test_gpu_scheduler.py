"""The host-side choices of a launch -- which instantiation (8 x 8, 16 x 8, 32 x 8 tiles; plain passes or batches of adaptive sampling),
how the pass range is cut into chunks (3/4 of the rest down to single passes, or equal chunks between one and two tiles per wave slot)
-- must not change a bit of the frame.  Random launches: frame size, row band (one in four: an interleaved band, every n-th tile row,
against those rows of the whole frame), pass count cut into two slices on a device-resident session, adaptive threshold, scene class; the library's own choices against the plainest configuration the test hooks can pin
(8 x 8 tiles, pixels sit passes out, 3/4 chunks with the old floor of 8 passes).  GPU against GPU: the oracle's bits are what
tests/test_gpu_parity.py and tests/test_gpu_configs.py compare with on small frames, this test carries them to full-size launches."""
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


def _render(L, d, name, W, H, rows, slices, mrr, error, seed, sky=None, stride=0):
    sc = pt.Scene.load_obj(d, name, device=0, library=L)
    if sky:
        sc.set_skybox(sky)
    ses = pt.Session(sc, W, H, rows=rows, row_stride=stride)
    at = 0
    for n in slices:
        ses.render(at, n, mrr, error=error, seed=seed)
        at += n
    out = ses.read()
    ses.close()
    sc.close()
    return out


@pytest.mark.parametrize("case", range(14 + int(os.environ.get("PT_SCHED_EXTRA", "0"))))      # PT_SCHED_EXTRA=n: a soak with n more launches
def test_random_launches_against_the_plainest_configuration(tmp_path, case):
    import make_open_scene as MO
    import make_replicated_scene as MR
    rng = np.random.default_rng(4000 + case)
    L = pt.load_library(pt.TESTHOOKS_LIB_PATH)
    models = os.path.join(ROOT, "models") + "/"
    d = str(tmp_path) + "/"
    kind = ("tor", "tor", "tor", "x9", "open_sky", "tor", "x9")[case % 7]
    sky = None
    if kind == "x9":
        MR.generate(os.path.join(ROOT, "models"), d, "s.obj", 9)
        dd, name = d, "s.obj"
    elif kind == "open_sky":
        MO.generate(os.path.join(ROOT, "models"), d)
        dd, name, sky = d, "TorOpen.obj", d + "sky.bmp"
    else:
        dd, name = models, "Tor.obj"
    W = int(rng.integers(500, 2300))
    H = int(rng.integers(280, 1300))
    r0 = int(rng.integers(0, H // 3)) if case % 3 == 0 else 0
    r1 = int(rng.integers(2 * H // 3, H + 1)) if case % 3 == 0 else H
    stride = 0
    if case % 4 == 3:      # an interleaved band: every n-th tile row from tile row k on, against those rows of the plain frame
        stride = int(rng.integers(2, 10))
        r0, r1 = 8 * int(rng.integers(0, min(stride, (H + 7) // 8))), H
    spp = int(rng.integers(12, 70)) if kind != "x9" else int(rng.integers(12, 36))
    cut = int(rng.integers(1, spp))
    error = float(rng.choice([-1.0, 0.001, 0.001, 0.02, 0.3]))
    mrr = int(rng.choice([8, 8, 5, 3]))
    seed = int(rng.integers(1, 1000))
    try:
        L.pt_test_set_mutation(b"reset", 0.0)
        mine = _render(L, dd, name, W, H, (r0, r1), (cut, spp - cut), mrr, error, seed, sky, stride)
        L.pt_test_set_mutation(b"tile_width", 1.0)
        L.pt_test_set_mutation(b"items_per_slot", -1.0)
        L.pt_test_set_mutation(b"chunk_min", 8.0)
        plain = _render(L, dd, name, W, H, (r0, r1) if not stride else (0, H), (spp,), mrr, error, seed, sky)
    finally:
        L.pt_test_set_mutation(b"reset", 0.0)
    what = (kind, W, H, (r0, r1), stride, spp, cut, error, mrr, seed)
    if stride:
        where = pt.interleaved_rows(H, r0, stride)
        inside = where >= 0
        mine = tuple(a.reshape(len(where), W, -1)[inside].reshape(-1, a.shape[-1] if a.ndim > 1 else 1).squeeze() for a in mine)
        plain = tuple(a.reshape(H, W, -1)[where[inside]].reshape(-1, a.shape[-1] if a.ndim > 1 else 1).squeeze() for a in plain)
    assert plain[2].sum() > 0, what
    assert np.array_equal(mine[2], plain[2]), what
    assert np.array_equal(mine[0].view(np.uint32), plain[0].view(np.uint32)) and np.array_equal(mine[1].view(np.uint32), plain[1].view(np.uint32)), what
