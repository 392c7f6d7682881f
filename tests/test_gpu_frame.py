"""pt_frame_*: one image on several devices from ONE C++ host program (the counterpart of the reference's row split inside
main(), main.cpp:115,132,141) -- row bands, every slice enqueued on all devices before any wait, one gather to the root.

On the one-GPU box the N-band code path runs as an explicit REHEARSAL (several bands per device, device-to-device copies in
place of the collective; the transfers -- source, destination, words -- are the same ones the RCCL group would issue), the
collective itself through PT_FRAME_SELF_COLLECTIVE (an RCCL send / receive to self), and the real two-device RCCL gather is
skipped below two devices.  Everything is compared bit for bit / byte for byte with the one-band result."""
import hashlib
import importlib
import json
import os
import subprocess

import numpy as np
import pytest

pt = importlib.import_module("path-tracing_amd")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "path-tracing_amd", "bin", "pt_render")


def _same(a, b):
    return (np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
            and np.array_equal(a[2], b[2]))


@pytest.fixture(scope="module")
def scene(models_dir):
    return pt.Scene.load_obj(models_dir, "Tor.obj", device=-1)      # host-only: the frame makes its own per-device copies


@pytest.fixture(scope="module")
def single(models_dir):
    """The one-device reference: pt_render_host of the same frame (itself bit-exact against the oracle, test_gpu_parity.py)."""
    s = pt.Scene.load_obj(models_dir, "Tor.obj", device=0)
    W, H, spp, mrr = 104, 50, 12, 8            # ragged: 50 rows do not divide by 3 or 8, tiles are 8 x 8
    return (W, H, spp, mrr), s.render_host(W, H, spp, mrr, error=0.001, seed=7, want_stats=True)


def test_one_band_frame_is_the_session_path(scene, single):
    (W, H, spp, mrr), ref = single
    f = pt.Frame(scene, [0], W, H)
    info = f.info()
    assert info == {"bands": 1, "rows": [[0, H]], "devices": [0], "transport": "none", "row_stride": 1}
    st = f.render(0, spp, mrr, error=0.001, seed=7, want_stats=True)
    assert _same(f.read(), ref)
    for k in ("samples_traced", "segments", "contributing", "misses"):
        assert st[k] == ref[3][k], k
    f.clear()
    s, s2, c = f.read()
    assert not s.any() and not s2.any() and not c.any()
    f.close()


@pytest.mark.parametrize("n_bands", [2, 3, 8])
def test_rehearsed_bands_equal_one_band_bit_for_bit(scene, single, n_bands):
    (W, H, spp, mrr), ref = single
    f = pt.Frame(scene, [0] * n_bands, W, H, flags=pt.FRAME_REHEARSE)
    info = f.info()
    assert info["bands"] == n_bands and info["transport"] == "device_copies" and info["devices"] == [0] * n_bands
    rows = info["rows"]
    if n_bands <= (H + 7) // 8:      # the interleaved split: band b = tile rows b, b + n, ... of the 7 this image has
        assert info["row_stride"] == n_bands and rows == [[8 * b, H] for b in range(n_bands)]
    else:                            # more bands than tile rows: contiguous bands that differ by at most a row
        assert info["row_stride"] == 1
        assert rows[0][0] == 0 and rows[-1][1] == H and all(rows[i][1] == rows[i + 1][0] for i in range(n_bands - 1))
        assert max(b - a for a, b in rows) - min(b - a for a, b in rows) <= 1
    # pass slices with a gather in between (a preview), adaptive sampling on: decisions depend on earlier passes of the band
    f.render(0, 5, mrr, error=0.001, seed=7)
    f.gather()
    part = f.read()
    assert part[2].sum() > 0
    st = f.render(5, spp - 5, mrr, error=0.001, seed=7, want_stats=True)
    assert _same(f.read(), ref)
    f.clear()
    st = f.render(0, spp, mrr, error=0.001, seed=7, want_stats=True)
    for k in ("samples_traced", "segments", "contributing", "misses"):
        assert st[k] == ref[3][k], k                              # sums over the bands
    assert _same(f.read(), ref)
    f.close()


def test_two_bands_on_one_device_need_the_rehearsal_flag(scene):
    with pytest.raises(pt.PtError) as e:
        pt.Frame(scene, [0, 0], 64, 64)
    assert e.value.status == 1 and "PT_FRAME_REHEARSE" in str(e.value)
    with pytest.raises(pt.PtError) as e:
        pt.Frame(scene, [0, pt.device_count()], 64, 64)
    assert e.value.status == 4                                    # PT_ERR_NO_DEVICE
    with pytest.raises(pt.PtError):
        f = pt.Frame(scene, [0], 64, 64)
        p = pt.RenderParams(64, 64, 0, 32, 0, 1, 8, 1e-4, -1.0, 42, 0)   # a frame owns the row split
        pt._check(f._L.pt_frame_render(f._h, p, None), f._L)


def _run(args, cwd, timeout=300):
    r = subprocess.run([EXE] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    return r


def _md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def test_cli_bands_write_the_same_bmp_as_one_gpu(tmp_path, models_dir):
    """Tor.obj 3840 x 2160 x 8 spp (BASELINE configs[3] geometry): pt_render -GPUS 1, and rehearsed 2 / 3 / 8 bands, write a
    BMP byte-identical to the plain one-device run."""
    common = ["--W", 3840, "--H", 2160, "-RPP", 8, "-MRR", 8, "-ERR", "-1", "-UPDATE", 0, "-QUIET", 1, "-SEED", 42, "-MODEL_PATH", models_dir]
    base = str(tmp_path / "base.bmp")
    _run(common + ["-OUT", base, "-DEVICE", 0], tmp_path)
    want = _md5(base)
    out = str(tmp_path / "g1.bmp")
    r = _run(common + ["-OUT", out, "-GPUS", 1, "-TIMING", 1], tmp_path)
    assert _md5(out) == want and '"transport": "none"' in r.stderr and "REHEARSAL" not in r.stderr
    for n in (2, 3, 8):
        out = str(tmp_path / f"g{n}.bmp")
        r = _run(common + ["-OUT", out, "-GPUS", n, "-REHEARSE", 1, "-TIMING", 1], tmp_path)
        assert _md5(out) == want, n
        assert "REHEARSAL" in r.stderr and '"transport": "device_copies"' in r.stderr and f'"bands": {n}' in r.stderr
    if pt.device_count() < 2:     # more bands than devices without -REHEARSE: refused, not silently rehearsed
        r = subprocess.run([EXE] + [str(a) for a in common + ["-OUT", str(tmp_path / "no.bmp"), "-GPUS", 2]], cwd=tmp_path,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "out of range" in r.stderr and not os.path.exists(tmp_path / "no.bmp")


def test_cli_previews_and_adaptive_sampling_through_bands(tmp_path, models_dir):
    """The progressive driver on a banded frame: previews gather the bands, the adaptive decisions are per pixel."""
    common = ["--W", 96, "--H", 70, "-RPP", 40, "-MRR", 8, "-UPDATE", 16, "-QUIET", 1, "-SEED", 42, "-MODEL_PATH", models_dir]
    a, b = str(tmp_path / "a.bmp"), str(tmp_path / "b.bmp")
    _run(common + ["-OUT", a], tmp_path)
    r = _run(common + ["-OUT", b, "-GPUS", 3, "-REHEARSE", 1], tmp_path)
    assert r.stderr.count("Image update") == 3 and _md5(a) == _md5(b)


def test_cli_collective_path_on_one_device(tmp_path, models_dir):
    """PT_FRAME_SELF_COLLECTIVE: the band is rendered into its own buffer and reaches the frame planes through
    ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on a communicator from ncclCommInitAll -- the collective code path
    (library loading, communicator, group, streams), with rank 0 sending to itself."""
    common = ["--W", 640, "--H", 360, "-RPP", 8, "-MRR", 8, "-ERR", "-1", "-UPDATE", 0, "-QUIET", 1, "-MODEL_PATH", models_dir]
    a, b = str(tmp_path / "a.bmp"), str(tmp_path / "b.bmp")
    _run(common + ["-OUT", a], tmp_path)
    r = _run(common + ["-OUT", b, "-SELFCOLL", 1, "-TIMING", 1], tmp_path)
    assert '"transport": "rccl"' in r.stderr and _md5(a) == _md5(b)


@pytest.mark.skipif(pt.device_count() < 2, reason="the RCCL gather between devices needs two GPUs")
def test_cli_two_devices_rccl_gather(tmp_path, models_dir):
    common = ["--W", 1920, "--H", 1080, "-RPP", 8, "-MRR", 8, "-ERR", "-1", "-UPDATE", 0, "-QUIET", 1, "-MODEL_PATH", models_dir]
    a, b = str(tmp_path / "a.bmp"), str(tmp_path / "b.bmp")
    _run(common + ["-OUT", a], tmp_path)
    r = _run(common + ["-OUT", b, "-GPUS", min(pt.device_count(), 4), "-TIMING", 1], tmp_path)
    assert '"transport": "rccl"' in r.stderr and "REHEARSAL" not in r.stderr and _md5(a) == _md5(b)


def test_cli_bench_mode_reports_whole_frames(tmp_path, models_dir):
    r = _run(["--W", 640, "--H", 360, "-RPP", 16, "-MRR", 8, "-ERR", "-1", "-MODEL_PATH", models_dir, "-GPUS", 2, "-REHEARSE", 1,
              "-BENCH_STEPS", 3, "-BENCH_WARMUP", 1], tmp_path)
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["cxx_frame"] and j["bands"] == 2 and j["transport"] == "device_copies" and j["steps"] == 3
    assert abs(j["value"] - 640 * 360 * 16 / (j["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * j["value"]


def test_cli_bands_with_skybox_and_a_big_scene(tmp_path, models_dir):
    """The per-device copies a frame makes inherit the skybox, and a big scene (box tree, accumulators in memory) comes out the
    same through bands: pt_render -SKYBOX / a 2 318-triangle replica, one device against three rehearsed bands."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import make_replicated_scene as M
    import oracle_lib as O
    rng = np.random.default_rng(3)
    sky = str(tmp_path / "sky.bmp")
    O.write_bmp(sky, rng.integers(0, 256, (24, 48, 3)).astype(np.uint8))
    d = str(tmp_path) + "/"
    M.generate(os.path.join(ROOT, "models"), d, "x9.obj", 9)
    for tag, extra in (("sky", ["-MODEL_PATH", models_dir, "-SKYBOX", sky]), ("big", ["-MODEL_PATH", d, "-MODEL_NAME", "x9.obj"])):
        common = ["--W", 200, "--H", 90, "-RPP", 12, "-MRR", 8, "-UPDATE", 0, "-QUIET", 1, "-SEED", 5] + extra
        a, b = str(tmp_path / f"{tag}_a.bmp"), str(tmp_path / f"{tag}_b.bmp")
        _run(common + ["-OUT", a], tmp_path)
        _run(common + ["-OUT", b, "-GPUS", 3, "-REHEARSE", 1], tmp_path)
        assert _md5(a) == _md5(b), tag
        assert np.frombuffer(open(a, "rb").read()[54:], np.uint8).any(), tag      # (not two black images)
