"""Post filters (main.cpp:11-33, 49-80, 187-201): the three-step resolve on the CPU, the GPU kernels against the
oracle's loops bit for bit, and the CLI's -GAUSS / -MEDIAN."""
import importlib
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pt = importlib.import_module("path-tracing_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _frame(oracle_scene, W=44, H=31, spp=24):
    return O.render(oracle_scene, W, H, spp, 8, rng=O.RNG_COUNTER, trig=O.TRIG_PORTABLE)[:3]


def test_three_step_resolve_equals_fused(oracle_scene):
    W, H = 44, 31
    s, s2, c = _frame(oracle_scene)
    rgb, disp = pt.resolve_float(W, H, s, s2, c)
    orgb, odisp = O.resolve_float(W, H, s, s2, c)
    assert np.array_equal(rgb.view(np.uint32), orgb.view(np.uint32)) and np.array_equal(disp.view(np.uint32), odisp.view(np.uint32))
    bgr, disp2 = pt.resolve(W, H, s, s2, c)
    assert np.array_equal(pt.quantize(rgb, c), bgr) and np.array_equal(disp, disp2)
    assert np.array_equal(O.quantize(orgb, c), bgr)


def test_filter_argument_checks():
    rgb = np.zeros((4, 4, 3), np.float32)
    assert np.array_equal(pt.post_filter(rgb, 0, 0, device=0), rgb)      # nothing to do: no device needed
    with pytest.raises(pt.PtError) as e:
        pt.post_filter(rgb, 0, 12)
    assert e.value.status == 1
    if pt.device_count() == 0:
        with pytest.raises(pt.PtError) as e:
            pt.post_filter(rgb, 2, 0)
        assert e.value.status == 4      # no CPU fallback


@pytest.mark.gpu
# -GAUSS 1..9 run the LDS-tiled kernel, 10 the global-memory one; -MEDIAN 1..3 the register kernels, 4 and 11 the generic one
@pytest.mark.parametrize("gauss,median", [(1, 0), (3, 0), (9, 0), (10, 0), (0, 1), (0, 2), (0, 3), (0, 4), (0, 11)])
def test_gpu_filters_bit_exact(oracle_scene, gauss, median):
    W, H = 44, 31
    s, s2, c = _frame(oracle_scene)
    rgb, _ = O.resolve_float(W, H, s, s2, c)
    rng = np.random.default_rng(gauss * 16 + median)
    noisy = (rgb + rng.uniform(0, 40, rgb.shape)).astype(np.float32)      # dense content: the filters have work to do
    for img in (rgb, noisy):
        want = O.gauss_blur(img, gauss) if gauss else O.median_filter(img, median)
        got = pt.post_filter(img, gauss, median)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_cli_gauss_matches_oracle(tmp_path, models_dir, oracle_scene):
    exe = os.path.join(ROOT, "path-tracing_amd", "bin", "pt_render")
    W, H, spp = 40, 32, 20
    out = str(tmp_path / "g.bmp")
    r = subprocess.run([exe, "--W", str(W), "--H", str(H), "-RPP", str(spp), "-ERR", "-1", "-UPDATE", "0", "-QUIET", "1",
                        "-GAUSS", "2", "-MODEL_PATH", models_dir, "-OUT", out], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    s, s2, c, _ = O.render(oracle_scene, W, H, spp, 8)
    rgb, _ = O.resolve_float(W, H, s, s2, c)
    ref = str(tmp_path / "ref.bmp")
    O.write_bmp(ref, O.quantize(O.gauss_blur(rgb, 2), c))
    assert open(ref, "rb").read() == open(out, "rb").read()
