"""The RNG-policy half of BASELINE.json's accuracy metric ("per-channel RMSE vs CPU ref", north_star: "match the
reference CPU render on identical RNG seeds within a stated per-channel tolerance").

Under the counter policy the GPU equals the oracle bit for bit (tests/test_gpu_parity.py: tolerance 0).  The
reference's own policy -- two serial minstd_rand0 streams consumed in path order, libm trigonometry (material.h:16-20,
main.cpp:91-92,126-128) -- cannot be evaluated in parallel (pt_hip.h: PT_RNG_REFERENCE_STREAM), so against it the GPU
render is a second, independent Monte-Carlo estimate of the same image.  The committed fixture
tests/golden/tor_reference_stream_128x128.npz is the oracle in exactly that mode (the only mode pinned to the
reference's recorded BMP md5s), seeds 42..49 x 512 passes; here the GPU renders the same seeds under the counter
policy and the two are compared against the Monte-Carlo error predicted from their own per-pixel variances.

STATED TOLERANCE (tests/rng_policy_stats.py: TOLERANCE), per channel, 128 x 128 x 4096 spp, -MRR 8:
    rms of the per-pixel z-scores in (0.93, 1.08)        -- 1 means "differs by exactly the Monte-Carlo error"
    |mean z| < 0.05                                      -- no bias between the policies
    RMSE of the pixel means and of the resolved float image (pow(mean, 1/2.2) * 255) <= 1.08 x the predicted RMSE
    (measured on the CPU oracle under the counter policy: z rms 1.008 / 1.005 / 1.012, image RMSE 23.2 / 23.0 / 24.6 of
     255 = 1.012 / 1.014 / 0.972 x predicted -- only ~1 % of the samples reach the light, so 4096 spp are ~40 contributing
     samples per pixel and the image is still noisy; the tolerance says the noise is all there is)
"""
import importlib
import os

import numpy as np
import pytest

import oracle_lib as O
import rng_policy_stats as R

pt = importlib.import_module("path-tracing_amd")
HERE = os.path.dirname(os.path.abspath(__file__))


def _fixture():
    f = np.load(os.path.join(HERE, "golden", "tor_reference_stream_128x128.npz"))
    return f, (f["sum"], f["sum2"], f["count"])


@pytest.mark.gpu
def test_gpu_counter_render_matches_the_reference_stream_render(models_dir):
    f, ref = _fixture()
    W, H, mrr, passes = int(f["width"]), int(f["height"]), int(f["mrr"]), int(f["passes_per_seed"])
    g = pt.Scene.load_obj(models_dir, "Tor.obj", device=0)
    s = np.zeros((W * H, 3), np.float64)
    s2 = np.zeros((W * H, 3), np.float64)
    c = np.zeros(W * H, np.int64)
    segments = 0
    for seed in f["seeds"]:
        a = g.render_host(W, H, passes, mrr, error=-1.0, seed=int(seed))
        s += a[0]; s2 += a[1]; c += a[2]
        segments += a[3]["segments"]
    r = R.compare(ref, (s.astype(np.float32), s2.astype(np.float32), c.astype(np.int32)))
    print({k: v for k, v in r.items() if k != "channels"}, *r["channels"], sep="\n")
    R.assert_same_image(r)
    # the image binned 8 x 8 and 16 x 16: the per-channel tolerance a reader can picture (TOLERANCE["blocks"]: at most 4 / 255 and
    # 2 / 255 RMSE of the resolved float image, and at most 1.1 x / 1.15 x what the per-pixel variances predict)
    gpu = (s.astype(np.float32), s2.astype(np.float32), c.astype(np.int32))
    for b in sorted(R.TOLERANCE["blocks"]):
        rb = R.compare_blocks(ref, gpu, W, H, b)
        print(rb["block"], *rb["channels"], sep="\n")
        R.assert_same_binned_image(rb)
    assert abs(segments - int(f["segments"])) < 0.002 * int(f["segments"])
    # the 8-bit images: same statistic on what the BMP would hold
    a_bgr, _ = pt.resolve(W, H, *ref)
    b_bgr, _ = pt.resolve(W, H, s.astype(np.float32), s2.astype(np.float32), c.astype(np.int32))
    rmse_bmp = np.sqrt(((a_bgr.astype(np.float64) - b_bgr.astype(np.float64)) ** 2).reshape(-1, 3).mean(0))
    # (per-PIXEL RMSE of the 8-bit images: predicted from the variances by compare() above -- 23-25 with the 8-seed fixture of
    # rounds 2-3, 8-9 with 64 seeds; all pixels + quantisation here, hence the margin)
    pred = max(ch["rmse_image_predicted"] for ch in r["channels"])
    assert (rmse_bmp < 1.2 * pred + 1.0).all(), (rmse_bmp, pred)


@pytest.mark.gpu
def test_reference_stream_policy_is_refused_not_approximated(models_dir):
    g = pt.Scene.load_obj(models_dir, "Tor.obj", device=0)
    with pytest.raises(pt.PtError) as e:
        g.render_host(16, 16, 1, 8, rng_policy=pt.RNG_REFERENCE_STREAM)
    assert e.value.status == 7 and "minstd_rand0" in str(e.value)


def test_statistic_on_the_cpu_oracle(oracle_scene):
    """The same comparison with the oracle's counter policy on one seed (not gpu): keeps the fixture and the statistic
    honest without a device.  One seed is 8x fewer samples than the fixture, so fewer pixels qualify."""
    f, ref = _fixture()
    W, H, mrr, passes = int(f["width"]), int(f["height"]), int(f["mrr"]), int(f["passes_per_seed"])
    a = O.render(oracle_scene, W, H, passes, mrr, error=-1.0, seed=1234, rng=O.RNG_COUNTER, trig=O.TRIG_PORTABLE)
    r = R.compare(ref, a[:3], min_count=4)
    tol = dict(R.TOLERANCE, z_rms=(0.88, 1.15), z_mean_abs=0.08, rmse_ratio=1.15)
    for ch in r["channels"]:
        assert ch["n"] > 3000
    for k, ch in enumerate(r["channels"]):
        assert tol["z_rms"][0] < ch["z_rms"] < tol["z_rms"][1], ch
        assert abs(ch["z_mean"]) < tol["z_mean_abs"], ch
        assert ch["rmse_mean"] < tol["rmse_ratio"] * ch["rmse_mean_predicted"], ch
    # and a deliberately wrong image is rejected: the same render with its sums scaled by 10 %
    wrong = (a[0] * np.float32(1.1), a[1] * np.float32(1.21), a[2])
    bad = R.compare(ref, wrong, min_count=4)
    assert max(abs(ch["z_mean"]) for ch in bad["channels"]) > 0.08
    # the binned comparison on the same pair: one seed against the fixture stays inside the ratio, the scaled image does not
    # (one seed leaves pixels without a contributing sample: bins of the pixels both renders sampled, at least half of each bin)
    good_b, bad_b = R.compare_blocks(ref, a[:3], W, H, 8, min_pixel_fraction=0.5), R.compare_blocks(ref, wrong, W, H, 8, min_pixel_fraction=0.5)
    for ch in good_b["channels"]:
        assert ch["n"] > 150 and ch["rmse_image"] < 1.2 * ch["rmse_image_predicted"] and abs(ch["brightness_z"]) < 4.0, ch
    assert min(abs(ch["brightness_z"]) for ch in bad_b["channels"]) > 8.0
    with pytest.raises(AssertionError):
        R.assert_same_binned_image(bad_b)
