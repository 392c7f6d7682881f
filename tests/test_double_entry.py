"""Double entry for the restatements no recorded reference output touches:
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
