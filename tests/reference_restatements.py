"""A SECOND restatement, in numpy, of three pieces of the reference that no recorded reference output touches -- test
infrastructure ("double entry": written from the reference's lines, float32 step by step, WITHOUT looking at oracle/pt_oracle.c,
and compared with it by tests/test_double_entry.py).

    sky_lookup      Scene::TraceRay's miss branch, /root/reference/scene.cpp:126-149, with bitmap_image::load_bitmap's row order
                    (bitmap_image.hpp:1596-1602: the file's rows are read into the image bottom row first) and get_pixel
                    (bitmap_image.hpp:169-179: data_[y * row_increment + 3 x + {0, 1, 2}] = blue, green, red)
    gauss_blur      GaussBlur, /root/reference/main.cpp:11-33
    median_filter   MedianFilter, /root/reference/main.cpp:49-80

It cannot pin the oracle to the reference -- nothing can while the reference holds no vectors -- but two independent readings of
the same lines that agree bit for bit take the single-author risk out of rows 8(f)-1 and 8(f)-3.

Conventions of the reference that matter here:  `pi` is the float 3.141593f (material.h:12);  `acos`, `atan2`, `exp` on float
arguments are the float overloads (std::acos(float) = acosf ...: main.cpp / scene.cpp say `using namespace std`), taken here from
the C library itself through ctypes so that both restatements see the same bits;  glm::mix(x, y, a) = x * (1 - a) + y * a;
glm::round = std::round (half away from zero);  vec3 / float divides every component.
"""
import ctypes
import ctypes.util
import struct

import numpy as np

F = np.float32
PI = F(3.141593)

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _name, _n in (("acosf", 1), ("atan2f", 2), ("expf", 1)):
    _f = getattr(_libm, _name)
    _f.restype = ctypes.c_float
    _f.argtypes = [ctypes.c_float] * _n


def _libm1(name, a):
    f = getattr(_libm, name)
    return np.array([f(float(v)) for v in np.asarray(a, F).ravel()], F).reshape(np.shape(a))


def _libm2(name, a, b):
    f = getattr(_libm, name)
    return np.array([f(float(u), float(v)) for u, v in zip(np.asarray(a, F).ravel(), np.asarray(b, F).ravel())], F).reshape(np.shape(a))


def load_bmp_top_down(path):
    """bitmap_image(filename): 24-bit BMP -> uint8 [height, width, 3] (blue, green, red), row 0 = the image's TOP row.
    The file stores its rows bottom-up, each padded to a multiple of 4 bytes; load_bitmap reads file row i into image row
    height - i - 1 (bitmap_image.hpp:1596-1602)."""
    raw = open(path, "rb").read()
    width, height = struct.unpack_from("<ii", raw, 18)
    assert struct.unpack_from("<H", raw, 0)[0] == 19778 and struct.unpack_from("<H", raw, 28)[0] == 24
    pad = (4 - (3 * width) % 4) % 4
    img = np.zeros((height, width, 3), np.uint8)
    at = 54
    for i in range(height):
        img[height - i - 1] = np.frombuffer(raw, np.uint8, 3 * width, at).reshape(width, 3)
        at += 3 * width + pad
    return img


def sky_lookup(sky_bgr, directions):
    """scene.cpp:126-149 for unit directions [n, 3] (float32): returns (rgb [n, 3] float32, defined [n] bool).
    `defined` is False where the reference indexes the bitmap out of range (x1 == width or y1 == height: phi or theta rounds
    to exactly 1) -- undefined behaviour there, nothing to restate."""
    d = np.asarray(directions, F)
    h, w = sky_bgr.shape[:2]
    theta = (_libm1("acosf", d[:, 1]) / PI).astype(F)                                        # :127
    at = _libm2("atan2f", d[:, 2], (-d[:, 0]).astype(F))
    phi = (((at / PI).astype(F) / F(2)).astype(F) + F(0.5)).astype(F)                        # :128  atan2(z, -x) / pi / 2 + 0.5f
    x = (phi * F(w)).astype(F)                                                               # :130  (float * unsigned -> float)
    y = (theta * F(h)).astype(F)
    defined = (x >= 0) & (y >= 0) & (x < w) & (y < h)
    xs, ys = np.where(defined, x, 0), np.where(defined, y, 0)
    x1 = xs.astype(np.uint32)                                                                # :131-132  static_cast<unsigned>: truncation
    y1 = ys.astype(np.uint32)
    x2 = (x1 + 1) % np.uint32(w)                                                             # :133-134
    y2 = (y1 + 1) % np.uint32(h)

    def colour(px, py):                                                                      # :136-144  vec3(red, green, blue)
        t = sky_bgr[py, px]
        return np.stack([t[:, 2], t[:, 1], t[:, 0]], 1).astype(F)

    c1, c2, c3, c4 = colour(x1, y1), colour(x2, y1), colour(x1, y2), colour(x2, y2)
    ax = ((F(1) - xs).astype(F) + x1.astype(F)).astype(F)[:, None]                           # :146  1 - x + x1, left to right
    ay = ((F(1) - ys).astype(F) + y1.astype(F)).astype(F)[:, None]

    def mix(p, q, a):
        return ((p * (F(1) - a).astype(F)).astype(F) + (q * a).astype(F)).astype(F)

    c12, c34 = mix(c1, c2, ax), mix(c3, c4, ax)                                              # :146-147
    return (mix(c12, c34, ay) / F(256)).astype(F), defined                                   # :149


def gauss_blur(rgb, r):
    """main.cpp:11-33 on a float32 image [H, W, 3] (the reference's color_map[x][y] is this image transposed)."""
    img = np.asarray(rgb, F)
    H, W = img.shape[:2]
    r = F(r)
    rs = int(np.ceil(np.float64(r) * 2.57))                                                  # :12  float * double -> double, ceil, int
    two_rr = F(F(F(2) * r) * r)                                                              # :24  2 * r * r, left to right
    norm = F(F(F(PI * F(2)) * r) * r)                                                        # :24  pi * 2 * r * r, left to right
    val = np.zeros((H, W, 3), F)
    wsum = F(0)
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    for dy in range(-rs, rs + 1):                                                            # :19  iy = i - rs .. i + rs
        for dx in range(-rs, rs + 1):                                                        # :20
            xc = np.minimum(W - 1, np.maximum(0, jj + dx))                                   # :21-22  clamp to the edge
            yc = np.minimum(H - 1, np.maximum(0, ii + dy))
            dsq = dx * dx + dy * dy                                                          # :23  int
            e = F(_libm.expf(float(F(F(-dsq) / two_rr))))                                    # :24  exp(-dsq / (2 r r)): int / float -> float
            wght = F(e / norm)
            val = (val + (img[yc, xc] * wght).astype(F)).astype(F)                           # :25
            wsum = F(wsum + wght)                                                            # :26
    q = (val / wsum).astype(F)
    return (np.sign(q) * np.floor(np.abs(q) + F(0.5))).astype(F)                             # :29  std::round: half away from zero


def median_filter(rgb, window_size):
    """main.cpp:49-80: element window_size * window_size / 2 (integer division) of the sorted (2 w + 1)^2 window, per channel."""
    img = np.asarray(rgb, F)
    H, W = img.shape[:2]
    ws = int(window_size)
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    taps = []
    for wx in range(-ws, ws + 1):                                                            # :58
        for wy in range(-ws, ws + 1):                                                        # :60
            i = np.maximum(np.minimum(wx + jj, W - 1), 0)                                    # :62  column, clamped
            j = np.maximum(np.minimum(wy + ii, H - 1), 0)                                    # :63  row, clamped
            taps.append(img[j, i])
    window = np.sort(np.stack(taps, 0), axis=0)                                              # :69, :72, :75
    return window[ws * ws // 2].astype(F)                                                    # :70  window[window_size * window_size / 2]
