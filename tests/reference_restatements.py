"""A SECOND restatement, in numpy, of the reference -- first of the three pieces no recorded reference output touches, then (further
down) of the core of the path: loader, triangles, intersection, materials, camera ray, adaptive skip, resolve -- test
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


# ---------------------------------------------------------------------------------------------------------------------
# The core of the path, restated a second time from the reference's lines: the geometry half of Scene::LoadModel
# (scene.cpp:26-109: a whitespace token stream), the Triangle constructor and SetNormal (triangles.h:26-44), PlaneIntersect and
# ParallelogramSquare (triangles.h:10-17), Triangle::Intersect (triangles.h:48-73) and the triangle loop of Scene::TraceRay
# (scene.cpp:113-120).  GLM is not in /root/reference (a dependency the CMake build fetches); what its functions compute is GLM's
# published source (0.9.9, detail/func_geometric.inl):
#     dot(vec3 a, vec3 b)   = (a.x b.x + a.y b.y) + a.z b.z        (compute_dot<vec<3>>: tmp = a * b; tmp.x + tmp.y + tmp.z)
#     cross(x, y)           = (x.y y.z - y.y x.z,  x.z y.x - y.z x.x,  x.x y.y - y.x x.y)
#     length(v)             = sqrt(dot(v, v))
#     normalize(v)          = v * inversesqrt(dot(v, v)),   inversesqrt(x) = 1 / sqrt(x)
# Every operation is one float32 operation, in C++ evaluation order (no contraction: the build has no -mfma / -march flag,
# CMakeLists.txt:1-13).  Text -> float goes through the C library's strtof, as `istream >> float` does.
# ---------------------------------------------------------------------------------------------------------------------
_libc = ctypes.CDLL(None)
_libc.strtof.restype = ctypes.c_float
_libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]


def _strtof(tok):
    return F(_libc.strtof(tok.encode(), None))


def _dot3(a, b):
    return (((a[..., 0] * b[..., 0]).astype(F) + (a[..., 1] * b[..., 1]).astype(F)).astype(F) + (a[..., 2] * b[..., 2]).astype(F)).astype(F)


def _cross(x, y):
    return np.stack([((x[..., 1] * y[..., 2]).astype(F) - (y[..., 1] * x[..., 2]).astype(F)).astype(F),
                     ((x[..., 2] * y[..., 0]).astype(F) - (y[..., 2] * x[..., 0]).astype(F)).astype(F),
                     ((x[..., 0] * y[..., 1]).astype(F) - (y[..., 0] * x[..., 1]).astype(F)).astype(F)], -1)


def _length(v):
    return np.sqrt(_dot3(v, v)).astype(F)


def _normalize(v):
    inv = (F(1) / np.sqrt(_dot3(v, v)).astype(F)).astype(F)
    return (v * inv[..., None]).astype(F)


def load_obj_triangles(path):
    """scene.cpp:37-107 for the tokens that make triangles: `v` (three floats), `vn` (three floats), `f` (three a/b/c groups, atoi of
    each part minus one; a missing part is -1), `usemtl` (an int), `mtllib` (one more token: the file name).  Returns
    (plane [T, 4], vertices [T, 3, 3], square [T], material [T])."""
    tok = open(path).read().split()
    verts, normals, planes, tv, squares, mats = [], [], [], [], [], []
    current_material = 0
    i = 0

    def atoi(s):                                        # C atoi: optional sign and leading digits, 0 if none
        j, sign = 0, 1
        if j < len(s) and s[j] in "+-":
            sign, j = (-1 if s[j] == "-" else 1), j + 1
        k = j
        while k < len(s) and s[k].isdigit():
            k += 1
        return sign * int(s[j:k]) if k > j else 0

    while i < len(tok):
        t = tok[i]
        i += 1
        if t == "mtllib":
            i += 1
        elif t == "v":
            verts.append(np.array([_strtof(tok[i]), _strtof(tok[i + 1]), _strtof(tok[i + 2])], F))
            i += 3
        elif t == "vt":
            i += 2
        elif t == "vn":
            normals.append(np.array([_strtof(tok[i]), _strtof(tok[i + 1]), _strtof(tok[i + 2])], F))
            i += 3
        elif t == "f":
            vi, ni = [], []
            for k in range(3):
                parts = (tok[i + k].split("/") + ["", "", ""])[:3]       # Split(cr, '/'), cur.resize(3)
                vi.append(atoi(parts[0]) - 1)
                ni.append(atoi(parts[2]) - 1)
            i += 3
            v = np.stack([verts[vi[0]], verts[vi[1]], verts[vi[2]]])     # triangles.h:28-30
            ab, ac = (v[1] - v[0]).astype(F), (v[2] - v[0]).astype(F)    # :31-32
            c = _cross(ab, ac)
            n = _normalize(c)                                            # :33 SetNormal(cross(AB, AC)) -> :41-43
            square = _length(c)                                          # :34
            if ni[0] >= 0:                                               # scene.cpp:101-103: the first vertex's vn replaces the normal
                n = _normalize(normals[ni[0]])
            planes.append(np.array([n[0], n[1], n[2], -_dot3(n, v[0])], F))   # :42-43  plane_.w = -dot(normal, vertices_[0])
            tv.append(v)
            squares.append(square)
            mats.append(current_material)
        elif t == "usemtl":
            current_material = atoi(tok[i])
            i += 1
    return np.array(planes, F), np.array(tv, F), np.array(squares, F), np.array(mats, np.int32)


def trace_rays(planes, verts, squares, origins, directions, eps):
    """Scene::TraceRay's loop (scene.cpp:114-120) over Triangle::Intersect (triangles.h:48-73) for rays [n, 3] (begin, unit
    direction): (index of the triangle hit or -1, distance or inf), both [n]."""
    o, d = np.asarray(origins, F), np.asarray(directions, F)
    eps = F(eps)
    n = len(o)
    distance = np.full(n, np.inf, F)                                     # scene.cpp:114
    current = np.full(n, -1, np.int64)                                   # :115
    with np.errstate(all="ignore"):
        for i in range(len(planes)):                                     # :116
            p, v, sq = planes[i], verts[i], squares[i]
            sd = (((d[:, 0] * p[0]).astype(F) + (d[:, 1] * p[1]).astype(F)).astype(F) + (d[:, 2] * p[2]).astype(F)).astype(F)      # triangles.h:11
            num = ((((o[:, 0] * p[0]).astype(F) + (o[:, 1] * p[1]).astype(F)).astype(F) + (o[:, 2] * p[2]).astype(F)).astype(F) + p[3]).astype(F)
            nd = ((-num).astype(F) / sd).astype(F)                                                                             # :12
            ok = ~((nd >= distance) | (nd < eps))                                                                              # :51
            drop = (o + (d * nd[:, None]).astype(F)).astype(F)                                                                 # :55
            f0, f1, f2 = (drop - v[0]).astype(F), (drop - v[1]).astype(F), (drop - v[2]).astype(F)                             # :56-58
            s1 = _length(_cross(f0, f1))                                                                                       # :59
            ok &= ~(s1 > (sq + eps).astype(F))                                                                                 # :60
            s2 = _length(_cross(f0, f2))                                                                                       # :63
            ok &= ~((s1 + s2).astype(F) > (sq + eps).astype(F))                                                                # :64
            s3 = _length(_cross(f2, f1))                                                                                       # :67
            ok &= ~(np.abs((((sq - s1).astype(F) - s2).astype(F) - s3).astype(F)) > eps)                                       # :68
            distance = np.where(ok, nd, distance)                                                                              # :71
            current = np.where(ok, i, current)                                                                                 # scene.cpp:117-118
    return current, distance


# ---------------------------------------------------------------------------------------------------------------------
# ... and what happens at the hit: the MTL half of Scene::LoadModel (scene.cpp:42-71), Factory (material.h:52-105: which lobes a
# material has and their chances), Material::Process (material.h:36-50), the three lobes (material.h:67-100), Ray::Reflect and
# MakeInvalid (ray.h:45-56), with Scene::TraceRay's hit branch around them (scene.cpp:121-124).  Random() is an input here, BY CALL
# SITE: what material.h:42 (`sample`) and material.h:91 (`xi1`, `xi2`) would be handed -- which number of which stream that is belongs
# to the RNG policy, not to this code.  More of GLM's published source:
#     dot(vec4 a, vec4 b)   = (a.x b.x + a.y b.y) + (a.z b.z + a.w b.w)          (the rays' and normals' w are 0 here)
#     reflect(I, N)         = I - N * dot(N, I) * 2
#     vec * scalar, vec + vec, vec * vec: component by component
# std::cos / std::sin / std::sqrt on floats are the float overloads: the C library's cosf / sinf, an IEEE square root.
# ---------------------------------------------------------------------------------------------------------------------
for _name in ("cosf", "sinf"):
    _f = getattr(_libm, _name)
    _f.restype = ctypes.c_float
    _f.argtypes = [ctypes.c_float]


def load_mtl(path):
    """scene.cpp:45-71: one material per `newmtl`, fields Kd / Ke / Ks (three floats) and Ns (one); [n, 10] = Kd, Ke, Ks, Ns."""
    tok = open(path).read().split()
    out = []
    i = 0
    cur = "1"                                            # :47
    eof = False

    def nxt():
        nonlocal i, eof, cur
        if i < len(tok):
            cur = tok[i]
            i += 1
        else:
            eof = True                                   # (the extraction fails, the string keeps its value)

    while not eof:                                       # :48
        kd, ke, ks, ns = [F(0)] * 3, [F(0)] * 3, [F(0)] * 3, F(0)      # :49
        while not eof and cur != "newmtl":               # :50-52
            nxt()
        nxt()                                            # :53 (the material's name)
        while not eof and cur != "newmtl":               # :54
            if cur in ("Kd", "Ke", "Ks"):
                v = [_strtof(tok[i]), _strtof(tok[i + 1]), _strtof(tok[i + 2])]
                i += 3
                if cur == "Kd": kd = v
                elif cur == "Ke": ke = v
                else: ks = v
            elif cur == "Ns":
                ns = _strtof(tok[i])
                i += 1
            nxt()                                        # :67
        out.append(kd + ke + ks + [ns])                  # :69
    return np.array(out, F)


def lobes_of(m10):
    """Factory, material.h:52-105: list of (kind, chance) with kind 0 emissive, 1 glossy, 2 diffuse."""
    kd, ke, ks, ns = m10[0:3], m10[3:6], m10[6:9], F(m10[9])
    if (ke != 0).any():                                  # :61
        return [(0, F(1))]                               # :76
    out = []
    if ns != 0 and (ks != 0).any():                      # :78
        out.append((1, F(ns / F(1000))))                 # :82
    if F(F(1) - F(ns / F(1000))) > 0:                    # :85
        out.append((2, F(F(1) - F(ns / F(1000)))))       # :98
    return out


def _normalize4(v):                                      # glm::normalize(vec4) with w = 0
    dd = (((v[0] * v[0]).astype(F) + (v[1] * v[1]).astype(F)).astype(F) + ((v[2] * v[2]).astype(F) + F(0) * F(0)).astype(F)).astype(F)
    inv = (F(1) / np.sqrt(dd).astype(F)).astype(F)
    return (v * inv).astype(F)


def trace_segment(planes, verts, squares, tri_mat, mats10, o, d, color, depth, xi, eps, mrr):
    """Scene::TraceRay for ONE ray (scene.cpp:113-157 without a skybox): returns (o, d, color, depth, contribution or None, defined).
    xi = (sample, xi1, xi2): the floats the three Random() call sites would get (a site that is not reached ignores its value).  defined is False where the reference
    indexes chance_ / functions_ out of range (material.h:44-48 when the chances do not use the sample up)."""
    o, d, color = np.asarray(o, F), np.asarray(d, F), np.asarray(color, F)
    eps = F(eps)
    i, t = trace_one(planes, verts, squares, o, d, eps)                 # (scene.cpp:114-120; the triangle-by-triangle form is trace_rays)
    if i < 0:                                                            # scene.cpp:125, 155
        return o, d, color, mrr, None, True
    drop = (o + (d * t).astype(F)).astype(F)                             # :122
    n = planes[i, :3]                                                    # :123 GetNormal()
    lobes = lobes_of(mats10[tri_mat[i]])
    if not lobes:                                                        # material.h:37-38
        return o, d, color, mrr, None, True
    if len(lobes) == 1:                                                  # :39-40
        kind = lobes[0][0]
    else:
        sample = F(xi[0])                                                # :42
        k = -1
        while sample > 0:                                                # :44-47
            k += 1
            if k >= len(lobes):
                return o, d, color, depth, None, False
            sample = F(sample - lobes[k][1])
        if k < 0:
            return o, d, color, depth, None, False
        kind = lobes[k][0]
    kd, ks = mats10[tri_mat[i], 0:3], mats10[tri_mat[i], 6:9]
    dn = (((d[0] * n[0]).astype(F) + (d[1] * n[1]).astype(F)).astype(F) + ((d[2] * n[2]).astype(F) + F(0) * F(0)).astype(F)).astype(F)
    if kind == 0:                                                        # :63-75
        if dn > 0:
            return o, d, color, mrr, None, True
        return o, d, color, mrr, (color * kd).astype(F), True
    begin = (drop + (n * eps).astype(F)).astype(F)                        # :84 / :99  drop_point + N * eps
    if kind == 1:                                                        # :83-85
        ndi = (((n[0] * d[0]).astype(F) + (n[1] * d[1]).astype(F)).astype(F) + ((n[2] * d[2]).astype(F) + F(0) * F(0)).astype(F)).astype(F)
        direction = (d - ((n * ndi).astype(F) * F(2)).astype(F)).astype(F)   # reflect(I, N)
        weight = ks
    else:                                                                # :90-100
        xi1, xi2 = F(xi[1]), F(xi[2])                                    # :91
        ang = F(F(F(2) * PI) * xi2)
        sq = np.sqrt(xi1).astype(F)
        rnd = np.array([F(sq * F(_libm.cosf(float(ang)))), F(sq * F(_libm.sinf(float(ang)))), np.sqrt(F(F(1) - xi1)).astype(F)], F)
        rnd = _normalize4(rnd)
        nr = (((n[0] * rnd[0]).astype(F) + (n[1] * rnd[1]).astype(F)).astype(F) + ((n[2] * rnd[2]).astype(F) + F(0) * F(0)).astype(F)).astype(F)
        if nr < 0:
            rnd = (rnd * F(-1)).astype(F)
        nr = (((n[0] * rnd[0]).astype(F) + (n[1] * rnd[1]).astype(F)).astype(F) + ((n[2] * rnd[2]).astype(F) + F(0) * F(0)).astype(F)).astype(F)
        dt = max(F(0), nr)
        direction = rnd
        weight = (kd * dt).astype(F)
    return begin, _normalize4(direction), (color * weight).astype(F), depth + 1, None, True     # ray.h:45-50


def primary_direction(x, y, jx, jy, width, height):
    """main.cpp:126-129 + the Ray constructor, ray.h:21-25: `distribution` is a uniform_real_distribution<> (double), so the two
    coordinates are computed in double and narrowed by vec4's constructor; then glm::normalize of (dx, dy, 1, 0)."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    dx = ((x + np.asarray(jx, np.float64)) / np.float64(width) - np.float64(F(0.5))).astype(F)       # :127
    dy = (-(y + np.asarray(jy, np.float64)) / np.float64(height) + np.float64(F(0.5))).astype(F)     # :128
    dz = np.ones_like(dx)
    dd = (((dx * dx).astype(F) + (dy * dy).astype(F)).astype(F) + ((dz * dz).astype(F) + F(0) * F(0)).astype(F)).astype(F)
    inv = (F(1) / np.sqrt(dd).astype(F)).astype(F)
    return np.stack([(dx * inv).astype(F), (dy * inv).astype(F), (dz * inv).astype(F)], -1)


def adaptive_skip(rays_count, color, color2, samples, error):
    """main.cpp:118-125: True where the pixel sits pass `rays_count` out."""
    rays_count = np.asarray(rays_count)
    sc = np.asarray(samples).astype(F)                                                   # :118
    with np.errstate(all="ignore"):
        m = (np.asarray(color, F) / sc[:, None]).astype(F)
        dvar = ((np.asarray(color2, F) / sc[:, None]).astype(F) - (m * m).astype(F)).astype(F)   # :120
    low = (dvar < F(error)).all(1)                                                       # :121-122
    return (rays_count > 10) & (sc > 0) & ((rays_count % 4) != 0) & low                  # :119, :121


_libm.powf.restype = ctypes.c_float
_libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]


def resolve(color, color2, samples, gamma):
    """main.cpp:162-185 on accumulators [H, W, 3], [H, W, 3], [H, W] (row-major: the loops run y outside, x inside): the float image
    (pixels without a sample stay at their accumulated 0) and (max, min, average) dispersion.  glm::pow(vec3, vec3) is std::pow per
    component -- powf --; the running sums are float, added in loop order."""
    color, color2 = np.asarray(color, F), np.asarray(color2, F)
    samples = np.asarray(samples)
    H, W = samples.shape
    has = samples > 0
    sc = samples.astype(F)
    with np.errstate(all="ignore"):
        m = (color / sc[..., None]).astype(F)
        dvar = ((color2 / sc[..., None]).astype(F) - (m * m).astype(F)).astype(F)         # :169
    disp = ((dvar[..., 0] + dvar[..., 1]).astype(F) + dvar[..., 2]).astype(F)              # :170
    max_d, min_d = F(0), F(np.inf)                                                        # :162
    for v in disp[has]:                                                                   # :171-176 (order does not matter for max / min)
        if v > max_d: max_d = v
        if v < min_d: min_d = v
    terms = np.where(has, disp, F(1)).astype(F).ravel()                                   # :165, :177 in y-outer, x-inner order
    avg = F(0)
    for chunk in np.array_split(terms, max(1, len(terms) // 4096)):                       # a sequential float sum, block by block
        avg = np.cumsum(np.concatenate([[avg], chunk]).astype(F), dtype=F)[-1]
    avg = F(avg / F(W * H))                                                               # :184  float / int
    rgb = color.copy()
    g = float(F(gamma))
    for yy, xx in zip(*np.nonzero(has)):                                                  # :178-181
        for k in range(3):
            rgb[yy, xx, k] = F(F(_libm.powf(float(m[yy, xx, k]), g)) * F(255))
    return rgb, np.array([max_d, min_d, avg], F)


# ---------------------------------------------------------------------------------------------------------------------
# main() itself, one thread: the pass loop (main.cpp:110-140) with the reference's own two random streams.  Both are
# std::default_random_engine = minstd_rand0 seeded with Config::getSeed() (main.cpp:91, material.h:17): x <- 16807 x mod (2^31 - 1),
# first state = the seed.  What the distributions do with the engine is libstdc++'s published <random> (bits/random.tcc,
# generate_canonical; this image's GCC 11 headers): with r = max - min + 1 = 2147483646 (a long double), log2r = 30,
#     float  (24 bits, 1 draw):   float(u - 1) / float(r)                                   [float(r) = 2147483648]
#     double (53 bits, 2 draws):  (double(u1 - 1) + double(u2 - 1) * 2147483646) / double(r * r)
# clamped below 1, then a + canonical * (b - a).  The two jitter draws of a pixel sit in one constructor call
# (main.cpp:126-128); g++ evaluates its arguments right to left, so y's is drawn first (what the recorded frames of SURVEY 8(c) fix).
# ---------------------------------------------------------------------------------------------------------------------
class MinStd0:
    def __init__(self, seed):
        s = int(seed) % 2147483647
        self.x = s if s != 0 else 1

    def __call__(self):
        self.x = (self.x * 16807) % 2147483647
        return self.x


def canonical_float(eng):
    ret = F(F(eng() - 1) * F(1)) / F(2147483646.0)          # __sum / __tmp, __tmp = float(1 * r)
    return np.nextafter(F(1), F(0)) if ret >= F(1) else F(ret)


def canonical_double(eng):
    s = np.float64(eng() - 1) * np.float64(1)
    s = s + np.float64(eng() - 1) * np.float64(2147483646.0)
    ret = s / np.float64(float(2147483646 * 2147483646))
    return np.nextafter(np.float64(1), np.float64(0)) if ret >= 1 else ret


def trace_one(planes, verts, squares, o, d, eps):
    """trace_rays for one ray, all triangles at once: the loop's running `distance` makes the answer the smallest accepted distance,
    the first such triangle on a tie (triangles.h:51: `>=` rejects an equal later one)."""
    o, d, eps = np.asarray(o, F), np.asarray(d, F), F(eps)
    with np.errstate(all="ignore"):
        p = planes
        sd = (((d[0] * p[:, 0]).astype(F) + (d[1] * p[:, 1]).astype(F)).astype(F) + (d[2] * p[:, 2]).astype(F)).astype(F)
        num = ((((o[0] * p[:, 0]).astype(F) + (o[1] * p[:, 1]).astype(F)).astype(F) + (o[2] * p[:, 2]).astype(F)).astype(F) + p[:, 3]).astype(F)
        nd = ((-num).astype(F) / sd).astype(F)
        drop = (o[None, :] + (d[None, :] * nd[:, None]).astype(F)).astype(F)
        f0, f1, f2 = (drop - verts[:, 0]).astype(F), (drop - verts[:, 1]).astype(F), (drop - verts[:, 2]).astype(F)
        s1, s2, s3 = _length(_cross(f0, f1)), _length(_cross(f0, f2)), _length(_cross(f2, f1))
        lim = (squares + eps).astype(F)
        ok = (~(nd < eps) & ~np.isnan(nd) & ~(s1 > lim) & ~((s1 + s2).astype(F) > lim) &
              ~(np.abs((((squares - s1).astype(F) - s2).astype(F) - s3).astype(F)) > eps) & (nd < np.inf))
    if not ok.any():
        return -1, F(np.inf)
    best = np.where(ok, nd, F(np.inf)).min()
    return int(np.flatnonzero(ok & (nd == best))[0]), F(best)


def render_sequential(planes, verts, squares, tri_mat, mats10, width, height, spp, mrr, eps, error, seed):
    """main.cpp:91-140 on one thread: (color_map, color2_map, samples_count) as [H, W, 3], [H, W, 3], [H, W]."""
    gen = MinStd0(seed)                                                  # main.cpp:91
    mat_gen = MinStd0(seed)                                              # material.h:17
    color = np.zeros((height, width, 3), F)
    color2 = np.zeros((height, width, 3), F)
    samples = np.zeros((height, width), np.int64)
    for rays_count in range(spp):                                        # :110
        rays = {}
        for y in range(height):                                          # :116-117
            for x in range(width):
                if adaptive_skip(np.array([rays_count]), color[y, x][None], color2[y, x][None], samples[y, x][None], error)[0]:
                    continue                                             # :118-125 (the ray of the previous pass stays: it is invalid)
                jy = canonical_double(gen) * np.float64(1.0) + np.float64(-0.5)      # :128 is evaluated before :127 (g++)
                jx = canonical_double(gen) * np.float64(1.0) + np.float64(-0.5)
                rays[(x, y)] = (np.array([0, 0, -20], F), primary_direction([x], [y], [jx], [jy], width, height)[0], np.ones(3, F), 0)
        for y in range(height):                                          # :132-140
            for x in range(width):
                if (x, y) not in rays:
                    continue
                o, d, c, depth = rays[(x, y)]
                while depth < mrr and (c != 0).any():                    # ray.h:52-54
                    i, _ = trace_one(planes, verts, squares, o, d, eps)
                    lobes = lobes_of(mats10[tri_mat[i]]) if i >= 0 else []
                    xi = [F(0)] * 3                                      # Random() is drawn where the code reaches it, in order
                    if len(lobes) > 1:
                        xi[0] = canonical_float(mat_gen)
                    # which lobe?  (the same walk as trace_segment's, to know whether the diffuse lobe's two draws happen)
                    kind = None
                    if len(lobes) == 1:
                        kind = lobes[0][0]
                    elif len(lobes) > 1:
                        sample, k = xi[0], -1
                        while sample > 0 and k + 1 < len(lobes):
                            k += 1
                            sample = F(sample - lobes[k][1])
                        kind = lobes[max(k, 0)][0]
                    if kind == 2:
                        xi[1] = canonical_float(mat_gen)
                        xi[2] = canonical_float(mat_gen)
                    o, d, c, depth, contrib, defined = trace_segment(planes, verts, squares, tri_mat, mats10, o, d, c, depth, xi, eps, mrr)
                    assert defined
                    if contrib is not None:                              # material.h:73-76
                        color[y, x] = (color[y, x] + contrib).astype(F)
                        color2[y, x] = (color2[y, x] + (contrib * contrib).astype(F)).astype(F)
                        samples[y, x] += 1
    return color, color2, samples
