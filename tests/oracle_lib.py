"""ctypes binding of the CPU oracle (oracle/pt_oracle.c).

Test infrastructure only: imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libpt_oracle.so")

RNG_SEQUENTIAL, RNG_COUNTER = 0, 1
TRIG_LIBM, TRIG_PORTABLE = 0, 1


class Params(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("row_begin", C.c_int), ("row_end", C.c_int),
                ("pass_begin", C.c_int), ("pass_count", C.c_int), ("max_ray_reflections", C.c_int),
                ("eps", C.c_float), ("error", C.c_float), ("seed", C.c_uint32),
                ("rng_policy", C.c_int), ("trig_policy", C.c_int), ("threads", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("samples_traced", C.c_uint64), ("segments", C.c_uint64), ("contributing", C.c_uint64),
                ("stage_exit", C.c_uint64 * 5), ("misses", C.c_uint64)]


def build(force=False):
    src = os.path.join(ORACLE_DIR, "pt_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p
        L.orc_scene_load.restype = vp
        L.orc_scene_load.argtypes = [C.c_char_p, C.c_char_p]
        L.orc_scene_from_arrays.restype = vp
        L.orc_scene_from_arrays.argtypes = [fp, ip, C.c_int, fp, C.c_int]
        L.orc_scene_free.argtypes = [vp]
        L.orc_scene_num_triangles.argtypes = [vp]
        L.orc_scene_num_materials.argtypes = [vp]
        L.orc_scene_get_triangles.argtypes = [vp, fp, ip]
        L.orc_scene_get_materials.argtypes = [vp, fp]
        L.orc_render.argtypes = [vp, C.POINTER(Params), fp, fp, ip, C.POINTER(Stats)]
        L.orc_resolve.argtypes = [C.c_int, C.c_int, fp, fp, ip, C.c_float, C.POINTER(C.c_ubyte), fp]
        L.orc_resolve_float.argtypes = [C.c_int, C.c_int, fp, fp, ip, C.c_float, fp, fp]
        L.orc_gauss_blur.argtypes = [C.c_int, C.c_int, fp, C.c_float, fp]
        L.orc_median_filter.argtypes = [C.c_int, C.c_int, fp, C.c_int, fp]
        L.orc_quantize.argtypes = [C.c_int, C.c_int, fp, ip, C.POINTER(C.c_ubyte)]
        L.orc_write_bmp.restype = C.c_size_t
        L.orc_write_bmp.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_ubyte)]
        L.orc_closest_hit.argtypes = [vp, fp, fp, C.c_float, fp]
        L.orc_closest_hit_batch.argtypes = [vp, C.c_int, fp, fp, C.c_float, ip, fp, C.POINTER(C.c_ubyte), C.c_int]
        L.orc_probe_minstd.argtypes = [C.c_uint32, C.c_int, C.POINTER(C.c_uint32), fp, C.POINTER(C.c_double)]
        L.orc_probe_philox.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.orc_probe_unit_float.restype = C.c_float
        L.orc_probe_unit_float.argtypes = [C.c_uint32]
        L.orc_probe_jitter.restype = C.c_double
        L.orc_probe_jitter.argtypes = [C.c_uint32]
        L.orc_scene_set_skybox.argtypes = [vp, C.c_char_p]
        L.orc_probe_skybox.argtypes = [vp, C.c_int, fp, C.c_int, fp]
        L.orc_probe_acos_atan2.argtypes = [fp, fp, fp, C.c_int, C.c_int, fp, fp]
        L.orc_probe_sincos.argtypes = [fp, C.c_int, C.c_int, fp, fp]
        L.orc_probe_intersect.argtypes = [vp, C.c_int, fp, fp, C.c_float, C.c_float, fp]
        L.orc_probe_primary.restype = None
        L.orc_probe_primary.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int, fp]
        L.orc_probe_adaptive_skip.restype = None
        L.orc_probe_adaptive_skip.argtypes = [C.c_int, C.POINTER(C.c_int), fp, fp, C.POINTER(C.c_int), C.c_float, C.POINTER(C.c_int)]
        L.orc_probe_segments.restype = None
        L.orc_probe_segments.argtypes = [vp, C.c_int, C.c_float, C.c_int, C.c_int, fp, fp, fp, C.POINTER(C.c_int), C.POINTER(C.c_uint32), fp, C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class Scene:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle: scene could not be loaded")
        self.h = handle

    @classmethod
    def load(cls, model_dir, model_name):
        if not model_dir.endswith("/"):
            model_dir += "/"
        return cls(lib().orc_scene_load(model_dir.encode(), model_name.encode()))

    @classmethod
    def from_arrays(cls, tri14, tri_mat, mats10):
        tri14 = np.ascontiguousarray(tri14, np.float32).reshape(-1, 14)
        tri_mat = np.ascontiguousarray(tri_mat, np.int32)
        mats10 = np.ascontiguousarray(mats10, np.float32).reshape(-1, 10)
        return cls(lib().orc_scene_from_arrays(_fp(tri14), _ip(tri_mat), len(tri14), _fp(mats10), len(mats10)))

    @property
    def n_tri(self):
        return lib().orc_scene_num_triangles(self.h)

    @property
    def n_mat(self):
        return lib().orc_scene_num_materials(self.h)

    def triangles(self):
        t = np.zeros((self.n_tri, 14), np.float32)
        m = np.zeros(self.n_tri, np.int32)
        lib().orc_scene_get_triangles(self.h, _fp(t), _ip(m))
        return t, m

    def materials(self):
        a = np.zeros((self.n_mat, 10), np.float32)
        lib().orc_scene_get_materials(self.h, _fp(a))
        return a

    def closest_hit(self, o, d, eps=1e-4):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        t = C.c_float()
        i = lib().orc_closest_hit(self.h, _fp(o), _fp(d), eps, C.byref(t))
        return i, t.value

    def set_skybox(self, path):
        rc = lib().orc_scene_set_skybox(self.h, (path or "").encode())
        if rc != 0:
            raise RuntimeError(f"oracle: skybox could not be loaded ({rc})")

    def sky_lookup(self, directions, trig=None):
        """The skybox sample (r, g, b) / 256 a ray missing everything adds (scene.cpp:126-149), for unit directions [n, 3]."""
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        out = np.zeros_like(d)
        if lib().orc_probe_skybox(self.h, len(d), _fp(d), TRIG_PORTABLE if trig is None else trig, _fp(out)) != 0:
            raise RuntimeError("the scene has no skybox")
        return out

    def segments(self, origins, directions, colors, depths, words, eps=1e-4, mrr=8, trig=TRIG_LIBM):
        """One Scene::TraceRay call per ray (begin, unit direction, throughput, depth) with the three random words given: the rays
        afterwards, what was added to the accumulators, and whether anything was."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3).copy()
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3).copy()
        c = np.ascontiguousarray(colors, np.float32).reshape(-1, 3).copy()
        dep = np.ascontiguousarray(depths, np.int32).copy()
        w = np.ascontiguousarray(words, np.uint32).reshape(-1, 3)
        contrib = np.zeros_like(o)
        did = np.zeros(len(o), np.int32)
        lib().orc_probe_segments(self.h, len(o), eps, mrr, trig, _fp(o), _fp(d), _fp(c), _ip(dep), w.ctypes.data_as(C.POINTER(C.c_uint32)),
                                 _fp(contrib), _ip(did))
        return o, d, c, dep, contrib, did.astype(bool)

    def closest_hits(self, origins, directions, eps=1e-4, threads=0):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        idx = np.zeros(len(o), np.int32)
        t = np.zeros(len(o), np.float32)
        nan_seen = np.zeros(len(o), np.uint8)
        lib().orc_closest_hit_batch(self.h, len(o), _fp(o), _fp(d), eps, _ip(idx), _fp(t),
                                    nan_seen.ctypes.data_as(C.POINTER(C.c_ubyte)), threads)
        return idx, t, nan_seen.astype(bool)

    def __del__(self):
        try:
            lib().orc_scene_free(self.h)
        except Exception:
            pass


def render(scene, width, height, spp, mrr, *, eps=1e-4, error=-1.0, seed=42, rng=RNG_COUNTER, trig=TRIG_PORTABLE,
           rows=None, pass_begin=0, threads=0, accum=None):
    r0, r1 = rows if rows is not None else (0, height)
    n = (r1 - r0) * width
    if accum is None:
        s, s2, c = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
    else:
        s, s2, c = accum
    p = Params(width, height, r0, r1, pass_begin, spp, mrr, eps, error, seed, rng, trig, threads)
    st = Stats()
    rc = lib().orc_render(scene.h, C.byref(p), _fp(s), _fp(s2), _ip(c), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"orc_render failed: {rc}")
    stats = {"samples_traced": st.samples_traced, "segments": st.segments, "contributing": st.contributing,
             "stage_exit": list(st.stage_exit), "misses": st.misses}
    return s, s2, c, stats


def primary_directions(x, y, jx, jy, width, height):
    x, y = np.ascontiguousarray(x, np.int32), np.ascontiguousarray(y, np.int32)
    jx, jy = np.ascontiguousarray(jx, np.float64), np.ascontiguousarray(jy, np.float64)
    d = np.zeros((len(x), 3), np.float32)
    dp = C.POINTER(C.c_double)
    lib().orc_probe_primary(len(x), _ip(x), _ip(y), jx.ctypes.data_as(dp), jy.ctypes.data_as(dp), width, height, _fp(d))
    return d


def adaptive_skip(passes, s, s2, c, error):
    passes, c = np.ascontiguousarray(passes, np.int32), np.ascontiguousarray(c, np.int32)
    s, s2 = np.ascontiguousarray(s, np.float32), np.ascontiguousarray(s2, np.float32)
    out = np.zeros(len(c), np.int32)
    lib().orc_probe_adaptive_skip(len(c), _ip(passes), _fp(s), _fp(s2), _ip(c), error, _ip(out))
    return out.astype(bool)


def resolve(width, height, s, s2, c, gamma=np.float32(1 / np.float32(2.2))):
    bgr = np.zeros((height, width, 3), np.uint8)
    disp = np.zeros(3, np.float32)
    lib().orc_resolve(width, height, _fp(s), _fp(s2), _ip(c), C.c_float(gamma),
                      bgr.ctypes.data_as(C.POINTER(C.c_ubyte)), _fp(disp))
    return bgr, disp


def resolve_float(width, height, s, s2, c, gamma=np.float32(1 / np.float32(2.2))):
    rgb = np.zeros((height, width, 3), np.float32)
    disp = np.zeros(3, np.float32)
    lib().orc_resolve_float(width, height, _fp(s), _fp(s2), _ip(c), C.c_float(gamma), _fp(rgb), _fp(disp))
    return rgb, disp


def gauss_blur(rgb, r):
    h, w, _ = rgb.shape
    out = np.zeros_like(rgb)
    lib().orc_gauss_blur(w, h, _fp(np.ascontiguousarray(rgb, np.float32)), C.c_float(r), _fp(out))
    return out


def median_filter(rgb, ws):
    h, w, _ = rgb.shape
    out = np.zeros_like(rgb)
    lib().orc_median_filter(w, h, _fp(np.ascontiguousarray(rgb, np.float32)), ws, _fp(out))
    return out


def quantize(rgb, c):
    h, w, _ = rgb.shape
    bgr = np.zeros((h, w, 3), np.uint8)
    lib().orc_quantize(w, h, _fp(np.ascontiguousarray(rgb, np.float32)), _ip(np.ascontiguousarray(c, np.int32)),
                       bgr.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return bgr


def write_bmp(path, bgr):
    h, w, _ = bgr.shape
    bgr = np.ascontiguousarray(bgr)
    n = lib().orc_write_bmp(path.encode(), w, h, bgr.ctypes.data_as(C.POINTER(C.c_ubyte)))
    if n == 0:
        raise RuntimeError("orc_write_bmp failed")
    return n
