"""MI355X radiance integrator: thin ctypes binding of libpt_hip.so (include/pt_hip.h).

The product is the C ABI; this module only loads it for the Python-side harnesses (tests, bench.py, smoke).
There is no CPU fallback: if the shared library is missing, or no HIP device is usable, calls raise.

The directory name contains a hyphen, so import it with
    importlib.import_module("path-tracing_amd")
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_HERE, "csrc")
# PT_HIP_LIB selects another build of the same ABI (kernel-variant experiments); the default is the in-tree library.
LIB_PATH = os.environ.get("PT_HIP_LIB") or os.path.join(_HERE, "lib", "libpt_hip.so")

PT_OK = 0
STATUS_NAMES = {0: "PT_OK", 1: "PT_ERR_INVALID_ARGUMENT", 2: "PT_ERR_IO", 3: "PT_ERR_PARSE", 4: "PT_ERR_NO_DEVICE",
                5: "PT_ERR_HIP", 6: "PT_ERR_OUT_OF_MEMORY", 7: "PT_ERR_UNSUPPORTED"}
PT_ABI_VERSION = 5
RNG_COUNTER, RNG_REFERENCE_STREAM = 0, 1
# test-only builds of the same ABI (csrc/Makefile): never loaded by the product path
VERIFY_LIB_PATH = os.path.join(_HERE, "lib", "libpt_verify.so")
TESTHOOKS_LIB_PATH = os.path.join(_HERE, "lib", "libpt_testhooks.so")

# every symbol include/pt_hip.h declares
ABI_SYMBOLS = ["pt_scene_load_obj", "pt_scene_create", "pt_scene_counts", "pt_scene_get_triangles",
               "pt_scene_get_materials", "pt_scene_destroy", "pt_render_device", "pt_render_host", "pt_trace_rays_host",
               "pt_session_create", "pt_session_render", "pt_session_wait", "pt_session_read", "pt_session_clear", "pt_session_destroy",
               "pt_scene_cull_tables", "pt_scene_cull_layout", "pt_scene_set_skybox_bmp", "pt_resolve", "pt_resolve_float",
               "pt_post_filter_host", "pt_quantize",
               "pt_write_bmp", "pt_host_alloc", "pt_host_free", "pt_abi_version", "pt_device_count", "pt_last_error",
               "pt_scene_clone_to_device", "pt_scene_timings", "pt_table_limits_check", "pt_rccl_available",
               "pt_frame_create", "pt_frame_info", "pt_frame_render", "pt_frame_gather", "pt_frame_wait", "pt_frame_read",
               "pt_frame_clear", "pt_frame_destroy", "pt_frame_band_kernel_ms", "pt_scene_skybox_size", "pt_table_limits_check_tree",
               "pt_band_rows", "pt_session_create_strided", "pt_frame_row_stride"]
FRAME_REHEARSE, FRAME_SELF_COLLECTIVE = 1, 2
BIG_SCENE_TRIANGLES = 1024     # csrc/pt_scene.hpp: kBigSceneTriangles -- scenes above it take the box-tree path (tests/test_abi_host.py compares)
TRANSPORT_NAMES = {0: "none", 1: "rccl", 2: "device_copies"}


class PtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("row_begin", C.c_int32), ("row_end", C.c_int32),
                ("pass_begin", C.c_int32), ("pass_count", C.c_int32), ("max_ray_reflections", C.c_int32),
                ("eps", C.c_float), ("error", C.c_float), ("seed", C.c_uint32), ("rng_policy", C.c_int32), ("row_stride", C.c_int32)]


class RenderStats(C.Structure):
    _fields_ = [("samples_traced", C.c_uint64), ("segments", C.c_uint64), ("contributing", C.c_uint64),
                ("exact_tests", C.c_uint64), ("misses", C.c_uint64), ("wave_segments", C.c_uint64),
                ("wave_node_rounds", C.c_uint64), ("wave_exact_iterations", C.c_uint64), ("kernel_ms", C.c_float),
                ("n_triangles", C.c_int32), ("n_chunks", C.c_int32), ("partial_commit_rounds", C.c_int32),
                ("verify_checked", C.c_uint64), ("verify_mismatches", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force=False):
    """Compile libpt_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC_DIR, "clean", "-s"])
    subprocess.check_call(["make", "-C", CSRC_DIR, "-s", "-j4"])   # the product library + the test / diagnostic builds
    return LIB_PATH


_lib = None


def _share_torch_hip_runtime():
    """PyTorch's ROCm wheels bundle their own libamdhip64.so.7.  A process that ends up with two HIP runtimes (the
    system one pulled in by libpt_hip.so, then torch's) loses the GPU in whichever initialises second -- torch reported
    "No HIP GPUs are available" when it was imported after this library.  If torch is installed, load its copy first
    (located without importing torch): the dynamic linker then resolves libpt_hip.so's libamdhip64.so.7 to it, as
    it already does when torch is imported first, and the order of imports stops mattering."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("PT_HIP_SYSTEM_RUNTIME"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library(path):
    """Load one build of the ABI (the product library by default; tests also load the verification / test-hook builds)."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: run `make -C {CSRC_DIR}` (there is no CPU fallback)")
    _share_torch_hip_runtime()
    L = C.CDLL(path)
    fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p
    L.pt_scene_load_obj.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp)]
    L.pt_scene_create.argtypes = [fp, ip, C.c_int32, fp, C.c_int32, C.c_int, C.POINTER(vp)]
    L.pt_scene_counts.argtypes = [vp, ip, ip]
    L.pt_scene_get_triangles.argtypes = [vp, fp, ip]
    L.pt_scene_get_materials.argtypes = [vp, fp]
    L.pt_scene_destroy.argtypes = [vp]
    L.pt_scene_destroy.restype = None
    L.pt_render_device.argtypes = [vp, C.POINTER(RenderParams), vp, vp, vp, vp, C.POINTER(RenderStats)]
    L.pt_render_host.argtypes = [vp, C.POINTER(RenderParams), fp, fp, ip, C.POINTER(RenderStats)]
    L.pt_session_create.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.pt_session_create_strided.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.pt_band_rows.restype = C.c_int32
    L.pt_band_rows.argtypes = [C.POINTER(RenderParams)]
    L.pt_session_render.argtypes = [vp, C.POINTER(RenderParams), C.POINTER(RenderStats)]
    L.pt_session_read.argtypes = [vp, fp, fp, ip]
    L.pt_session_wait.argtypes = [vp]
    L.pt_session_clear.argtypes = [vp]
    L.pt_session_destroy.argtypes = [vp]
    L.pt_session_destroy.restype = None
    L.pt_trace_rays_host.argtypes = [vp, C.c_int32, fp, fp, C.c_float, ip, fp]
    L.pt_scene_cull_tables.argtypes = [vp, C.c_float, ip, fp, fp, fp, fp]
    L.pt_scene_cull_layout.argtypes = [vp, C.c_float, ip, ip, vp]
    L.pt_scene_set_skybox_bmp.argtypes = [vp, C.c_char_p]
    L.pt_resolve.argtypes = [C.c_int32, C.c_int32, fp, fp, ip, C.c_float, C.POINTER(C.c_uint8), fp]
    L.pt_resolve_float.argtypes = [C.c_int32, C.c_int32, fp, fp, ip, C.c_float, fp, fp]
    L.pt_post_filter_host.argtypes = [C.c_int, C.c_int32, C.c_int32, fp, C.c_int32, C.c_int32]
    L.pt_quantize.argtypes = [C.c_int32, C.c_int32, fp, ip, C.POINTER(C.c_uint8)]
    L.pt_write_bmp.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]
    L.pt_last_error.restype = C.c_char_p
    L.pt_host_alloc.restype = C.c_void_p
    L.pt_host_alloc.argtypes = [C.c_size_t]
    L.pt_host_free.argtypes = [C.c_void_p]
    L.pt_host_free.restype = None
    L.pt_scene_clone_to_device.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.pt_scene_timings.argtypes = [vp, C.POINTER(C.c_double)]
    L.pt_table_limits_check.argtypes = [C.c_uint64, C.c_uint64, C.c_int32]
    L.pt_rccl_available.argtypes = [ip]
    L.pt_frame_create.argtypes = [vp, ip, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.POINTER(vp)]
    L.pt_frame_info.argtypes = [vp, ip, ip, ip, ip]
    L.pt_frame_render.argtypes = [vp, C.POINTER(RenderParams), C.POINTER(RenderStats)]
    L.pt_frame_gather.argtypes = [vp]
    L.pt_frame_wait.argtypes = [vp]
    L.pt_frame_read.argtypes = [vp, fp, fp, ip]
    L.pt_frame_clear.argtypes = [vp]
    L.pt_frame_destroy.argtypes = [vp]
    L.pt_frame_destroy.restype = None
    L.pt_frame_band_kernel_ms.argtypes = [vp, fp]
    L.pt_frame_row_stride.argtypes = [vp, ip]
    L.pt_scene_skybox_size.argtypes = [vp, ip, ip]
    L.pt_table_limits_check_tree.argtypes = [C.c_uint64, C.c_uint64, C.c_int32, C.c_int32]
    if hasattr(L, "pt_test_set_mutation"):
        L.pt_test_set_mutation.argtypes = [C.c_char_p, C.c_double]
    return L


def lib():
    global _lib
    if _lib is None:
        _lib = load_library(LIB_PATH)
    return _lib


def _check(status, L=None):
    if status != PT_OK:
        raise PtError(status, (L or lib()).pt_last_error().decode(errors="replace"))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def device_count():
    return lib().pt_device_count()


class Scene:
    """Owns a pt_scene handle (Scene + LoadModel of the reference, scene.h:17-19)."""

    def __init__(self, handle, library=None):
        self._h = handle
        self._L = library or lib()

    @classmethod
    def load_obj(cls, model_dir, model_name, device=0, library=None):
        L = library or lib()
        h = C.c_void_p()
        _check(L.pt_scene_load_obj(model_dir.encode(), model_name.encode(), device, C.byref(h)), L)
        return cls(h, L)

    @classmethod
    def create(cls, triangles, triangle_material, materials, device=0, library=None):
        L = library or lib()
        t = np.ascontiguousarray(triangles, np.float32).reshape(-1, 14)
        m = np.ascontiguousarray(triangle_material, np.int32)
        k = np.ascontiguousarray(materials, np.float32).reshape(-1, 10)
        h = C.c_void_p()
        _check(L.pt_scene_create(_fp(t), _ip(m), len(t), _fp(k), len(k), device, C.byref(h)), L)
        return cls(h, L)

    def counts(self):
        nt, nm = C.c_int32(), C.c_int32()
        _check(self._L.pt_scene_counts(self._h, C.byref(nt), C.byref(nm)), self._L)
        return nt.value, nm.value

    def triangles(self):
        nt, _ = self.counts()
        t = np.zeros((nt, 14), np.float32)
        m = np.zeros(nt, np.int32)
        _check(self._L.pt_scene_get_triangles(self._h, _fp(t), _ip(m)), self._L)
        return t, m

    def materials(self):
        _, nm = self.counts()
        k = np.zeros((nm, 10), np.float32)
        _check(self._L.pt_scene_get_materials(self._h, _fp(k)), self._L)
        return k

    def render_host(self, width, height, spp, mrr, *, eps=1e-4, error=-1.0, seed=42, rows=None, pass_begin=0,
                    accum=None, want_stats=True, rng_policy=RNG_COUNTER, row_stride=0):
        r0, r1 = rows if rows is not None else (0, height)
        p = RenderParams(width, height, r0, r1, pass_begin, spp, mrr, eps, error, seed, rng_policy, row_stride)
        n = band_rows(p, self._L) * width
        if accum is None:
            s, s2, c = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
        else:
            s, s2, c = accum
        st = RenderStats()
        _check(self._L.pt_render_host(self._h, C.byref(p), _fp(s), _fp(s2), _ip(c), C.byref(st) if want_stats else None), self._L)
        return s, s2, c, st.as_dict()

    def clone_to_device(self, device):
        """The same scene on another device; host side (parsed model, hierarchies) shared."""
        h = C.c_void_p()
        _check(self._L.pt_scene_clone_to_device(self._h, device, C.byref(h)), self._L)
        return Scene(h, self._L)

    def timings(self):
        t = (C.c_double * 2)()
        _check(self._L.pt_scene_timings(self._h, t), self._L)
        return {"load_s": t[0], "hierarchy_build_s": t[1]}

    def set_skybox(self, path):
        """-SKYBOX: a 24-bit BMP sampled by rays that hit nothing (scene.cpp:126-154); None or "" removes it."""
        _check(self._L.pt_scene_set_skybox_bmp(self._h, (path or "").encode()), self._L)

    def skybox_size(self):
        w, h = C.c_int32(), C.c_int32()
        _check(self._L.pt_scene_skybox_size(self._h, C.byref(w), C.byref(h)), self._L)
        return w.value, h.value

    def cull_tables(self, eps=1e-4):
        """The culling hierarchy for `eps` (diagnostics): dict of clusters, spheres, bary records, constants."""
        counts = np.zeros(4, np.int32)
        _check(self._L.pt_scene_cull_tables(self._h, eps, _ip(counts), None, None, None, None), self._L)
        cl = np.zeros((counts[0], 16), np.float32)
        sp = np.zeros((counts[1], 4), np.float32)
        ba = np.zeros((counts[2], 12), np.float32)
        k = np.zeros(5, np.float32)
        _check(self._L.pt_scene_cull_tables(self._h, eps, _ip(counts), _fp(cl), _fp(sp), _fp(ba), _fp(k)), self._L)
        meta = cl[:, 4:16].view(np.uint32)
        return {"cluster_sphere": cl[:, :4], "first_tri": meta[:, 0].astype(int), "n_tri": meta[:, 1].astype(int),
                "kind": meta[:, 2].astype(int), "data_off": meta[:, 3].astype(int), "n_levels": meta[:, 4].astype(int),
                "level_off": np.concatenate([np.zeros((len(cl), 1), int), meta[:, 5:12].astype(int)], 1),
                "spheres": sp, "bary": ba,
                "constants": dict(zip(["k1", "k2", "a_max", "m0", "t_guard"], k.tolist())), "n_large": int(counts[3])}

    def cull_layout(self, eps=1e-4):
        """Slot order of the hierarchy: dict with slot_triangle (original index per slot, -1 = padding), node counts."""
        counts = np.zeros(4, np.int32)
        _check(self._L.pt_scene_cull_layout(self._h, eps, _ip(counts), None, None), self._L)
        st = np.zeros(counts[0], np.int32)
        nodes = np.zeros((counts[1], 64), np.uint8)
        _check(self._L.pt_scene_cull_layout(self._h, eps, _ip(counts), _ip(st), nodes.ctypes.data_as(C.c_void_p)), self._L)
        return {"slot_triangle": st, "bvh": nodes, "bvh_inner_nodes": int(counts[2]), "clusters": int(counts[3]),
                "bvh_depth": bvh_depth(nodes)}

    def trace_rays(self, origins, directions, eps=1e-4):
        """Closest hit per ray (scene.cpp:114-120).  directions must be unit length (normalised as ray.h:23 does)."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        idx = np.full(len(o), -2, np.int32)
        t = np.zeros(len(o), np.float32)
        _check(self._L.pt_trace_rays_host(self._h, len(o), _fp(o), _fp(d), eps, _ip(idx), _fp(t)), self._L)
        return idx, t

    def render_device(self, params, d_sum, d_sum2, d_count, stream=None, want_stats=False):
        """d_* are raw device pointers (ints), e.g. torch tensors' data_ptr(); stream is a hipStream_t value."""
        st = RenderStats()
        _check(self._L.pt_render_device(self._h, C.byref(params), C.c_void_p(d_sum), C.c_void_p(d_sum2),
                                      C.c_void_p(d_count), C.c_void_p(stream or 0),
                                      C.byref(st) if want_stats else None), self._L)
        return st.as_dict() if want_stats else None

    def close(self):
        if self._h:
            self._L.pt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def band_rows(params, L=None):
    """Rows the accumulator buffers of a call with these RenderParams hold (pt_band_rows)."""
    return (L or lib()).pt_band_rows(C.byref(params))


def interleaved_rows(height, row_begin, row_stride):
    """Image rows, in buffer order, of the interleaved band (row_begin, height, row_stride): -1 for buffer rows beyond the image."""
    out = []
    for t0 in range(row_begin, height, 8 * max(1, row_stride)):
        out += [y if y < height else -1 for y in range(t0, t0 + 8)]
    return np.array(out, np.int64)


class Session:
    """pt_session: one row band's accumulators kept on the device between pass slices (progressive driver)."""

    def __init__(self, scene, width, height, rows=None, row_stride=0):
        r0, r1 = rows if rows is not None else (0, height)
        self._scene, self._L = scene, scene._L
        self.width, self.height, self.rows, self.row_stride = width, height, (r0, r1), row_stride
        self._h = C.c_void_p()
        if row_stride > 1:
            _check(self._L.pt_session_create_strided(scene._h, width, height, r0, r1, row_stride, C.byref(self._h)), self._L)
        else:
            _check(self._L.pt_session_create(scene._h, width, height, r0, r1, C.byref(self._h)), self._L)

    def render(self, pass_begin, pass_count, mrr, *, eps=1e-4, error=-1.0, seed=42, want_stats=False):
        p = RenderParams(self.width, self.height, self.rows[0], self.rows[1], pass_begin, pass_count, mrr, eps, error, seed, 0, self.row_stride)
        st = RenderStats()
        _check(self._L.pt_session_render(self._h, C.byref(p), C.byref(st) if want_stats else None), self._L)
        return st.as_dict() if want_stats else None

    def read(self):
        n = band_rows(RenderParams(self.width, self.height, self.rows[0], self.rows[1], 0, 0, 0, 0, 0, 0, 0, self.row_stride), self._L) * self.width
        s, s2, c = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
        _check(self._L.pt_session_read(self._h, _fp(s), _fp(s2), _ip(c)), self._L)
        return s, s2, c

    def clear(self):
        _check(self._L.pt_session_clear(self._h), self._L)

    def close(self):
        if self._h:
            self._L.pt_session_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bvh_depth(nodes):
    """Levels of a box tree from its 64-byte nodes (BvhNode::meta: children - 1 in bits 8-10, leaf flag bit 11, base in bits
    12-31; children are consecutive nodes with larger indices): root = 1, no tree = 0."""
    n = len(nodes)
    if n == 0:
        return 0
    meta = np.ascontiguousarray(nodes).view(np.uint32).reshape(n, 16)[:, 3]
    level = np.ones(n, np.int64)
    for w in range(n):
        m = int(meta[w])
        if not (m >> 11) & 1:
            base, k = m >> 12, ((m >> 8) & 7) + 1
            level[base:base + k] = level[w] + 1
    return int(level.max())


class Frame:
    """pt_frame: one image on several devices from one host program -- row bands, one gather to the first device."""

    def __init__(self, scene, devices, width, height, flags=0):
        self._scene, self._L = scene, scene._L
        self.width, self.height = width, height
        devs = np.ascontiguousarray(devices, np.int32)
        self._h = C.c_void_p()
        _check(self._L.pt_frame_create(scene._h, _ip(devs), len(devs), width, height, flags, C.byref(self._h)), self._L)

    def info(self):
        n = C.c_int32()
        _check(self._L.pt_frame_info(self._h, C.byref(n), None, None, None), self._L)
        rows, dev, tr = np.zeros(2 * n.value, np.int32), np.zeros(n.value, np.int32), C.c_int32()
        _check(self._L.pt_frame_info(self._h, C.byref(n), _ip(rows), _ip(dev), C.byref(tr)), self._L)
        stride = C.c_int32()
        _check(self._L.pt_frame_row_stride(self._h, C.byref(stride)), self._L)
        return {"bands": n.value, "rows": rows.reshape(-1, 2).tolist(), "devices": dev.tolist(), "transport": TRANSPORT_NAMES[tr.value],
                "row_stride": stride.value}

    def render(self, pass_begin, pass_count, mrr, *, eps=1e-4, error=-1.0, seed=42, want_stats=False):
        p = RenderParams(self.width, self.height, 0, self.height, pass_begin, pass_count, mrr, eps, error, seed, 0)
        st = RenderStats()
        _check(self._L.pt_frame_render(self._h, C.byref(p), C.byref(st) if want_stats else None), self._L)
        return st.as_dict() if want_stats else None

    def band_kernel_ms(self):
        """Kernel time of every band of the last render(want_stats=True), -1 where there is none."""
        ms = np.zeros(self.info()["bands"], np.float32)
        _check(self._L.pt_frame_band_kernel_ms(self._h, _fp(ms)), self._L)
        return ms.tolist()

    def gather(self):
        _check(self._L.pt_frame_gather(self._h), self._L)

    def wait(self):
        _check(self._L.pt_frame_wait(self._h), self._L)

    def read(self):
        n = self.width * self.height
        s, s2, c = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
        _check(self._L.pt_frame_read(self._h, _fp(s), _fp(s2), _ip(c)), self._L)
        return s, s2, c

    def clear(self):
        _check(self._L.pt_frame_clear(self._h), self._L)

    def close(self):
        if self._h:
            self._L.pt_frame_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_version(library=None):
    """ncclGetVersion of the RCCL the frame's gather would use; raises PtError if it cannot be loaded.  Needs no GPU."""
    L = library or lib()
    v = C.c_int32()
    _check(L.pt_rccl_available(C.byref(v)), L)
    return v.value


def resolve(width, height, s, s2, c, gamma=None):
    """main.cpp:162-201: returns (bgr uint8 [H,W,3], dispersion float32[3] = max, min, average)."""
    if gamma is None:
        gamma = np.float32(1) / np.float32(2.2)   # config.h:25
    bgr = np.zeros((height, width, 3), np.uint8)
    disp = np.zeros(3, np.float32)
    s = np.ascontiguousarray(s, np.float32)
    s2 = np.ascontiguousarray(s2, np.float32)
    c = np.ascontiguousarray(c, np.int32)
    _check(lib().pt_resolve(width, height, _fp(s), _fp(s2), _ip(c), C.c_float(gamma),
                            bgr.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(disp)))
    return bgr, disp


def resolve_float(width, height, s, s2, c, gamma=None):
    """main.cpp:162-185: (rgb float32 [H,W,3] tonemapped image, dispersion float32[3])."""
    if gamma is None:
        gamma = np.float32(1) / np.float32(2.2)
    rgb = np.zeros((height, width, 3), np.float32)
    disp = np.zeros(3, np.float32)
    _check(lib().pt_resolve_float(width, height, _fp(np.ascontiguousarray(s, np.float32)), _fp(np.ascontiguousarray(s2, np.float32)),
                                  _ip(np.ascontiguousarray(c, np.int32)), C.c_float(gamma), _fp(rgb), _fp(disp)))
    return rgb, disp


def post_filter(rgb, gauss=0, median=0, device=0):
    """-GAUSS / -MEDIAN (main.cpp:187-192) on the GPU; returns the filtered float image."""
    out = np.ascontiguousarray(rgb, np.float32).copy()
    h, w, _ = out.shape
    _check(lib().pt_post_filter_host(device, w, h, _fp(out), gauss, median))
    return out


def quantize(rgb, c):
    h, w, _ = rgb.shape
    bgr = np.zeros((h, w, 3), np.uint8)
    _check(lib().pt_quantize(w, h, _fp(np.ascontiguousarray(rgb, np.float32)), _ip(np.ascontiguousarray(c, np.int32)),
                             bgr.ctypes.data_as(C.POINTER(C.c_uint8))))
    return bgr


def write_bmp(path, bgr):
    h, w, _ = bgr.shape
    bgr = np.ascontiguousarray(bgr, np.uint8)
    _check(lib().pt_write_bmp(path.encode(), w, h, bgr.ctypes.data_as(C.POINTER(C.c_uint8))))
