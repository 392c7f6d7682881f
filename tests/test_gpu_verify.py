"""Every segment of a full frame against the reference's all-triangles loop, on the device.

libpt_verify.so is the product library compiled with -DPT_VERIFY_BRUTE: after the culled closest-hit search of every
path segment, the same lane runs Scene::TraceRay's loop as written (scene.cpp:116-120: every triangle, in index order,
through Triangle::Intersect) for its own ray and the two (distance bits, triangle index) results are compared.  That
turns "sampled rows against the CPU oracle" into EVERY segment of the frame -- about 10^9 at BASELINE configs[1] --
without needing the CPU.  The frame itself must be the shipped library's frame bit for bit (same digest), so what is
verified is the shipped search, not a variant of it.
"""
import hashlib
import importlib
import os
import sys

import numpy as np
import pytest

pt = importlib.import_module("path-tracing_amd")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vlib():
    assert pt.device_count() >= 1, "no HIP device: the integrator has no CPU fallback"
    L = pt.load_library(pt.VERIFY_LIB_PATH)
    L.pt_test_set_mutation(b"reset", 0.0)
    yield L
    L.pt_test_set_mutation(b"reset", 0.0)


def _digest(s, s2, c):
    return hashlib.sha256(s.tobytes() + s2.tobytes() + c.tobytes()).hexdigest()


def _replica(tmp, instances):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_replicated_scene as M
    d = str(tmp) + "/"
    name = f"TorX{instances}.obj"
    M.generate(os.path.join(ROOT, "models"), d, name, instances)
    return d, name


def test_every_segment_of_config1(models_dir, vlib):
    """BASELINE configs[1], the whole frame: 1920 x 1080 x 64 spp x MRR 8, about 1.05e9 segments x 270 triangles."""
    W, H, spp, mrr = 1920, 1080, 64, 8
    v = pt.Scene.load_obj(models_dir, "Tor.obj", device=0, library=vlib)
    s, s2, c, st = v.render_host(W, H, spp, mrr, error=-1.0)
    assert st["verify_checked"] == st["segments"] > 7.5 * W * H * spp
    assert st["verify_mismatches"] == 0
    g = pt.Scene.load_obj(models_dir, "Tor.obj", device=0)
    q = g.render_host(W, H, spp, mrr, error=-1.0, want_stats=False)
    assert _digest(*q[:3]) == _digest(s, s2, c)            # the shipped library renders exactly the verified frame
    assert g.render_host(16, 16, 1, 2)[3]["verify_checked"] == 0   # and carries no verification code


def test_every_segment_with_adaptive_sampling_and_other_eps(models_dir, vlib):
    v = pt.Scene.load_obj(models_dir, "Tor.obj", device=0, library=vlib)
    for eps, err in ((1e-4, 0.001), (1e-3, -1.0), (1e-5, -1.0)):
        st = v.render_host(960, 540, 32, 8, error=err, eps=eps)[3]
        assert st["verify_checked"] == st["segments"] > 0 and st["verify_mismatches"] == 0, (eps, err, st)


@pytest.mark.parametrize("instances,W,H,spp", [(64, 1920, 1080, 4), (195, 960, 540, 2)])
def test_every_segment_of_the_replicated_scene(tmp_path, vlib, instances, W, H, spp):
    """BASELINE configs[4] geometry (x64 -> 16 398 triangles, x195 -> 49 934): the deep-queue kernel with the pair
    pre-filter, every segment against all triangles."""
    d, name = _replica(tmp_path, instances)
    v = pt.Scene.load_obj(d, name, device=0, library=vlib)
    s, s2, c, st = v.render_host(W, H, spp, 8, error=-1.0)
    assert st["verify_checked"] == st["segments"] > 7 * W * H * spp
    assert st["verify_mismatches"] == 0
    g = pt.Scene.load_obj(d, name, device=0)
    q = g.render_host(W, H, spp, 8, error=-1.0, want_stats=False)
    assert _digest(*q[:3]) == _digest(s, s2, c)


def test_the_check_notices_a_cull_that_is_too_tight(models_dir, vlib):
    """Negative control: with every sphere's r^2 halved the culled search loses real hits and the comparison counts them."""
    vlib.pt_test_set_mutation(b"sphere_r2", 0.5)
    try:
        v = pt.Scene.load_obj(models_dir, "Tor.obj", device=0, library=vlib)
        st = v.render_host(480, 270, 8, 8, error=-1.0)[3]
    finally:
        vlib.pt_test_set_mutation(b"reset", 0.0)
    assert st["verify_checked"] == st["segments"] and st["verify_mismatches"] > 100


@pytest.mark.parametrize("seed,n_small,n_large,n_dup,sky", [(11, 300, 6, 20, False), (12, 40, 30, 10, True), (13, 3000, 10, 40, False),
                                                          (14, 7000, 4, 30, True), (15, 2040, 0, 0, False)])
def test_every_segment_of_random_scenes(tmp_path, vlib, seed, n_small, n_large, n_dup, sky):
    """Randomised scenes (tests/test_gpu_fuzz.py's generator: slivers, duplicates, coplanar overlaps, interleaved classes;
    300 to 7000 triangles, i.e. both the sphere-tree and the box-tree kernel), with and without a skybox: every segment
    of a 640x360x8 frame against the all-triangles loop on the device."""
    import test_gpu_fuzz as F
    import oracle_lib as O
    d = str(tmp_path) + "/"
    F._random_scene(d, seed, n_small, n_large, n_dup)
    v = pt.Scene.load_obj(d, "f.obj", device=0, library=vlib)
    if sky:
        bgr = np.random.default_rng(seed).integers(0, 256, (9, 16, 3)).astype(np.uint8)
        O.write_bmp(d + "sky.bmp", bgr)
        v.set_skybox(d + "sky.bmp")
    st = v.render_host(640, 360, 8, 8, error=-1.0)[3]
    # These scenes have near-degenerate slivers, which the reference "hits" at any distance; the next segment then starts
    # millions of units away, outside the envelope the culling margins are derived for.  Such rays get every triangle as
    # a candidate (before they did, this test found them: 10-14 of 5 million segments differed).
    assert st["verify_checked"] == st["segments"] > 640 * 360 * 8
    assert st["verify_mismatches"] == 0


def test_smallest_queues_find_every_hit(models_dir, tmp_path):
    """libpt_verify_tiny.so = the verification build with the work queues at their smallest legal sizes (64-entry node stack,
    128-entry pair queue, small and big scenes): almost every round of the big-scene walk then takes the rare paths --
    the queue does not fit, a prefix of the lanes commits, exact rounds are forced early -- and every segment is still
    compared with the all-triangles loop.  The frames must be the shipped library's frames."""
    path = os.path.join(os.path.dirname(pt.VERIFY_LIB_PATH), "libpt_verify_tiny.so")
    L = pt.load_library(path)
    L.pt_test_set_mutation(b"reset", 0.0)
    v = pt.Scene.load_obj(models_dir, "Tor.obj", device=0, library=L)
    s, s2, c, st = v.render_host(960, 540, 16, 8, error=-1.0)
    assert st["verify_checked"] == st["segments"] > 0 and st["verify_mismatches"] == 0
    assert st["partial_commit_rounds"] > 1000
    g = pt.Scene.load_obj(models_dir, "Tor.obj", device=0)
    assert _digest(*g.render_host(960, 540, 16, 8, error=-1.0, want_stats=False)[:3]) == _digest(s, s2, c)
    d, name = _replica(tmp_path, 64)
    v = pt.Scene.load_obj(d, name, device=0, library=L)
    s, s2, c, st = v.render_host(960, 540, 2, 8, error=-1.0)
    assert st["verify_checked"] == st["segments"] > 0 and st["verify_mismatches"] == 0
    assert st["partial_commit_rounds"] > 100000
    g = pt.Scene.load_obj(d, name, device=0)
    assert _digest(*g.render_host(960, 540, 2, 8, error=-1.0, want_stats=False)[:3]) == _digest(s, s2, c)


def test_smallest_queues_on_a_deep_tree_of_nested_triangles(tmp_path):
    """The box tree's depth follows the geometry (build_bvh_sah), and a node stack that is full commits its top item whatever the
    children need: the stack then holds a depth-first path, up to 7 siblings per level beyond its nominal size -- what the 64
    entries of slack and the depth bound (kMaxBvhDepth) are for.  Nested, geometrically growing triangles on the camera axis
    (tools/make_nested_scene.py) give a tree deeper than any other scene of the suite, rays through the centre keep most
    children of every node, and the smallest legal stack (libpt_verify_tiny.so: 64 entries) forces commit after commit: every
    segment is compared with the all-triangles loop, for the SAH tree and for the uniform-depth tree a deeper scene falls back to."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_nested_scene as N
    d = str(tmp_path) + "/"
    N.generate(os.path.join(ROOT, "models"), d, "Nested.obj", 12000)
    L = pt.load_library(os.path.join(os.path.dirname(pt.VERIFY_LIB_PATH), "libpt_verify_tiny.so"))
    g = pt.Scene.load_obj(d, "Nested.obj", device=0)
    shipped = _digest(*g.render_host(480, 270, 4, 8, error=-1.0, want_stats=False)[:3])
    for cap, deep in ((9, True), (4, False)):
        L.pt_test_set_mutation(b"reset", 0.0)
        L.pt_test_set_mutation(b"bvh_depth_cap", float(cap))
        try:
            v = pt.Scene.load_obj(d, "Nested.obj", device=0, library=L)
            depth = v.cull_layout()["bvh_depth"]
            assert (depth >= 7) if deep else (depth == 5), depth          # 12 014 triangles: 8^5 holds them in the uniform tree
            s, s2, c, st = v.render_host(480, 270, 4, 8, error=-1.0)
        finally:
            L.pt_test_set_mutation(b"reset", 0.0)
        assert st["verify_checked"] == st["segments"] > 480 * 270 * 4 and st["verify_mismatches"] == 0, (cap, st)
        assert st["partial_commit_rounds"] > 10000, st
        assert _digest(s, s2, c) == shipped, cap


def test_both_forms_of_the_box_test_keep_what_the_exact_test_keeps():
    """pt_kernels.hip has the box tree's child test twice: in float32 (box_children_kept, what ships) and in packed half precision
    (box_children_kept_h, -DPT_BOX_F16=1: built, verified and measured in round 4 -- 3.4 % slower, profiles/r04_ab_logs.txt slab16
    -- and kept as a compile-time variant).  The test build runs BOTH on 200 000 random (node, ray, t_best) items
    (tools/box_mask_probe.py: frames of every size, origins inside, near and far, nearly and exactly axis-parallel rays): neither
    may drop a child the exact float64 test keeps, and each agrees with its numpy restatement (tests/bvh_emulation.py) except where
    v_rcp_f32's last bit decides."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import box_mask_probe as P
    import bvh_emulation as B
    n = 200_000
    raw, o, d, t_best = P.items(n)
    L = pt.load_library(os.path.join(os.path.dirname(pt.LIB_PATH), "libpt_testhooks.so"))
    L.pt_test_box_masks.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int32, C.POINTER(C.c_uint32)]
    rays = np.ascontiguousarray(np.concatenate([o, d], 1), np.float32)
    out = np.zeros(2 * n, np.uint32)
    assert L.pt_test_box_masks(raw.ctypes.data_as(C.c_void_p), rays.ctypes.data_as(C.POINTER(C.c_float)), t_best.ctypes.data_as(C.POINTER(C.c_float)),
                               C.c_float(5e-7), n, out.ctypes.data_as(C.POINTER(C.c_uint32))) == 0, L.pt_last_error()
    bits = lambda m: ((m[:, None] >> np.arange(8)[None, :]) & 1).astype(bool)
    dev32, dev16, devmix = bits(out[0::2]), bits(out[1::2] & 0xFF), bits(out[1::2] >> 8)
    t = B.decode(raw)
    ex = P.exact_keep(t, o, d, t_best)
    assert ex.sum() > 20000
    assert not (ex & ~dev32).any() and not (ex & ~dev16).any()
    assert np.array_equal(devmix, dev32)            # box_children_kept_mix: the same arithmetic, so the same masks, always
    node = np.arange(n)
    assert (dev32 != B.children_kept(t, node, o, d, t_best, 5e-7)).any(1).mean() < 0.05
    assert (dev16 != B.children_kept_f16(t, node, o, d, t_best)).any(1).mean() < 0.02
    # how loose: the half-precision form keeps a little more than the float form, both far less than everything
    assert dev32.sum() <= dev16.sum() < 1.25 * dev32.sum()
