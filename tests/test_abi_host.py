"""CPU-side checks of the C ABI: the library loads, exports every declared symbol, and its host-side pieces
(OBJ/MTL ingestion, resolve, BMP writer, error codes) agree with the oracle.  No GPU compute here."""
import ctypes as C
import hashlib
import importlib
import os
import re

import numpy as np
import pytest

import oracle_lib as O

pt = importlib.import_module("path-tracing_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(pt.LIB_PATH):
        pt.build()


def test_header_symbols_are_exported():
    header = open(os.path.join(ROOT, "include", "pt_hip.h")).read()
    declared = set(re.findall(r"\b(pt_[a-z_]+)\s*\(", header))
    declared -= {"pt_status"}
    assert declared == set(pt.ABI_SYMBOLS)
    for name in declared:
        assert getattr(pt.lib(), name) is not None
    assert pt.lib().pt_abi_version() == 5 == pt.PT_ABI_VERSION
    assert not hasattr(pt.lib(), "pt_test_set_mutation")      # the test hooks exist only in the test builds


def test_struct_layouts_match_header():
    assert C.sizeof(pt.RenderParams) == 48
    assert C.sizeof(pt.RenderStats) == 96


def test_loader_matches_oracle_bit_for_bit(models_dir, oracle_scene):
    s = pt.Scene.load_obj(models_dir, "Tor.obj", device=-1)
    assert s.counts() == (270, 5)
    t, m = s.triangles()
    ot, om = oracle_scene.triangles()
    assert np.array_equal(t.view(np.uint32), ot.view(np.uint32))
    assert np.array_equal(m, om)
    assert np.array_equal(s.materials().view(np.uint32), oracle_scene.materials().view(np.uint32))


def test_loader_quirks(tmp_path):
    # comments are not skipped line-wise, unknown tokens are ignored one by one, usemtl is an integer index,
    # only the first vertex's normal is used, faces may omit vt/vn (scene.cpp:37-106)
    (tmp_path / "q.mtl").write_text("newmtl 0\nKd 0.5 0.25 1\nNs 10\nKs 1 1 1\nnewmtl 1\nKe 1 0 0\nKd 1 1 1\n")
    (tmp_path / "q.obj").write_text(
        "# v 9 9 9 is a comment, but its tokens are still parsed\nmtllib q.mtl\n"
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nvn 0 0 2\nvn 1 0 0\n"
        "usemtl 1\nf 2 3 4\nusemtl 0\nf 2/7/1 3//2 4\nf 1 2 3 4\n")
    d = str(tmp_path) + "/"
    s = pt.Scene.load_obj(d, "q.obj", device=-1)
    o = O.Scene.load(d, "q.obj")
    t, m = s.triangles()
    ot, om = o.triangles()
    assert len(t) == 3 and list(m) == [1, 0, 0]
    assert np.array_equal(t.view(np.uint32), ot.view(np.uint32)) and np.array_equal(m, om)
    # "v 9 9 9" inside the comment became vertex 1, so face "2 3 4" is (0,0,0),(1,0,0),(0,1,0)
    assert np.allclose(t[0, 4:13], [0, 0, 0, 1, 0, 0, 0, 1, 0])
    assert np.allclose(t[1, 0:3], [0, 0, 1])      # vn 1 = (0,0,2) normalised; vn of the first vertex only
    assert np.allclose(t[2, 4:13], [9, 9, 9, 0, 0, 0, 1, 0, 0])   # "f 1 2 3 4": the 4th index is a stray token
    assert np.array_equal(s.materials().view(np.uint32), o.materials().view(np.uint32))


def test_error_codes(tmp_path):
    with pytest.raises(pt.PtError) as e:
        pt.Scene.load_obj(str(tmp_path) + "/", "missing.obj", device=-1)
    assert e.value.status == 2
    (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(pt.PtError) as e:
        pt.Scene.load_obj(str(tmp_path) + "/", "bad.obj", device=-1)
    assert e.value.status == 3
    (tmp_path / "nomtl.obj").write_text("mtllib none.mtl\n")
    with pytest.raises(pt.PtError) as e:
        pt.Scene.load_obj(str(tmp_path) + "/", "nomtl.obj", device=-1)
    assert e.value.status == 2
    with pytest.raises(pt.PtError) as e:
        pt.Scene.create(np.zeros((1, 14), np.float32), np.array([3], np.int32), np.zeros((1, 10), np.float32), device=-1)
    assert e.value.status == 1


def test_render_without_device_fails_loudly(models_dir):
    s = pt.Scene.load_obj(models_dir, "Tor.obj", device=-1)
    with pytest.raises(pt.PtError) as e:
        s.render_host(8, 8, 1, 2)
    assert e.value.status == 4   # PT_ERR_NO_DEVICE: there is no CPU fallback
    if pt.device_count() == 0:
        with pytest.raises(pt.PtError) as e:
            pt.Scene.load_obj(models_dir, "Tor.obj", device=0)
        assert e.value.status == 4


def test_resolve_and_bmp_match_oracle(tmp_path, oracle_scene):
    W, H = 37, 21   # row padding: 3*37 = 111 -> 1 pad byte
    s, s2, c, _ = O.render(oracle_scene, W, H, 24, 6, rng=O.RNG_COUNTER, trig=O.TRIG_PORTABLE)
    bgr, disp = pt.resolve(W, H, s, s2, c)
    obgr, odisp = O.resolve(W, H, s, s2, c)
    assert (c > 0).sum() > 20
    assert np.array_equal(bgr, obgr)
    assert np.array_equal(disp.view(np.uint32), odisp.view(np.uint32))
    a, b = str(tmp_path / "a.bmp"), str(tmp_path / "b.bmp")
    pt.write_bmp(a, bgr)
    O.write_bmp(b, obgr)
    da = open(a, "rb").read()
    assert da == open(b, "rb").read()
    assert len(da) == 54 + 112 * H and da[:2] == b"BM"


def test_reference_frame_through_product_resolve(tmp_path, oracle_scene):
    # the reference's 64x64x4 MRR 3 frame: oracle accumulators -> product resolve + BMP -> same md5
    s, s2, c, _ = O.render(oracle_scene, 64, 64, 4, 3, error=0.001, rng=O.RNG_SEQUENTIAL, trig=O.TRIG_LIBM)
    bgr, disp = pt.resolve(64, 64, s, s2, c)
    path = str(tmp_path / "r.bmp")
    pt.write_bmp(path, bgr)
    assert hashlib.md5(open(path, "rb").read()).hexdigest() == "994782793a83d584cb8f0815a5a65b90"
    assert "%f" % disp[2] == "0.986328"


def test_cli_fails_loudly_without_gpu(models_dir, tmp_path):
    import subprocess
    exe = os.path.join(ROOT, "path-tracing_amd", "bin", "pt_render")
    if not os.path.exists(exe):
        pt.build()
    if pt.device_count() > 0:
        pytest.skip("a GPU is present; the GPU suite covers the CLI")
    r = subprocess.run([exe, "--W", "8", "--H", "8", "-RPP", "1", "-MODEL_PATH", models_dir], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


def test_table_limits_are_refused_not_truncated(tmp_path):
    """The kernels pack a slot into 24 bits of a (ray, slot) pair and a box-tree node's child base into 20 bits of
    BvhNode::meta; a hierarchy beyond either must be refused (synthetic counts: no 16 M-triangle scene needed)."""
    L = pt.lib()
    assert L.pt_table_limits_check((1 << 24) - 1, 0, 8) == pt.PT_OK
    assert L.pt_table_limits_check((1 << 23) - 1, (1 << 20) - 1, 8) == pt.PT_OK
    assert L.pt_table_limits_check(1 << 24, 10, 3) == 7 and b"24 bits" in L.pt_last_error()          # PT_ERR_UNSUPPORTED
    assert L.pt_table_limits_check(1000, 1 << 20, 3) == 7 and b"20 bits" in L.pt_last_error()
    assert L.pt_table_limits_check(1000, 10, 9) == 7 and b"levels" in L.pt_last_error()
    # a leaf's base is its first slot / 8 in the same 20 bits: with a box tree the slots stay below 2^23
    assert L.pt_table_limits_check((1 << 23) - 1, 1000, 3) == pt.PT_OK
    assert L.pt_table_limits_check(1 << 23, 1000, 3) == 7 and b"first slot / 8" in L.pt_last_error()
    assert L.pt_table_limits_check(1 << 23, 0, 3) == pt.PT_OK                       # (no box tree: only the 24 bits of a pair)
    # the box tree's depth is variable and the walk's stack slack bounds it (pt_hip.h: PT_MAX_BVH_DEPTH)
    assert L.pt_table_limits_check_tree(1000, 100, 3, 9) == pt.PT_OK
    assert L.pt_table_limits_check_tree(1000, 100, 3, 10) == 7 and b"levels" in L.pt_last_error() and b"stack slack" in L.pt_last_error()
    # what the builder produces for a scene of the size class the advisor worried about stays far inside: slots per triangle
    # and nodes per triangle of the biggest scene of the suite
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_replicated_scene as M
    d = str(tmp_path) + "/"
    M.generate(os.path.join(ROOT, "models"), d, "x64.obj", 64)
    s = pt.Scene.load_obj(d, "x64.obj", device=-1)
    lay = s.cull_layout()
    n_tri = s.counts()[0]
    assert len(lay["slot_triangle"]) < 4 * n_tri and len(lay["bvh"]) < n_tri      # so 2^20 nodes are beyond the 2^23-triangle bound's reach only
                                                                                    # for trees the check above refuses
    assert 1 <= lay["bvh_depth"] <= 9
    # the tree's slots are the FIRST slots (a leaf's base = its ordinal among the leaves < node count): every leaf's base x 8
    # names slots of the tree, the large class follows
    meta = lay["bvh"].view(np.uint32).reshape(-1, 16)[:, 3]
    leaves = meta[(meta >> 11) & 1 == 1]
    assert len(leaves) == len(lay["bvh"]) - lay["bvh_inner_nodes"]
    assert sorted((leaves >> 12).tolist()) == list(range(len(leaves)))


def test_the_small_big_switch_is_where_the_header_says(tmp_path):
    import re
    import sys
    hdr = open(os.path.join(ROOT, "path-tracing_amd", "csrc", "pt_scene.hpp")).read()
    assert int(re.search(r"kBigSceneTriangles = (\d+);", hdr).group(1)) == pt.BIG_SCENE_TRIANGLES
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_replicated_scene as M
    d = str(tmp_path) + "/"
    for inst, tree in ((3, False), (4, True)):
        n = M.generate(os.path.join(ROOT, "models"), d, f"x{inst}.obj", inst)
        lay = pt.Scene.load_obj(d, f"x{inst}.obj", device=-1).cull_layout()
        assert (n > pt.BIG_SCENE_TRIANGLES) == tree == (len(lay["bvh"]) > 0), (inst, n)


def test_box_tree_depth_is_bounded(tmp_path):
    """A SAH tree deeper than the walk's stack slack allows is replaced by the uniform-depth tree over the same triangles (the
    test hook lowers the bound so that an ordinary scene trips it), and the header's constant is the builder's."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_nested_scene as N
    d = str(tmp_path) + "/"
    N.generate(os.path.join(ROOT, "models"), d, "Nested.obj", 6000)
    H = pt.load_library(os.path.join(os.path.dirname(pt.LIB_PATH), "libpt_testhooks.so"))
    try:
        H.pt_test_set_mutation(b"reset", 0.0)
        sah = pt.Scene.load_obj(d, "Nested.obj", device=-1, library=H).cull_layout()
        H.pt_test_set_mutation(b"bvh_depth_cap", 4.0)
        uni = pt.Scene.load_obj(d, "Nested.obj", device=-1, library=H).cull_layout()
    finally:
        H.pt_test_set_mutation(b"reset", 0.0)
    assert sah["bvh_depth"] >= 6 and uni["bvh_depth"] == 5          # 8^5 >= 6 014 > 8^4
    live = lambda lay: sorted(t for t in lay["slot_triangle"].tolist() if t >= 0)
    assert live(sah) == live(uni) == list(range(6014))
    assert 'define PT_MAX_BVH_DEPTH 9' in open(os.path.join(ROOT, "include", "pt_hip.h")).read()


def test_a_skybox_belongs_to_the_handle_it_is_set_on(models_dir, tmp_path):
    path = str(tmp_path / "sky.bmp")
    O.write_bmp(path, np.random.default_rng(1).integers(0, 256, (5, 7, 3)).astype(np.uint8))
    s = pt.Scene.load_obj(models_dir, "Tor.obj", device=-1)
    before = s.clone_to_device(-1)
    s.set_skybox(path)
    after = s.clone_to_device(-1)
    after.set_skybox(None)                       # clearing a copy's skybox leaves the source's alone ...
    again = s.clone_to_device(-1)                # ... so a copy made from the source now still gets one
    assert s.skybox_size() == (7, 5) and before.skybox_size() == (0, 0) and after.skybox_size() == (0, 0) and again.skybox_size() == (7, 5)
    with pytest.raises(pt.PtError):
        s.set_skybox(str(tmp_path / "missing.bmp"))
    assert s.skybox_size() == (7, 5)


def test_rccl_loads_and_exports_what_the_gather_calls():
    """pt_frame's gather uses RCCL directly (ncclCommInitAll, ncclGroupStart/End, ncclSend, ncclRecv); the library is opened
    at run time.  On a box without a GPU it must still load and resolve."""
    assert pt.rccl_version() >= 20000


def test_clone_shares_the_host_side_and_frame_needs_a_device(models_dir):
    s = pt.Scene.load_obj(models_dir, "Tor.obj", device=-1)
    c = s.clone_to_device(-1)
    assert c.counts() == s.counts() == (270, 5)
    s.cull_layout()                                   # builds the hierarchy for eps 1e-4 once ...
    built = s.timings()["hierarchy_build_s"]
    assert built > 0
    c.cull_layout()                                   # ... the copy finds it in the shared cache
    assert c.timings()["hierarchy_build_s"] == built
    s.close()                                         # either may go first
    assert c.counts() == (270, 5)
    if pt.device_count() == 0:
        with pytest.raises(pt.PtError) as e:
            pt.Frame(c, [0], 64, 64)
        assert e.value.status == 4                    # PT_ERR_NO_DEVICE: no CPU fallback


def test_threaded_resolve_is_the_sequential_resolve_bit_for_bit():
    """Above 2^18 pixels pt_resolve runs bands of rows on several cores; what depends on the pixel order (the float sum of the
    dispersion terms, ties and signed zeros of max / min) must come out as the reference's sequential loop computes it
    (main.cpp:162-185, restated in the oracle)."""
    rng = np.random.default_rng(5)
    W, H = 1024, 600
    n = W * H
    c = rng.integers(0, 4, n).astype(np.int32)
    s = (rng.random((n, 3)) * c[:, None]).astype(np.float32)
    s2 = (s * s / np.maximum(c, 1)[:, None]).astype(np.float32)          # variance estimates around +-0: signed zeros and ties
    s2[::7] *= np.float32(1.5)
    got_bgr, got_d = pt.resolve(W, H, s, s2, c)
    ref_bgr, ref_d = O.resolve(W, H, s, s2, c)
    assert np.array_equal(got_bgr, ref_bgr)
    assert np.array_equal(np.asarray(got_d, np.float32).view(np.uint32), np.asarray(ref_d, np.float32).view(np.uint32))
    got_rgb, got_d2 = pt.resolve_float(W, H, s, s2, c)
    ref_rgb, _ = O.resolve_float(W, H, s, s2, c)
    assert np.array_equal(got_rgb.view(np.uint32), np.asarray(ref_rgb, np.float32).reshape(got_rgb.shape).view(np.uint32))
    assert np.array_equal(np.asarray(got_d2, np.float32).view(np.uint32), np.asarray(got_d, np.float32).view(np.uint32))
