"""A path's last segment searches the emitters first (pt_kernels.hip, RenderArgs::emis_*).

After depth + 1 == mrr no ray follows (Ray::IsValid, ray.h:52-54), so the segment can only matter by hitting an emitter:
the statistics-free, skybox-free small-scene kernel finds the closest hit among the emitters alone and runs the full
search only for rays that hit one.  Nothing of that may show: frames must equal the oracle's (and the frames of the same
library with the filter switched off through the test hook) bit for bit -- for every path length, with emitters in the
large class (the light of Tor.obj), in a sphere-tree cluster (an emissive torus), in both, and nowhere.

Big scenes (box tree) have the one-search form of it: the table builder keeps a big scene's few emitters in the large class
whatever their size, and the conservative test of their records alone decides which rays of a last segment are searched at
all.  Same checks on a replicated scene: the light alone (test active), emissive tori as well (emitters under the box tree:
test off), no emitter (no last segment is searched)."""
import importlib
import os
import shutil

import numpy as np
import pytest

import sys

import oracle_lib as O

pt = importlib.import_module("path-tracing_amd")
pytestmark = pytest.mark.gpu


def _scene_dir(tmp_path, models_dir, torus_emits, light_emits, replicas=0):
    d = str(tmp_path) + "/"
    if replicas:
        sys.path.insert(0, os.path.join(os.path.dirname(models_dir), "tools"))
        import make_replicated_scene as M
        assert M.generate(models_dir, d, "Tor.obj", replicas) > pt.BIG_SCENE_TRIANGLES      # the box-tree kernel
    else:
        shutil.copy(os.path.join(models_dir, "Tor.obj"), d + "Tor.obj")
    out, cur = [], None
    for line in open(os.path.join(models_dir, "Tor.mtl")):
        tok = line.split()
        if tok and tok[0] == "newmtl":
            cur = tok[1]
        if tok and tok[0] == "Ke":
            if cur == "4":      # the torus (usemtl 4)
                line = "Ke 0.8 0.6 0.2\n" if torus_emits else "Ke 0 0 0\n"
            elif not light_emits:
                line = "Ke 0 0 0\n"
        out.append(line)
    text = "".join(out)
    if torus_emits and "Ke 0.8 0.6 0.2" not in text:      # the torus material has no Ke line of its own: give it one
        text = text.replace("newmtl 4\n", "newmtl 4\nKe 0.8 0.6 0.2\n")
    open(d + "Tor.mtl", "w").write(text)
    return d


@pytest.mark.parametrize("torus_emits,light_emits,replicas", [(False, True, 0), (True, True, 0), (True, False, 0), (False, False, 0),
                                                              (False, True, 9), (True, True, 9), (False, False, 9)])
def test_frames_equal_the_oracle_for_every_path_length(tmp_path, models_dir, torus_emits, light_emits, replicas):
    d = _scene_dir(tmp_path, models_dir, torus_emits, light_emits, replicas)
    g = pt.Scene.load_obj(d, "Tor.obj", device=0)
    o = O.Scene.load(d, "Tor.obj")
    hooks = pt.load_library(pt.TESTHOOKS_LIB_PATH)
    hooks.pt_test_set_mutation(b"reset", 0.0)
    W, H, spp = (96, 64, 6) if not replicas else (64, 48, 4)
    if replicas:      # the light sits in the large class exactly when it is the scene's only emitter
        n_large = int((g.cull_layout()["slot_triangle"][(len(g.cull_layout()["bvh"]) - g.cull_layout()["bvh_inner_nodes"]) * 8:] >= 0).sum())
        assert n_large == (14 if light_emits and not torus_emits else 12)
    contributing = 0
    for mrr in (1, 2, 3, 8):
        rs, rs2, rc, rst = O.render(o, W, H, spp, mrr)
        q = g.render_host(W, H, spp, mrr, want_stats=False)          # the filtered search
        assert np.array_equal(q[2], rc), (mrr, torus_emits, light_emits)
        assert np.array_equal(q[0].view(np.uint32), rs.view(np.uint32)) and np.array_equal(q[1].view(np.uint32), rs2.view(np.uint32))
        s = g.render_host(W, H, spp, mrr)                            # the statistics instantiation: the full search on every segment
        assert np.array_equal(s[0].view(np.uint32), rs.view(np.uint32)) and s[3]["segments"] == rst["segments"]
        try:
            hooks.pt_test_set_mutation(b"no_last_segment_filter", 1.0)
            h = pt.Scene.load_obj(d, "Tor.obj", device=0, library=hooks)
            u = h.render_host(W, H, spp, mrr, want_stats=False)      # the same kernel with the filter off
        finally:
            hooks.pt_test_set_mutation(b"reset", 0.0)
        assert np.array_equal(u[0].view(np.uint32), q[0].view(np.uint32)) and np.array_equal(u[2], q[2])
        contributing += int(rc.sum())
    assert (contributing > 0) == (torus_emits or light_emits)
