#!/usr/bin/env python3
"""Triangle-count sweep across the switch between the sphere-tree path (small-scene kernels) and the box-tree path (big-scene
kernels): the torus replicated 1 ... 32 times in the room (270 ... 8 206 triangles), 1920x1080 x 64 spp, -MRR 8, each scene once
through either path (the test-hook library's `big_threshold` forces it) and once as the shipped library chooses.

    python tools/t_sweep.py [--instances 1,2,4,6,7,8,9,12,16,32] [--spp 64] > profiles/rNN_t_sweep.jsonl

One JSON line per (scene, path): Msamples/s of the statistics-free launch (median of 3, HIP events), table-build seconds, frame
digest -- the two paths must render the same frame bit for bit.
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", default="1,2,4,6,7,8,9,12,16,32")
    ap.add_argument("--spp", type=int, default=64)
    a = ap.parse_args()
    import torch
    import make_replicated_scene as M
    pt = importlib.import_module("path-tracing_amd")
    H_ = pt.load_library(os.path.join(ROOT, "path-tracing_amd", "lib", "libpt_testhooks.so"))
    tmp = tempfile.mkdtemp() + "/"
    dev = torch.device("cuda", 0)
    W, H = 1920, 1080
    buf = torch.zeros(7 * W * H, dtype=torch.float32, device=dev)
    ptrs = (buf.data_ptr(), buf.data_ptr() + 12 * W * H, buf.data_ptr() + 24 * W * H)
    stream = torch.cuda.current_stream(dev)
    for inst in [int(x) for x in a.instances.split(",")]:
        name = f"x{inst}.obj"
        tri = M.generate(os.path.join(ROOT, "models"), tmp, name, inst)
        digests = {}
        for path, lib, thr in (("sphere_trees", H_, 16384), ("box_tree", H_, 0), ("shipped", pt.lib(), None)):
            if thr is not None:
                lib.pt_test_set_mutation(b"reset", 0.0)
                lib.pt_test_set_mutation(b"big_threshold", float(thr))
            sc = pt.Scene.load_obj(tmp, name, device=0, library=lib)
            p = pt.RenderParams(W, H, 0, H, 0, a.spp, 8, 1e-4, -1.0, 42)
            buf.zero_()
            sc.render_device(p, *ptrs, stream=stream.cuda_stream)      # warm-up: builds and uploads the hierarchy
            torch.cuda.synchronize(dev)
            ms = []
            for _ in range(3):
                buf.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                sc.render_device(p, *ptrs, stream=stream.cuda_stream)
                e1.record(stream)
                torch.cuda.synchronize(dev)
                ms.append(e0.elapsed_time(e1))
            ms.sort()
            digests[path] = hashlib.sha1(buf.cpu().numpy().tobytes()).hexdigest()[:12]
            lay = sc.cull_layout()
            print(json.dumps({"instances": inst, "triangles": tri, "path": path, "msamples_per_s": W * H * a.spp / ms[1] / 1e3,
                              "kernel_ms": ms[1], "hierarchy_build_s": sc.timings()["hierarchy_build_s"], "bvh_nodes": len(lay["bvh"]),
                              "bvh_depth": lay["bvh_depth"], "clusters": lay["clusters"], "frame": digests[path]}), flush=True)
            sc.close()
            if thr is not None:
                lib.pt_test_set_mutation(b"reset", 0.0)
        if len(set(digests.values())) != 1:
            print(json.dumps({"instances": inst, "error": "the paths render different frames", "digests": digests}), flush=True)
            sys.exit(3)


if __name__ == "__main__":
    main()
