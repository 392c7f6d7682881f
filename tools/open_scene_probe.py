#!/usr/bin/env python3
"""Row N1 of the review: how many lanes of a wave still carry a live path, segment by segment, on an OPEN scene with a skybox?

    python tools/open_scene_probe.py [--spp 64]

Prints one JSON line per (scene, -MRR): segments per sample, live rays per wave-segment (of 64: the statistics instantiation has
one ray per lane), misses, contributing samples, and the rate of the statistics-free launch (the instantiation a caller gets).
"""
import argparse
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
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--lib", default="hip")
    ap.add_argument("--regen", default="", help="values of the test hook regen_min_dead to sweep (uses libpt_testhooks.so), e.g. 1,8,16,32,64")
    ap.add_argument("--scenes", default="closed,open_sky,open,x9_open_sky")
    a = ap.parse_args()
    if a.regen:
        a.lib = "testhooks"
    import torch
    import make_open_scene as M
    import make_replicated_scene as R
    pt = importlib.import_module("path-tracing_amd")
    L = pt.load_library(os.path.join(ROOT, "path-tracing_amd", "lib", f"libpt_{a.lib}.so"))
    tmp = tempfile.mkdtemp() + "/"
    M.generate(os.path.join(ROOT, "models"), tmp)
    R.generate(os.path.join(ROOT, "models"), tmp, "X9.obj", 9)
    M.generate(os.path.join(ROOT, "models"), tmp, name="X9Open.obj", source="X9.obj", source_dir=tmp)
    dev = torch.device("cuda", 0)
    W, H = 1920, 1080
    buf = torch.zeros(7 * W * H, dtype=torch.float32, device=dev)
    ptrs = (buf.data_ptr(), buf.data_ptr() + 12 * W * H, buf.data_ptr() + 24 * W * H)
    stream = torch.cuda.current_stream(dev)
    all_scenes = {"closed": ("closed room", os.path.join(ROOT, "models") + "/", "Tor.obj", False),
                  "closed_sky": ("closed room + skybox (sampled by the rays that slip through)", os.path.join(ROOT, "models") + "/", "Tor.obj", True), "open_sky": ("open + skybox", tmp, "TorOpen.obj", True),
                  "open": ("open, no skybox", tmp, "TorOpen.obj", False), "x9_open_sky": ("x9 open + skybox (box tree)", tmp, "X9Open.obj", True)}
    import hashlib
    for key, regen in [(k, r) for k in a.scenes.split(",") for r in ([int(x) for x in a.regen.split(",")] if a.regen else [None])]:
        label, d, name, sky = all_scenes[key]
        if regen is not None:
            if not sky:
                continue
            L.pt_test_set_mutation(b"reset", 0.0)
            L.pt_test_set_mutation(b"regen_min_dead", float(regen))
            label += f", regen_min_dead {regen}"
        sc = pt.Scene.load_obj(d, name, device=0, library=L)
        if sky:
            sc.set_skybox(tmp + "sky.bmp")
        for mrr in (8, 5, 3, 1):
            p = pt.RenderParams(W, H, 0, H, 0, a.spp, mrr, 1e-4, -1.0, 42)
            buf.zero_()
            st = sc.render_device(p, *ptrs, stream=stream.cuda_stream, want_stats=True)
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
            n = W * H * a.spp
            print(json.dumps({"scene": label, "mrr": mrr, "spp": a.spp, "segments_per_sample": st["segments"] / n,
                              "live_rays_per_wave_segment": st["segments"] / max(1, st["wave_segments"]),
                              "misses_per_sample": st["misses"] / n, "contributing_per_sample": st["contributing"] / n,
                              "stats_kernel_ms": st["kernel_ms"], "kernel_ms": ms[1], "msamples_per_s": n / ms[1] / 1e3,
                              "frame": hashlib.sha1(buf.cpu().numpy().tobytes()).hexdigest()[:12]}), flush=True)
        sc.close()


if __name__ == "__main__":
    main()
