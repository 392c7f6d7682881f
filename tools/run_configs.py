#!/usr/bin/env python3
"""Runs the BASELINE.json configurations on one GPU and prints one JSON line per configuration.

    python tools/run_configs.py [--spp-scale 1.0] [--only 2,5]

Config 1 is the reference's CPU-sized case (256x256x4, MRR 3); 2 and 3 are 1080p at 64 / 1024 spp; 4 is the 4K frame
(rendered here on ONE GPU; bench.py --gpus N splits it); 5 is the replicated scene (x64 and x195 torus instances).
Every configuration is run with -ERR -1 (all samples traced) and, for 2, also with the reference's default -ERR 0.001.
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
    ap.add_argument("--spp-scale", type=float, default=1.0)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    import numpy as np
    import torch
    import make_replicated_scene as M
    pt = importlib.import_module("path-tracing_amd")
    models = os.path.join(ROOT, "models") + "/"
    tmp = tempfile.mkdtemp() + "/"
    scenes = {"Tor.obj": (models, "Tor.obj")}
    for n in (64, 195):
        M.generate(os.path.join(ROOT, "models"), tmp, f"TorX{n}.obj", n)
        scenes[f"TorX{n}.obj"] = (tmp, f"TorX{n}.obj")
    import make_open_scene as MO
    MO.generate(os.path.join(ROOT, "models"), tmp)             # Tor.obj without its back wall + sky.bmp (row N1: path regeneration)
    scenes["TorOpen.obj"] = (tmp, "TorOpen.obj")
    import shuffle_obj
    shuffle_obj.shuffle(os.path.join(ROOT, "models", "Tor.obj"), tmp + "TorShuffled.obj")
    scenes["TorShuffled.obj"] = (tmp, "TorShuffled.obj")
    shuffle_obj.shuffle(tmp + "TorX64.obj", tmp + "TorX64Shuffled.obj")
    scenes["TorX64Shuffled.obj"] = (tmp, "TorX64Shuffled.obj")
    configs = [
        ("1", "Tor.obj", 256, 256, 4, 3, -1.0),
        ("2-256spp", "Tor.obj", 1920, 1080, 256, 8, -1.0),
        ("2-256spp-shuffled-faces", "TorShuffled.obj", 1920, 1080, 256, 8, -1.0),
        ("5-x64-shuffled-faces", "TorX64Shuffled.obj", 1920, 1080, 256, 8, -1.0),
        ("2", "Tor.obj", 1920, 1080, 64, 8, -1.0),
        ("2-adaptive", "Tor.obj", 1920, 1080, 64, 8, 0.001),
        ("3", "Tor.obj", 1920, 1080, 1024, 8, -1.0),
        ("4-one-gpu", "Tor.obj", 3840, 2160, 256, 8, -1.0),
        ("open-sky", "TorOpen.obj", 1920, 1080, 256, 8, -1.0),
        ("open-sky-mrr3", "TorOpen.obj", 1920, 1080, 256, 3, -1.0),
        ("5-x64", "TorX64.obj", 1920, 1080, 256, 8, -1.0),
        ("5-x195", "TorX195.obj", 1920, 1080, 256, 8, -1.0),
    ]
    only = set(a.only.split(",")) if a.only else None
    dev = torch.device("cuda", 0)
    for name, scene_name, W, H, spp, mrr, err in configs:
        if only and name.split("-")[0] not in only and name not in only:
            continue
        spp = max(1, int(round(spp * a.spp_scale)))
        sc = pt.Scene.load_obj(*scenes[scene_name], device=0)
        if scene_name == "TorOpen.obj":
            sc.set_skybox(tmp + "sky.bmp")
        n = W * H
        buf = torch.zeros(7 * n, dtype=torch.float32, device=dev)
        p = pt.RenderParams(W, H, 0, H, 0, spp, mrr, 1e-4, err, 42)
        st = sc.render_device(p, buf.data_ptr(), buf.data_ptr() + 12 * n, buf.data_ptr() + 24 * n,
                              stream=torch.cuda.current_stream(dev).cuda_stream, want_stats=True)
        # the same frame again as a caller that does not ask for statistics would run it (the instantiation without
        # counters), timed with HIP events on the launch stream -- this is the rate; the counters come from the launch above
        buf.zero_()
        stream = torch.cuda.current_stream(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        sc.render_device(p, buf.data_ptr(), buf.data_ptr() + 12 * n, buf.data_ptr() + 24 * n, stream=stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1)
        out = {"config": name, "scene": scene_name, "triangles": st["n_triangles"], "width": W, "height": H, "spp": spp,
               "mrr": mrr, "error": err, "kernel_ms": round(ms, 3), "kernel_ms_with_statistics": round(st["kernel_ms"], 3),
               "nominal_Msamples_per_s": round(W * H * spp / ms / 1e3, 1),
               "traced_samples": st["samples_traced"], "traced_Msamples_per_s": round(st["samples_traced"] / ms / 1e3, 1),
               "segments_per_sample": round(st["segments"] / max(1, st["samples_traced"]), 3),
               "exact_tests_per_segment": round(st["exact_tests"] / max(1, st["segments"]), 3),
               "node_rounds_per_wave_segment": round(st["wave_node_rounds"] / max(1, st["wave_segments"]), 2),
               "exact_rounds_per_wave_segment": round(st["wave_exact_iterations"] / max(1, st["wave_segments"]), 2),
               "partial_commit_rounds": st["partial_commit_rounds"],
               "live_rays_per_wave_segment": round(st["segments"] / max(1, st["wave_segments"]), 2),
               "contributing_fraction": round(st["contributing"] / max(1, st["samples_traced"]), 5)}
        print(json.dumps(out), flush=True)
        sc.close()


if __name__ == "__main__":
    main()
