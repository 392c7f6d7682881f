#!/usr/bin/env python3
"""Quick probe of the replicated scene (BASELINE config 5 geometry, 8 spp): rate and rounds per wave-segment."""
import importlib
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_replicated_scene as M  # noqa: E402

pt = importlib.import_module("path-tracing_amd")
tmp = tempfile.mkdtemp() + "/"
for n in [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "64").split(",")]:
    M.generate(os.path.join(ROOT, "models"), tmp, f"x{n}.obj", n)
    s = pt.Scene.load_obj(tmp, f"x{n}.obj", 0)
    st = s.render_host(1920, 1080, 8, 8)[3]
    ws = st["wave_segments"]
    print(n, round(st["kernel_ms"], 2), "ms", round(1920 * 1080 * 8 / st["kernel_ms"] / 1e3, 1), "Ms/s; node rounds/wseg",
          round(st["wave_node_rounds"] / ws, 2), "exact rounds/wseg", round(st["wave_exact_iterations"] / ws, 2),
          "exact/seg", round(st["exact_tests"] / st["segments"], 3))
