#!/usr/bin/env python3
"""Runs the post filters (-GAUSS / -MEDIAN, main.cpp:187-192) on a 1920x1080 tonemapped image; meant to be run under
`rocprofv3 --kernel-trace --stats`, whose per-kernel durations are the measurement (the C entry point stages the
image through host memory, so wall-clock here is PCIe-inclusive).

    python tools/bench_filters.py [--gauss 2] [--median 1,2,3] [--repeat 3]
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gauss", default="1,2,4")
    ap.add_argument("--median", default="1,2,3")
    ap.add_argument("--repeat", type=int, default=3)
    a = ap.parse_args()
    pt = importlib.import_module("path-tracing_amd")
    W, H = 1920, 1080
    rng = np.random.default_rng(1)
    img = (rng.random((H, W, 3), dtype=np.float32) * 255.0).astype(np.float32)
    for kind, sizes in (("gauss", a.gauss), ("median", a.median)):
        for r in [int(t) for t in sizes.split(",") if t]:
            best = 1e9
            for _ in range(a.repeat):
                t = time.perf_counter()
                pt.post_filter(img, gauss=r if kind == "gauss" else 0, median=r if kind == "median" else 0)
                best = min(best, time.perf_counter() - t)
            print(json.dumps({"filter": kind, "size": r, "width": W, "height": H, "host_call_ms": round(best * 1e3, 2)}), flush=True)


if __name__ == "__main__":
    main()
