#!/usr/bin/env python3
"""A/B timing of kernel variants on the GPU box: one line per (library, scene), statistics-free launches timed with
HIP events (median of 3), counters from one launch with statistics.

    python tools/ab_scenes.py [--libs hip,e32,...] [--scenes tor,x64,x195] [--spp 64] [--order-modes 0,1,2,3]

--libs names path-tracing_amd/lib/libpt_<name>.so (built with `make -C path-tracing_amd/csrc variant NAME=.. DEFS=..`);
--order-modes uses the test-hook build's `order_mode` knob (arrangement of small-scene clusters).
"""
import argparse
import importlib
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="hip")
    ap.add_argument("--scenes", default="tor,x64,x195")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--error", type=float, default=-1.0, help="adaptive-sampling threshold (-ERR); negative = off")
    ap.add_argument("--order-modes", default="")
    ap.add_argument("--sweep", default="", help="test-hook knob and values, e.g. bvh_fill=0.6,0.75,1.0 (libpt_testhooks.so)")
    a = ap.parse_args()
    import torch
    import make_replicated_scene as M
    pt = importlib.import_module("path-tracing_amd")
    tmp = tempfile.mkdtemp() + "/"
    scenes = {"tor": (os.path.join(ROOT, "models") + "/", "Tor.obj")}
    for n in (64, 195):
        if f"x{n}" in a.scenes.split(","):
            M.generate(os.path.join(ROOT, "models"), tmp, f"x{n}.obj", n)
            scenes[f"x{n}"] = (tmp, f"x{n}.obj")
    dev = torch.device("cuda", 0)
    W, H = 1920, 1080
    buf = torch.zeros(7 * W * H, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    runs = [(name, None) for name in a.libs.split(",")]
    knob = "order_mode"
    if a.order_modes:
        runs = [("testhooks", int(m)) for m in a.order_modes.split(",")]
    if a.sweep:
        knob, values = a.sweep.split("=")
        runs = [("testhooks", float(v)) for v in values.split(",")]
    baseline = {}   # scene -> frame digest of the block's first run: every variant must render the same frame
    for name, mode in runs:
        L = pt.load_library(os.path.join(ROOT, "path-tracing_amd", "lib", f"libpt_{name}.so"))
        if mode is not None:
            L.pt_test_set_mutation(b"reset", 0.0)
            L.pt_test_set_mutation(knob.encode(), float(mode))
        for sn in a.scenes.split(","):
            sc = pt.Scene.load_obj(*scenes[sn], device=0, library=L)
            p = pt.RenderParams(W, H, 0, H, 0, a.spp, 8, 1e-4, a.error, 42)
            ptrs = (buf.data_ptr(), buf.data_ptr() + 12 * W * H, buf.data_ptr() + 24 * W * H)
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
            import hashlib
            digest = hashlib.sha1(buf.cpu().numpy().tobytes()).hexdigest()[:12]   # equal digests = bit-identical frames
            ws = max(1, st["wave_segments"])
            print(f"{name}{'' if mode is None else ' ' + knob + '=' + str(mode)} {sn}: {W * H * a.spp / ms[1] / 1e3:.1f} Msamples/s  kernel {ms[1]:.2f} ms  "
                  f"node rounds/wseg {st['wave_node_rounds'] / ws:.2f}  exact rounds/wseg {st['wave_exact_iterations'] / ws:.2f}  "
                  f"exact/seg {st['exact_tests'] / max(1, st['segments']):.3f}  partial {st['partial_commit_rounds']}  frame {digest}", flush=True)
            sc.close()
            if a.sweep or a.order_modes:
                continue    # (table-builder knobs keep the frame as well, but each run is its own baseline there)
            if baseline.setdefault(sn, digest) != digest:
                # a variant that renders another frame is WRONG, not slow: stop the block here instead of timing more of it
                # (round 2's ab20 went on after a digest change and ended in a memory fault, profiles/r02_ab_logs.txt)
                print(f"DIGEST MISMATCH: {name} on {sn} renders {digest}, the block's baseline {baseline[sn]}: stopping", flush=True)
                sys.exit(3)


if __name__ == "__main__":
    main()
