#!/usr/bin/env python3
"""Soak: render the same frame many times (different dispatch timing every time) and require identical accumulators.
    python tools/soak.py [--frames 40] [--spp 64]"""
import argparse, hashlib, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=40); ap.add_argument("--spp", type=int, default=64)
a = ap.parse_args()
import torch
pt = importlib.import_module("path-tracing_amd")
sc = pt.Scene.load_obj(os.path.join(ROOT, "models") + "/", "Tor.obj", device=0)
W, H = 1920, 1080
n = W * H
dev = torch.device("cuda", 0)
buf = torch.zeros(7 * n, dtype=torch.float32, device=dev)
p = pt.RenderParams(W, H, 0, H, 0, a.spp, 8, 1e-4, 0.001, 42)
digests = set()
last = None
for f in range(a.frames):
    buf.zero_()
    st = sc.render_device(p, buf.data_ptr(), buf.data_ptr() + 12 * n, buf.data_ptr() + 24 * n,
                          stream=torch.cuda.current_stream(dev).cuda_stream, want_stats=(f % 2 == 0))   # both instantiations
    st = st or last
    last = st
    digests.add(hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest())
print(f"{a.frames} frames of {W}x{H}x{a.spp} (adaptive on): {len(digests)} distinct digest(s); last kernel {st['kernel_ms']:.2f} ms")
sys.exit(0 if len(digests) == 1 else 1)
