#!/usr/bin/env python3
"""Quick probe of Tor.obj at 1080p x 16 spp: rounds and pairs per wave-segment (the numbers DESIGN.md quotes)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("path-tracing_amd")
s = pt.Scene.load_obj(os.path.join(ROOT, "models") + "/", "Tor.obj", 0)
st = s.render_host(1920, 1080, 16, 8)[3]
ws = st["wave_segments"]
print(f"kernel {st['kernel_ms']:.2f} ms (with statistics); per wave-segment: {st['segments'] / ws:.1f} live rays, "
      f"{st['wave_node_rounds'] / ws:.2f} tree rounds, {st['wave_exact_iterations'] / ws:.2f} exact rounds of "
      f"{st['exact_tests'] / st['wave_exact_iterations']:.1f} pairs; {st['exact_tests'] / st['segments']:.3f} exact tests per segment; "
      f"{st['partial_commit_rounds']} partial commits")
