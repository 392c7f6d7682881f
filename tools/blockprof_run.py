#!/usr/bin/env python3
"""Renders one frame through libpt_blockprof.so (instrumented code object) and prints wave-segments; the library writes
the execution counters to $PT_BLOCKPROF_OUT.<kernel>.txt.   python tools/blockprof_run.py tor|open|x64|x195 [spp]
(open = Tor.obj without its back wall under a sky bitmap: the skybox instantiation with path regeneration)"""
import importlib
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
which = sys.argv[1] if len(sys.argv) > 1 else "tor"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
os.environ.setdefault("PT_BLOCKPROF_HSACO", os.path.join(ROOT, "path-tracing_amd", "lib", "blockprof", "pt_bp.hsaco"))
pt = importlib.import_module("path-tracing_amd")
sky = None
if which == "tor":
    d, n = os.path.join(ROOT, "models") + "/", "Tor.obj"
elif which == "open":
    import make_open_scene as MO
    d, n = tempfile.mkdtemp() + "/", "TorOpen.obj"
    MO.generate(os.path.join(ROOT, "models"), d)
    sky = d + "sky.bmp"
else:
    import make_replicated_scene as M
    d, n = tempfile.mkdtemp() + "/", which + ".obj"
    M.generate(os.path.join(ROOT, "models"), d, n, int(which[1:]))
# the statistics of the same frame from the product library (the instrumented kernels are the statistics-free ones)
s0 = pt.Scene.load_obj(d, n, 0)
if sky:
    s0.set_skybox(sky)
st = s0.render_host(1920, 1080, spp, 8)[3]
s0.close()
L = pt.load_library(os.path.join(ROOT, "path-tracing_amd", "lib", "libpt_blockprof.so"))
s = pt.Scene.load_obj(d, n, 0, library=L)
if sky:
    s.set_skybox(sky)
s.render_host(1920, 1080, spp, 8, want_stats=False)
print(which, "wave_segments", st["wave_segments"], "segments", st["segments"])
