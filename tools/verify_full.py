#!/usr/bin/env python3
"""Every segment of the full-size frames against the reference's all-triangles loop, beyond what the test suite runs:

    python tools/verify_full.py [shipped] [lib=<name>] [only=<part of a frame's name>]   ->  one line per frame (segments checked, mismatches, seconds)

Without an argument: libpt_verify.so (the statistics instantiations, full search on every segment).  `shipped`:
libpt_verify_shipped.so -- the statistics-free instantiations a caller without pt_render_stats gets (two pixels per lane,
emitter-first last segments; big scenes: the can-reach filter), where a filtered last segment may report a miss only if the
reference's hit has no emissive lobe (tests/test_gpu_verify_shipped.py)."""
import importlib
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_replicated_scene as M

pt = importlib.import_module("path-tracing_amd")
shipped = len(sys.argv) > 1 and sys.argv[1] == "shipped"
named = [a[4:] for a in sys.argv[1:] if a.startswith("lib=")]      # lib=<name>: path-tracing_amd/lib/libpt_<name>.so (make verify-variant NAME=...)
L = pt.load_library(os.path.join(ROOT, "path-tracing_amd", "lib", f"libpt_{named[0]}.so") if named else
                    os.path.join(ROOT, "path-tracing_amd", "lib", "libpt_verify_shipped.so") if shipped else pt.VERIFY_LIB_PATH)
L.pt_test_set_mutation(b"reset", 0.0)
models = os.path.join(ROOT, "models") + "/"
jobs = [("configs[2] Tor.obj 1920x1080x1024spp", models, "Tor.obj", 1920, 1080, 1024, -1.0),
        ("configs[3] Tor.obj 3840x2160x256spp", models, "Tor.obj", 3840, 2160, 256, -1.0)]
if shipped:   # the reference's default -ERR 0.001: the two-pixel kernel's adaptive instantiations (batches of the tile's pixels, 32 x 8 tiles at this size)
    jobs.append(("Tor.obj 1920x1080x512spp -ERR 0.001", models, "Tor.obj", 1920, 1080, 512, 0.001))
d = tempfile.mkdtemp() + "/"
for n, spp in ((64, 32), (195, 16)):
    M.generate(os.path.join(ROOT, "models"), d, f"x{n}.obj", n)
    jobs.append((f"configs[4] x{n} replica 1920x1080x{spp}spp", d, f"x{n}.obj", 1920, 1080, spp, -1.0))
    if shipped and n == 64:   # the box-tree kernel's batches of adaptive sampling (16 x 8 tiles at this size)
        jobs.append((f"x{n} replica 1920x1080x48spp -ERR 0.001", d, f"x{n}.obj", 1920, 1080, 48, 0.001))
# an open scene under a sky: the skybox instantiations (path regeneration; the torus x9 likewise for the box tree)
import make_open_scene as MO
MO.generate(os.path.join(ROOT, "models"), d)
M.generate(os.path.join(ROOT, "models"), d, "x9.obj", 9)
MO.generate(os.path.join(ROOT, "models"), d, name="X9Open.obj", source="x9.obj", source_dir=d)
jobs.append(("open Tor.obj + skybox 1920x1080x128spp", d, "TorOpen.obj", 1920, 1080, 128, -1.0))
jobs.append(("open x9 + skybox 1920x1080x64spp -ERR 0.001", d, "X9Open.obj", 1920, 1080, 64, 0.001))
only = [a[5:] for a in sys.argv[1:] if a.startswith("only=")]
for name, dd, obj, W, H, spp, err in jobs:
    if only and not any(o in name for o in only):
        continue
    s = pt.Scene.load_obj(dd, obj, device=0, library=L)
    if "skybox" in name:
        s.set_skybox(d + "sky.bmp")
    t = time.perf_counter()
    st = s.render_host(W, H, spp, 8, error=err)[3]
    extra = f", {st['partial_commit_rounds']} batches in which a pixel ran on another lane" if shipped and err >= 0 else ""
    print(f"{name}: {st['verify_checked']} segments checked against all {st['n_triangles']} triangles, "
          f"{st['verify_mismatches']} mismatches{extra}, {time.perf_counter() - t:.1f} s", flush=True)
    s.close()
