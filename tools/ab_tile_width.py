#!/usr/bin/env python3
"""8 x 8 against 16 x 8 tiles of the statistics-free small-scene kernel over frame sizes (test-hook knob `tile_width` of
libpt_testhooks.so, same kernels as the product): where integrator_plan_tiles should switch.  profiles/r03_ab_logs.txt ab53.

    python tools/ab_tile_width.py
"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("path-tracing_amd")
L = pt.load_library(pt.TESTHOOKS_LIB_PATH)
sc = pt.Scene.load_obj(os.path.join(ROOT, "models") + "/", "Tor.obj", device=0, library=L)
for (W, H, rows) in [(256, 256, None), (640, 360, None), (960, 540, None), (1280, 720, None), (1920, 1080, None), (3840, 2160, (0, 270)), (3840, 2160, (0, 540)), (1920, 1080, (0, 135))]:
    for spp in (64, 256):
        res = []
        for mode in (1.0, 2.0):
            L.pt_test_set_mutation(b"tile_width", mode)
            ses = pt.Session(sc, W, H, rows=rows)
            ses.render(0, spp, 8)
            ses.read()
            ts = []
            for _ in range(3):
                ses.clear()
                t = time.perf_counter(); ses.render(0, spp, 8); L.pt_session_wait(ses._h); ts.append(time.perf_counter() - t)
            ses.close()
            res.append(min(ts) * 1e3)
        r = rows or (0, H)
        tiles16 = ((W + 15) // 16) * ((r[1] - r[0] + 7) // 8)
        print(f"{W}x{r[1]-r[0]} x{spp}: 8x8 {res[0]:.3f} ms  16x8 {res[1]:.3f} ms  ratio {res[0]/res[1]:.3f}  wide tiles {tiles16}", flush=True)
L.pt_test_set_mutation(b"reset", 0.0)
