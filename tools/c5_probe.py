import importlib, os, sys, tempfile
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import make_replicated_scene as M
pt = importlib.import_module("path-tracing_amd")
tmp = tempfile.mkdtemp() + "/"
for n in (64,):
    M.generate("models", tmp, f"x{n}.obj", n)
    s = pt.Scene.load_obj(tmp, f"x{n}.obj", 0)
    r = s.render_host(1920, 1080, 8, 8)
    st = r[3]; ws = st["wave_segments"]
    print(n, st["kernel_ms"], "Ms/s", 1920*1080*8/st["kernel_ms"]/1e3, "node rounds/wseg", st["wave_node_rounds"]/ws, "exact rounds/wseg", st["wave_exact_iterations"]/ws, "exact/seg", st["exact_tests"]/st["segments"])
