"""The host-side scene code (loader with the reference's token semantics, device tables, culling hierarchy incl. quad
fusion and the big-scene tables) built with g++ -fsanitize=address,undefined and run on Tor.obj, on random scenes
with degenerate triangles and on malformed OBJ files.  GPU sanitizers are not available on the pool; this is the CPU
half (the kernels' indexing is covered by the parity and fuzz suites)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_scene_code_is_clean_under_asan_and_ubsan(tmp_path, models_dir):
    exe = str(tmp_path / "scene_san")
    csrc = os.path.join(ROOT, "path-tracing_amd", "csrc")
    build = subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            "-fno-omit-frame-pointer", "-I", csrc, os.path.join(ROOT, "tests", "native", "scene_sanitizer_main.cpp"),
                            os.path.join(csrc, "pt_scene.cpp"), "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe, models_dir, str(tmp_path) + "/"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr
    out = run.stdout
    assert "eps 0.0001: 2 clusters" in out and "n 3000:" in out and " bvh" in out
    import re
    m = re.search(r"n 3000: \d+ clusters \d+ spheres \d+ bary (\d+) bary_all (\d+) slots (\d+) bvh", out)
    assert m and int(m.group(1)) == int(m.group(2)) + 4 and int(m.group(2)) >= 3000 and int(m.group(3)) > 300
    assert out.count("bad obj -> 0") == 4 and "bad obj -> 1" in out        # four rejected with a message, the empty file loads (no triangles)


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"), reason="g++ or the HIP headers are not available")
def test_c_api_host_side_is_clean_under_tsan(tmp_path, models_dir):
    """pt_resolve runs bands of rows on several host threads and per-device copies of a scene share one hierarchy cache:
    the C API's host code (pt_capi.cpp, pt_frame.cpp, pt_scene.cpp; kernel launchers stubbed) under ThreadSanitizer, on the CPU."""
    exe = str(tmp_path / "capi_tsan")
    csrc = os.path.join(ROOT, "path-tracing_amd", "csrc")
    build = subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=thread", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__",
                            "-I", csrc, "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                            os.path.join(ROOT, "tests", "native", "resolve_tsan_main.cpp"), os.path.join(csrc, "pt_capi.cpp"),
                            os.path.join(csrc, "pt_frame.cpp"), os.path.join(csrc, "pt_scene.cpp"), "-o", exe,
                            "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe, models_dir], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-3000:])
    assert "WARNING: ThreadSanitizer" not in run.stderr, run.stderr[-3000:]
    assert "resolve ok" in run.stdout and "hierarchy threads bad 0" in run.stdout


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"), reason="g++ or the HIP headers are not available")
def test_c_api_error_paths_are_clean_under_asan(tmp_path, models_dir):
    """Every loader's failure path destroys the half-built scene inside the call (no device, ordinal out of range): nothing may
    touch it afterwards (round 3's pt_scene_load_obj wrote its timing into the freed host side).  The C API's host code under
    AddressSanitizer + UBSan, kernel launchers stubbed; runs with or without a GPU (the bad ordinal is 9999)."""
    import numpy as np
    import oracle_lib as O
    exe = str(tmp_path / "capi_asan")
    csrc = os.path.join(ROOT, "path-tracing_amd", "csrc")
    build = subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                            "-D__HIP_PLATFORM_AMD__", "-I", csrc, "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                            os.path.join(ROOT, "tests", "native", "capi_asan_main.cpp"), os.path.join(csrc, "pt_capi.cpp"),
                            os.path.join(csrc, "pt_frame.cpp"), os.path.join(csrc, "pt_scene.cpp"), "-o", exe,
                            "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    sky = str(tmp_path / "sky.bmp")
    O.write_bmp(sky, np.random.default_rng(2).integers(0, 256, (6, 9, 3)).astype(np.uint8))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")      # (the HIP runtime's own start-up allocations are not ours to judge)
    run = subprocess.run([exe, models_dir, sky], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-3000:]
    assert "capi error paths ok" in run.stdout
