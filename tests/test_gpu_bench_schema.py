"""bench.py on the GPU box: one short run, JSON contract checked field by field."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_emits_one_valid_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--spp", "8",
                        "--cpu-seconds", "2"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["unit"] == "Msamples/s" and j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32"
    assert "workload" in j["config"] and "model" not in j["config"]
    # value is consistent with the step time: W*H*spp / ms_per_step
    c = j["config"]
    assert abs(j["value"] - c["width"] * c["height"] * c["spp"] / j["ms_per_step"] / 1e3) < 1e-6 * j["value"]
    assert j["value"] > 100.0                                  # the target of BASELINE.md, with a wide margin even at 8 spp
    rf = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "valu_issue" and rf["unit"] == "Tlane-op/s" and rf["kernel_ms"] > 0
    # counters come from rocprofv3 --pmc child runs of this very command (or are null, never stale)
    assert rf["achieved"] is not None, rf["counters_source"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.2 < rf["frac"] <= 1.0
    # (the timed kernel keeps its accumulators in memory and touches them per contribution: at 8 spp that is less than the
    # algorithmic W*H*56 B of "every pixel in and out", at 256 spp several times more -- either way a vanishing part of HBM peak)
    assert rf["traffic"] is not None and rf["traffic"] > 0 and rf["hbm"]["frac"] < 0.05
    assert rf["hbm"]["algorithmic_bytes"] == c["width"] * c["height"] * 56 + c["triangles"] * 56
    assert rf["reference_equivalent_tflops"] > 0
    assert rf["kernel_ms"] <= j["ms_per_step"] * 1.001         # the HIP-event kernel time fits inside the wall-clock step
    cb = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] >= 1 and 0 < cb["value"] < j["value"]
    # the accuracy half of BASELINE.json's metric: GPU vs the CPU path on the sample both rendered
    ac = j["accuracy"]
    assert ac["accumulators_bit_identical"] is True and ac["bmp_bytes_differing"] == 0
    assert ac["rmse_rgb_float_image"] == [0.0, 0.0, 0.0] and ac["max_abs_diff_float_image"] == 0.0
    # the other two rates of SURVEY 8(d): PCIe-inclusive entry point and process start -> BMP
    assert 0 < j["pcie_inclusive"]["value"] <= j["value"] * 1.05
    assert j["end_to_end"]["value"] is not None and 0 < j["end_to_end"]["value"] < j["value"]
    assert j["end_to_end"]["phases"]["hip_startup_s"] > 0
    vr = ac["vs_reference_stream"]
    assert vr["within_tolerance"] is True and all(0.8 < z < 1.25 for z in vr["z_rms"])
    # configs[3] as written (3840 x 2160 x 256 spp), reported beside the headline value
    c3 = j["configs3_strong"]
    assert c3["scaling"] == "strong" and c3["value"] > 100.0
    # what the roofline fraction means: issued instructions, the part of their lanes that was on, the reference's own arithmetic
    assert "x 8 spp" in j["metric"]
    assert 0 < rf["useful_fraction"] < rf["frac_active_lanes"] < rf["frac"]
    assert abs(rf["frac_active_lanes"] - rf["frac"] * rf["valu_active_lane_fraction"]) < 1e-9
    # the accuracy sample ran the instantiation the timed launches run
    assert "false,false,false,false,false" in ac["gpu_instantiation"]
    # BASELINE configs[1] on fixed work, the reference-default adaptive run, configs[4] with its own counters: driver-timed
    assert j["configs1_64spp"]["value"] > 100.0 and "64 spp" in j["configs1_64spp"]["workload"]
    ad = j["adaptive_default"]
    assert 0 < ad["samples_traced"] < ad["nominal_samples"] == 1920 * 1080 * 256
    assert ad["value_traced"] < ad["value_nominal"] == ad["value"] and ad["value"] > j["configs1_64spp"]["value"]
    rep = j["configs4_replica"]
    assert rep["x64"]["triangles"] == 16398 and rep["x195"]["triangles"] == 49934
    assert rep["x64"]["value"] > rep["x195"]["value"] > 100.0
    rb = rep["x64"]["roofline"]
    assert rb["kernel"] == "pt::integrate_kernel<false,true,false,false,false,0>" and rb["achieved"] is not None, rb["counters_source"]
    assert 0 < rb["useful_fraction"] < rb["frac_active_lanes"] < rb["frac"] <= 1.0
    assert 0.3 < rb["l1_hit_rate"] < 1.0 and rb["issue"]["instructions_per_wave_segment"] > 500
    assert rep["x64"]["node_rounds_per_wave_segment"] > 1
    ra = rep["x64_adaptive_default"]      # the reference's default -ERR 0.001 on the same scene: fewer samples traced, more nominal samples per second
    assert ra["samples_traced"] < 0.9 * 1920 * 1080 * 256 and ra["value"] > rep["x64"]["value"]
    # BASELINE configs[2] (1024 spp) with the HBM bytes of that launch, and the open scene under a sky (path regeneration)
    c2 = j["configs2_1024spp"]
    assert "1024 spp" in c2["workload"] and c2["value"] > 100.0 and c2["steps"] == 3
    assert c2["hbm"]["traffic_bytes"] is not None and c2["hbm"]["traffic_bytes"] > 0, c2["hbm"]["counters_source"]
    assert c2["hbm"]["algorithmic_bytes"] == 1920 * 1080 * 56 + 270 * 56 and 0 < c2["hbm"]["frac"] < 0.05
    sk = j["skybox_open"]
    assert sk["value"] > 100.0 and sk["accumulators_bit_identical_to_oracle"] is True
    assert 56 < sk["live_rays_per_wave_segment"] <= 64 and 1.0 < sk["segments_per_sample"] < 5.0 and sk["misses_per_sample"] > 0.5
    assert sk["roofline"]["achieved"] is not None, sk["roofline"]["counters_source"]
    assert sk["roofline"]["kernel"] == "pt::integrate_kernel<true,false,false,false,false,0>"
    # the C++ host alone on the same frame (pt_render -GPUS 1 -BENCH_STEPS): the same rate (a 2.8 ms frame at 8 spp: loose here)
    cx = j["cxx_frame"]["weak"]
    assert cx["cxx_frame"] and cx["bands"] == 1 and cx["transport"] == "none" and abs(cx["over_torch_leg"] - 1) < 0.15
    # end to end: the phases account for the wall time
    ee = j["end_to_end"]
    assert abs(ee["seconds"] - ee["phases"]["pre_main_s"] - ee["phases"]["main_s"] - ee["exit_and_wait_s"]) < 1e-6


def test_two_rank_rehearsal_frame_equals_single_process(tmp_path):
    """bench.py's N > 1 code path (row bands, packed accumulators, the asynchronous gather pipelined over two band
    buffers, assembly, resolve) with two ranks sharing this box's one GPU and gloo as the transport: the gathered
    1920x2160 frame must be byte-identical to the stand-alone front end's render of the same frame."""
    bmp = str(tmp_path / "two_ranks.bmp")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--spp", "8",
                        "--rehearse-on-one-gpu", "--no-configs3", "--write-bmp", bmp], capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["config"]["height"] == 2160 and "REHEARSAL" in j["data"]
    ref = str(tmp_path / "one_process.bmp")
    exe = os.path.join(ROOT, "path-tracing_amd", "bin", "pt_render")
    c = subprocess.run([exe, "--W", "1920", "--H", "2160", "-RPP", "8", "-MRR", "8", "-ERR", "-1", "-UPDATE", "0", "-QUIET", "1", "-SEED", "42",
                        "-MODEL_PATH", os.path.join(ROOT, "models") + "/", "-OUT", ref], capture_output=True, text=True, cwd=tmp_path)
    assert c.returncode == 0, c.stderr
    assert open(bmp, "rb").read() == open(ref, "rb").read()


def test_self_launch_two_ranks_and_too_many_gpus(tmp_path):
    """`python bench.py --gpus N` without a launcher starts its own rank processes (here: two ranks rehearsing on this
    box's one GPU) and refuses, with a clear message and nothing run, when N exceeds the visible devices."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", "4",
                        "--rehearse-on-one-gpu", "--no-configs3"], capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["rccl_ranks_seen"] == 2 and j["config"]["parallelism"] == "tilerows_interleaved2"
    cx = j["cxx_frame"]["weak"]       # the C++ host's two-band frame next to the two-rank torch leg (rehearsed: device copies)
    assert cx["bands"] == 2 and cx["transport"] == "device_copies" and cx["value"] > 100.0
    # the diagnosis a first multi-device run should leave behind: every band's / rank's kernel time and the gather alone
    assert len(cx["band_kernel_ms"]) == 2 and min(cx["band_kernel_ms"]) > 0 and cx["gather_alone_ms"] > 0
    dg = j["multi_gpu_diagnosis"][0]
    assert len(dg["kernel_ms_per_rank"]) == 2 and min(dg["kernel_ms_per_rank"]) > 0 and dg["gather_alone_ms"] > 0
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True, cwd=ROOT,
                       timeout=300, env=env)
    assert r.returncode != 0 and "HIP device(s) are visible" in r.stderr and not r.stdout.strip()


def test_two_rank_rccl_frame_equals_single_process(tmp_path):
    """The real N = 2 path: two processes, one GPU each, backend "nccl" (RCCL), device-tensor gather overlapping the next
    frame's kernel.  Needs two GPUs, so it is skipped on the one-GPU test box; the gathered frame must be byte-identical
    to the stand-alone front end's."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices: the RCCL branch cannot run on a one-GPU box")
    bmp = str(tmp_path / "two_ranks.bmp")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--spp", "8",
                        "--write-bmp", bmp], capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["rccl_ranks_seen"] == 2 and "REHEARSAL" not in j["data"]
    assert j["configs3_strong"]["value"] > 100.0
    ref = str(tmp_path / "one_process.bmp")
    exe = os.path.join(ROOT, "path-tracing_amd", "bin", "pt_render")
    c = subprocess.run([exe, "--W", "1920", "--H", "2160", "-RPP", "8", "-MRR", "8", "-ERR", "-1", "-UPDATE", "0", "-QUIET", "1", "-SEED", "42",
                        "-MODEL_PATH", os.path.join(ROOT, "models") + "/", "-OUT", ref], capture_output=True, text=True, cwd=tmp_path)
    assert c.returncode == 0, c.stderr
    assert open(bmp, "rb").read() == open(ref, "rb").read()
