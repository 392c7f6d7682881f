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
    assert rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["kernel_ms"] > 0
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


def test_two_rank_rehearsal_frame_equals_single_process(tmp_path):
    """bench.py's N > 1 code path (row bands, packed accumulators, the asynchronous gather pipelined over two band
    buffers, assembly, resolve) with two ranks sharing this box's one GPU and gloo as the transport: the gathered
    1920x2160 frame must be byte-identical to the stand-alone front end's render of the same frame."""
    bmp = str(tmp_path / "two_ranks.bmp")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--spp", "8",
                        "--rehearse-on-one-gpu", "--write-bmp", bmp], capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["config"]["height"] == 2160 and "REHEARSAL" in j["data"]
    ref = str(tmp_path / "one_process.bmp")
    exe = os.path.join(ROOT, "path-tracing_amd", "bin", "pt_render")
    c = subprocess.run([exe, "--W", "1920", "--H", "2160", "-RPP", "8", "-MRR", "8", "-ERR", "-1", "-UPDATE", "0", "-QUIET", "1", "-SEED", "42",
                        "-MODEL_PATH", os.path.join(ROOT, "models") + "/", "-OUT", ref], capture_output=True, text=True, cwd=tmp_path)
    assert c.returncode == 0, c.stderr
    assert open(bmp, "rb").read() == open(ref, "rb").read()
