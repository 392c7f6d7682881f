#!/usr/bin/env python3
"""Benchmark of the radiance-integrator hot path (BASELINE.json metric: Msamples/s = W*H*spp / wall on Tor.obj).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Without WORLD_SIZE in the environment `--gpus N` starts its own N rank processes (one per GPU, RCCL) and prints rank
0's line; under torch.distributed.run it is one of the ranks.

One "step" = one complete frame: zero the accumulators, trace every sample of every pixel of this rank's band of rows
(all passes x segments x triangles in one kernel launch through the C ABI), and -- for N > 1 -- the single RCCL
gather of the accumulator bands to rank 0 (issued asynchronously: it overlaps the next frame's kernel, every gather is
complete before the timed region ends).  At N = 1 the frame is the configuration north_star quotes its target on:
models/Tor.obj, 1920x1080, 256 spp, -MRR 8, adaptive sampling off (-ERR -1, so all W*H*spp samples are traced;
`--spp 64` is BASELINE configs[1], `--spp 1024` configs[2]).  For N > 1 the image grows with N
(path-tracing_amd/bands.py: frame_for) so that every GPU keeps a 1080p-sized share: weak scaling, `value`.  Rank k of N renders
every N-th tile row of 8 image rows (an interleaved split, pt_render_params::row_stride: contiguous bands cost unequal amounts,
bands.py).  Next to it every run also times BASELINE configs[3] as written -- 3840x2160 x 256 spp split over the N ranks, one
gather -- and reports it as `configs3_strong` (strong scaling: the frame is fixed, 2160/N rows per GPU).

Rank 0 prints ONE JSON line.  Besides the contract's fields it carries
  roofline      the integrator kernel against the roofline that bounds it, vector-ALU issue (the path is
                neither HBM- nor MFMA-bound): achieved = VALU lane-operations the kernel EXECUTES per second
                (SQ_INSTS_VALU x 64, counted by rocprofv3 --pmc child runs of the C++ front end on the same frame) over
                the HIP-event kernel time measured live on the launch stream; peak = 78.6 T lane-op/s (1024 SIMDs x 32
                lanes x 2.4 GHz, no FMA: parity forbids contraction).  frac counts issued vector instructions whatever their
                lane mask; frac_active_lanes = frac x the fraction of lanes that were on; useful_fraction = the part of the
                executed lane-operations that is the reference's own arithmetic (one exact test + shading per segment).
                traffic = HBM bytes per launch from FETCH_SIZE and WRITE_SIZE (separate passes).
                reference_equivalent_tflops = the REFERENCE's brute-force work (segments x triangles x 31.5 flop) per
                second -- what the culling hierarchy saves, not a roofline.
  cpu_baseline  the CPU oracle (a port of the reference's algorithm) timed on this box's host cores on a bounded sample
  accuracy      the other half of BASELINE's metric: per-channel RMSE against the CPU path
  cxx_frame     the same frame(s) driven by the C++ host alone (pt_render -GPUS N: pt_frame_*, direct RCCL), as a child process
  configs1_64spp / adaptive_default / configs4_replica   (N = 1) BASELINE configs[1], the reference-default -ERR 0.001 run
                with its traced-sample count, and configs[4] (the x64 / x195 replicated scenes) with its own counters
  configs2_1024spp   (N = 1) BASELINE configs[2], the "rocprof HBM GB/s run": Tor.obj 1080p x 1024 spp, with FETCH_SIZE / WRITE_SIZE
                of that launch from live --pmc passes at 1024 spp
  skybox_open   (N = 1) an OPEN scene under a sky bitmap (Tor.obj without its back wall, tools/make_open_scene.py): the skybox
                instantiation with path regeneration; rate, live rays per wave-segment, segments per sample, bit identity with
                the oracle on a band
  multi_gpu_diagnosis   (N > 1) every rank's own kernel time and one gather timed alone after all kernels are done
"""
import argparse
import datetime
import csv
import glob
import hashlib
import importlib
import json
import os
import re
import shutil
import signal
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

FLOP_PER_TEST = 31.5            # reference's own average over its stage-exit mix
FLOP_FULL_TEST = 79             # stages A-D of Triangle::Intersect
FLOP_SHADING = 100              # Material::Process + lobe + Ray::Reflect, per segment
PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz
PEAK_VALU_TLANEOPS = 78.6432    # the same without FMA: one operation per lane per clock
PEAK_HBM_GBS = 8000.0
BASE_W, BASE_H, SPP, MRR = 1920, 1080, 256, 8
C3_W, C3_H, C3_SPP = 3840, 2160, 256      # BASELINE configs[3]
C2_SPP = 1024                             # BASELINE configs[2]
KERNEL_SOURCES = ["pt_kernels.hip", "pt_kernels.hpp", "pt_fastfp.hpp", "pt_scene.cpp", "pt_scene.hpp", "pt_capi.cpp"]
EXE = os.path.join(ROOT, "path-tracing_amd", "bin", "pt_render")
MODELS = os.path.join(ROOT, "models") + "/"


def host_cores():
    """CPU threads this process may really use: min(affinity, cgroup quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def instantiation(kernel_name):
    """integrate_kernel<SKY, BIG, STATS, ENV, NARROW, ADAPT> -> (sky, big, stats, env, narrow, adapt) as booleans (ADAPT is an
    int: 0 = off), or None."""
    m = re.search(r"integrate_kernel<(\w+),(\w+),(\w+),(\w+)(?:,(\w+))?(?:,(\w+))?>", kernel_name.replace(" ", ""))
    return tuple(x not in (None, "false", "0") for x in m.groups()) if m else None      # (missing trailing arguments read as False)


def kernel_source_sha():
    """Identifies the kernel a PMC summary was taken from: sha256 over the sources the integrator is built from."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "path-tracing_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def visible_devices():
    """HIP devices this process may use, WITHOUT touching the HIP runtime (torch.cuda.device_count() would bring the
    runtime up in this launcher process): the *_VISIBLE_DEVICES lists if set, else the GPU nodes of the KFD topology."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    for p in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(line.split(None, 1) for line in open(p).read().splitlines() if " " in line)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except (OSError, ValueError):
            pass
    if n == 0:      # topology not readable here: ask the runtime, but in a short-lived child, never in this process
        try:
            r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=180)
            n = int(r.stdout.strip().splitlines()[-1])
        except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
            n = 0
    return n


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves.  This parent never touches the GPU and only ever spawns children.
# ---------------------------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    have = visible_devices()
    if args.gpus > have and not args.rehearse_on_one_gpu:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} HIP device(s) are visible; nothing was run "
                         f"(use --rehearse-on-one-gpu to exercise the {args.gpus}-rank code path on one device)\n")
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def run_group(cmd, timeout, **kw):
    """subprocess.run in a process group of its own; on timeout the WHOLE group is killed and reaped (rocprofv3's child --
    the program that holds the GPU -- must not survive its parent and keep the GPU busy under the timed frames)."""
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True, **kw)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        p.communicate()
        raise
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


# ---------------------------------------------------------------------------------------------------------------------
# Hardware counters of the timed kernels: rocprofv3 --pmc over the C++ front end rendering the same frame through the same
# C ABI (pt_render -BENCH_STEPS 1: the statistics-free instantiation, like the timed launches), in child processes, before
# this process touches the GPU.  One counter group per pass, never combined with a trace domain (MI355X_MICROARCH.md
# "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one pass).
# ---------------------------------------------------------------------------------------------------------------------
PMC_SQ = ["SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_SMEM",
          "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE"]
PMC_PASSES_TOR = [PMC_SQ, ["FETCH_SIZE"], ["WRITE_SIZE"]]
PMC_PASSES_BIG = [PMC_SQ, ["TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"]]


def pmc_live(passes, model_dir, model_name, spp, big, deadline, sky=None):
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    if not os.path.exists(EXE):
        return None, "path-tracing_amd/bin/pt_render is not built"
    counters, t0 = {}, time.perf_counter()
    td = tempfile.mkdtemp(prefix="pt_pmc_")
    try:
        for i, group in enumerate(passes):
            left = deadline - time.perf_counter()
            if left < 10:
                return None, "time budget of the counter passes exhausted"
            d = os.path.join(td, f"p{i}")
            cmd = [exe, "--pmc", *group, "-d", d, "-o", "p", "--output-format", "csv", "--", EXE, "--W", str(BASE_W), "--H", str(BASE_H),
                   "-RPP", str(spp), "-MRR", str(MRR), "-ERR", "-1", "-SEED", "42", "-MODEL_PATH", model_dir, "-MODEL_NAME", model_name,
                   "-BENCH_STEPS", "1", "-BENCH_WARMUP", "1"]
            if sky:
                cmd += ["-SKYBOX", sky]
            r = run_group(cmd, left, cwd=td, env=dict(os.environ, TMPDIR=td))
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 pass {group} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}"
            rows = []
            for x in csv.DictReader(open(max(files, key=os.path.getmtime))):
                inst = instantiation(x["Kernel_Name"])
                if inst and not inst[2] and inst[1] == big and inst[0] == bool(sky):      # the statistics-free instantiation of this scene class
                    rows.append(x)
            launches = len({x["Dispatch_Id"] for x in rows})
            if not launches:
                return None, f"no integrate_kernel dispatch in pass {group}"
            for x in rows:
                counters[x["Counter_Name"]] = counters.get(x["Counter_Name"], 0.0) + float(x["Counter_Value"]) / launches
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        return None, f"counter passes failed: {e!r}"
    finally:
        shutil.rmtree(td, ignore_errors=True)
    return counters, f"rocprofv3 --pmc, {len(passes)} passes over pt_render on the same frame in {time.perf_counter() - t0:.0f} s"


def pmc_from_file(spp, W, H, kernel_ms):
    """Fallback: a committed summary (profiles/r*_pmc_hbm.json) -- only if it was taken from THIS kernel source and its
    kernel time agrees with the live one within 3 %; otherwise the counters are unknown, not stale."""
    sha = kernel_source_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")), reverse=True):
        j = json.load(open(f))
        if (j.get("spp"), j.get("width"), j.get("height")) != (spp, W, H) or j.get("kernel_source_sha") != sha:
            continue
        if not j.get("kernel_ms") or abs(j["kernel_ms"] - kernel_ms) > 0.03 * kernel_ms:
            continue
        return j["counters_per_launch"], f"{os.path.relpath(f, ROOT)} (kernel_source_sha {sha}, kernel_ms {j['kernel_ms']:.2f})"
    return None, f"no committed summary for kernel_source_sha {sha} at this configuration"


def issue_view(pmc, kms, wave_segments):
    """The second hardware view: the kernel's time follows the number of instructions its waves issue,
    vector or scalar alike.  Issue slots = 2 cycles per wave64 vector instruction on a 32-lane SIMD + 1 per scalar / LDS /
    memory / branch instruction, against 1024 SIMDs x kernel time x 2.4 GHz."""
    keys = ("SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD")
    if not pmc or "SQ_INSTS_VALU" not in pmc or not all(k in pmc for k in keys):
        return None
    others = sum(pmc[k] for k in keys)
    return {"instructions_per_launch": pmc["SQ_INSTS_VALU"] + others,
            "valu": pmc["SQ_INSTS_VALU"], "salu": pmc["SQ_INSTS_SALU"], "branch": pmc["SQ_INSTS_BRANCH"],
            "lds": pmc["SQ_INSTS_LDS"], "smem": pmc["SQ_INSTS_SMEM"], "vmem_rd": pmc["SQ_INSTS_VMEM_RD"],
            "instructions_per_wave_segment": (pmc["SQ_INSTS_VALU"] + others) / wave_segments if wave_segments else None,
            "issue_slot_fraction": (2.0 * pmc["SQ_INSTS_VALU"] + others) / (1024 * kms * 1e-3 * 2.4e9) if kms > 0 else None,
            "what": "(2 x vector + 1 x every other instruction) / (1024 SIMDs x kernel time x 2.4 GHz); s_nop / s_waitcnt not counted"}


def valu_view(pmc, kms, segments):
    """achieved / frac / frac_active_lanes / useful_fraction of one kernel from its SQ counters and live kernel time."""
    if not pmc or "SQ_INSTS_VALU" not in pmc or kms <= 0:
        return {"achieved": None, "frac": None, "valu_active_lane_fraction": None, "frac_active_lanes": None, "useful_fraction": None}
    lane_ops = pmc["SQ_INSTS_VALU"] * 64.0
    achieved = lane_ops / (kms * 1e-3) / 1e12
    lf = None
    if pmc.get("SQ_THREAD_CYCLES_VALU"):
        lf = pmc["SQ_THREAD_CYCLES_VALU"] / lane_ops
        lf = lf if lf <= 1.0 else None
    frac = achieved / PEAK_VALU_TLANEOPS
    return {"achieved": achieved, "frac": frac, "valu_active_lane_fraction": lf,
            "frac_active_lanes": frac * lf if lf is not None else None,
            # the reference's own arithmetic in what was executed: one full Triangle::Intersect + the shading of every segment
            "useful_fraction": segments * (FLOP_FULL_TEST + FLOP_SHADING) / lane_ops if segments else None}


def cpu_baseline(models, target_seconds, pt, scene):
    """Times the CPU oracle (kind "port") on a bounded sample of the same workload: the central rows of the
    1920x1080 frame, -MRR 8, counter RNG; once on every core this process may use and once on 4 threads (the
    reference's shipped THREADS_TO_RUN, CMakeLists.txt:10).  The all-core sample doubles as the accuracy check of
    BASELINE.json's metric: the GPU renders the same rows and passes and the two resolved images are compared."""
    import numpy as np
    import oracle_lib as O
    sc = O.Scene.load(models, "Tor.obj")
    cores = min(O.lib().orc_max_threads(), host_cores())

    def timed(threads, seconds):
        r0 = BASE_H // 2 - 8
        t = time.perf_counter()
        O.render(sc, BASE_W, BASE_H, 1, MRR, rows=(r0, r0 + 16), threads=threads)       # calibration: 30 720 samples
        rate = BASE_W * 16 / max(time.perf_counter() - t, 1e-3)
        rows = 120
        spp = max(1, min(64, int(rate * seconds / (BASE_W * rows))))
        r0 = BASE_H // 2 - rows // 2
        t = time.perf_counter()
        *acc, st = O.render(sc, BASE_W, BASE_H, spp, MRR, rows=(r0, r0 + rows), threads=threads)
        dt = time.perf_counter() - t
        n = BASE_W * rows * spp
        return (n / dt / 1e6, f"rows {r0}-{r0 + rows} x {spp} spp = {n} samples in {dt:.1f} s ({st['segments']} segments)",
                (r0, r0 + rows, spp, acc))

    v_all, what_all, (r0, r1, spp, cpu_acc) = timed(cores, target_seconds * 0.65)
    v_4, what_4, _ = timed(min(4, cores), target_seconds * 0.35)
    # accuracy: the same rows x passes through the C ABI on the GPU -- WITHOUT statistics, i.e. the instantiation the timed
    # launches run (emitter-first last segment and all) -- then the reference's resolve on both
    gs, gs2, gc, _ = scene.render_host(BASE_W, BASE_H, spp, MRR, rows=(r0, r1), want_stats=False)
    g_rgb, _ = pt.resolve_float(BASE_W, r1 - r0, gs, gs2, gc)
    c_rgb, _ = pt.resolve_float(BASE_W, r1 - r0, *cpu_acc)
    d = g_rgb.astype(np.float64) - c_rgb.astype(np.float64)
    g_bgr, c_bgr = pt.quantize(g_rgb, gc), pt.quantize(c_rgb, cpu_acc[2])
    accuracy = {"vs": "cpu_baseline sample (same rows, passes, seed; counter RNG on both sides)",
                "gpu_instantiation": "integrate_kernel<false,false,false,false,false,0> (statistics-free, two pixels per lane: the one the timed launches run)",
                "rmse_rgb_float_image": [float(np.sqrt(np.mean(d[..., k] ** 2))) for k in range(3)],
                "max_abs_diff_float_image": float(np.abs(d).max()),
                "bmp_bytes_differing": int(np.count_nonzero(g_bgr != c_bgr)),
                "accumulators_bit_identical": bool(np.array_equal(gs.view(np.uint32), cpu_acc[0].view(np.uint32)) and
                                                   np.array_equal(gs2.view(np.uint32), cpu_acc[1].view(np.uint32)) and
                                                   np.array_equal(gc, cpu_acc[2])),
                "tolerance": "bit-exact (0); tests/test_gpu_parity.py holds the same bar"}
    return {"value": v_all, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "value_4_threads": v_4,
            "sample": f"Tor.obj 1920x1080 frame, MRR {MRR}, oracle/pt_oracle.c with OpenMP over rows; {cores} threads: {what_all}; "
                      f"4 threads: {what_4}"}, accuracy


def reference_stream_accuracy(models, pt, scene):
    """GPU (counter RNG) against the committed render of the oracle under the REFERENCE's serial streams
    (tests/golden/tor_reference_stream_128x128.npz): two independent Monte-Carlo estimates, compared with the error
    their own variances predict (tests/rng_policy_stats.py states the tolerance)."""
    import numpy as np
    import rng_policy_stats as R
    f = np.load(os.path.join(ROOT, "tests", "golden", "tor_reference_stream_128x128.npz"))
    W, H, passes = int(f["width"]), int(f["height"]), int(f["passes_per_seed"])
    s, s2, c = np.zeros((W * H, 3)), np.zeros((W * H, 3)), np.zeros(W * H, np.int64)
    for seed in f["seeds"]:
        a = scene.render_host(W, H, passes, int(f["mrr"]), error=-1.0, seed=int(seed), want_stats=False)
        s += a[0]; s2 += a[1]; c += a[2]
    r = R.compare((f["sum"], f["sum2"], f["count"]), (s.astype(np.float32), s2.astype(np.float32), c.astype(np.int32)))
    gpu = (s.astype(np.float32), s2.astype(np.float32), c.astype(np.int32))
    binned = {b: R.compare_blocks((f["sum"], f["sum2"], f["count"]), gpu, W, H, b) for b in sorted(R.TOLERANCE["blocks"])}
    ok = True
    try:
        R.assert_same_image(r)
        for rb in binned.values():
            R.assert_same_binned_image(rb)
    except AssertionError:
        ok = False
    seeds = [int(x) for x in f["seeds"]]
    out = {"vs": f"oracle with ORC_RNG_SEQUENTIAL + libm trig (the reference's minstd_rand0 streams), {W}x{H}, seeds {seeds[0]}..{seeds[-1]} x {passes} passes "
                 f"= {len(seeds) * passes} samples per pixel",
           "rmse_rgb_float_image": [ch["rmse_image"] for ch in r["channels"]],
           "rmse_predicted_from_variance": [ch["rmse_image_predicted"] for ch in r["channels"]],
           "z_rms": [ch["z_rms"] for ch in r["channels"]], "z_mean": [ch["z_mean"] for ch in r["channels"]],
           "tolerance": R.TOLERANCE, "within_tolerance": ok}
    # the stated per-channel image tolerance: the resolved float image binned 8 x 8 (and 16 x 16) pixels, RMSE in units of 1/255
    for b, rb in binned.items():
        out[f"binned_{b}x{b}"] = {"rmse_rgb_float_image": [ch["rmse_image"] for ch in rb["channels"]],
                                  "rmse_predicted_from_variance": [ch["rmse_image_predicted"] for ch in rb["channels"]],
                                  "max_abs_diff_float_image": [ch["max_abs_diff_image"] for ch in rb["channels"]],
                                  "z_rms": [ch["z_rms"] for ch in rb["channels"]], "bins": [ch["n"] for ch in rb["channels"]],
                                  "tolerance_rmse": R.TOLERANCE["blocks"][b]["rmse_image_max"]}
    out["brightness_z"] = [ch["brightness_z"] for ch in binned[min(binned)]["channels"]]
    return out


def cxx_frame_leg(n_bands, W, H, spp, steps, warmup, rehearse, error=-1.0, model_dir=MODELS, model_name="Tor.obj"):
    """The frame driven by the C++ host program alone (tools/pt_render.cpp over pt_frame_*: per-device sessions on row bands,
    every slice enqueued on all devices before any wait, ONE direct RCCL group of sends / receives to the root) as a child
    process: the same W x H x spp frame and the same step as the torch-driven legs."""
    if not os.path.exists(EXE):
        return {"error": "path-tracing_amd/bin/pt_render is not built"}
    cmd = [EXE, "--W", str(W), "--H", str(H), "-RPP", str(spp), "-MRR", str(MRR), "-ERR", str(error), "-SEED", "42", "-MODEL_PATH", model_dir,
           "-MODEL_NAME", model_name, "-GPUS", str(n_bands), "-BENCH_STEPS", str(steps), "-BENCH_WARMUP", str(warmup)]
    if rehearse:
        cmd += ["-REHEARSE", "1"]
    try:
        r = run_group(cmd, 150)     # (a frame loop of a second or two + RCCL start-up; a hang must not eat the run's budget)
    except subprocess.TimeoutExpired:
        return {"error": "pt_render timed out after 150 s (killed with its process group)"}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"pt_render rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
    j = json.loads(lines[-1])
    j["what"] = ("pt_render -GPUS N -BENCH_STEPS k (C++ host, pt_frame_*; RCCL send/recv group for N > 1" +
                 ("; REHEARSAL: device copies, bands share devices" if rehearse else "") + "), a child process")
    return j


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp", type=int, default=SPP, help="samples per pixel of the frame (BASELINE configs: 64 / 256 / 1024)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget; 0 skips it")
    ap.add_argument("--pmc", choices=["live", "file", "off"], default="live",
                    help="hardware counters of the timed kernels: collected now with rocprofv3 in child processes (N = 1), read "
                         "from a committed summary of the same kernel source, or omitted")
    ap.add_argument("--no-configs3", action="store_true", help="skip the 3840x2160 x 256 spp strong-scaling leg")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip configs1_64spp / adaptive_default / configs4_replica / cxx_frame")
    ap.add_argument("--save-pmc", default="", help="write the live counters as a summary `--pmc file` can read later (profiles/rNN_pmc_hbm.json)")
    ap.add_argument("--write-bmp", default="", help="resolve rank 0's gathered frame and write it here")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks all render on device 0 and gather over gloo (host copies): exercises the multi-rank "
                         "code path on a one-GPU box; the number it prints is NOT a scaling result")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    extra = world == 1 and not args.no_extra_legs

    # the replicated scenes of BASELINE configs[4] (generated: 64 and 195 torus instances in the room)
    replica_dir, replicas = None, []
    if extra:
        import make_replicated_scene as M
        replica_dir = tempfile.mkdtemp(prefix="pt_replica_")
        for inst in (64, 195):
            name = f"TorX{inst}.obj"
            replicas.append((inst, name, M.generate(os.path.join(ROOT, "models"), replica_dir + "/", name, inst)))

        # the open scene of the skybox leg: Tor.obj without its back wall + a generated sky bitmap
        import make_open_scene
        make_open_scene.generate(os.path.join(ROOT, "models"), replica_dir)

    pmc, pmc_source, pmc_big, pmc_big_source = None, "not collected", None, "not collected"
    pmc_c2, pmc_c2_source, pmc_sky, pmc_sky_source = None, "not collected", None, "not collected"
    if world == 1 and args.pmc == "live":      # child processes; this process has not touched the GPU yet
        deadline = time.perf_counter() + 150
        pmc, pmc_source = pmc_live(PMC_PASSES_TOR, MODELS, "Tor.obj", args.spp, False, deadline)
        if extra:
            pmc_big, pmc_big_source = pmc_live(PMC_PASSES_BIG, replica_dir + "/", replicas[0][1], SPP, True, deadline)
            # BASELINE configs[2] is "the rocprof HBM GB/s run": its own FETCH_SIZE / WRITE_SIZE, at 1024 spp
            pmc_c2, pmc_c2_source = pmc_live([["FETCH_SIZE"], ["WRITE_SIZE"]], MODELS, "Tor.obj", C2_SPP, False, deadline)
            pmc_sky, pmc_sky_source = pmc_live([PMC_SQ], replica_dir + "/", "TorOpen.obj", SPP, False, deadline, sky=os.path.join(replica_dir, "sky.bmp"))

    import numpy as np
    import torch
    import torch.distributed as dist

    pt = importlib.import_module("path-tracing_amd")
    bands = importlib.import_module("path-tracing_amd.bands")
    if not torch.cuda.is_available() or pt.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the integrator has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rccl_ranks_seen = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm
        one = torch.ones(1, dtype=torch.float32, device=torch.device("cpu") if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(one)                  # every rank is really in the communicator
        rccl_ranks_seen = int(one.item())

    scene = pt.Scene.load_obj(MODELS, "Tor.obj", device=local)
    n_tri = scene.counts()[0]
    stream = torch.cuda.current_stream(dev)

    diagnosis = []      # N > 1: one entry per run_frames call

    def run_frames(sc, W, H, spp, steps, warmup, error=-1.0):
        """Renders `warmup` untimed and `steps` timed frames of a W x H x spp image split over `world` ranks (bands.split); returns
        wall seconds (max over ranks), per-step kernel times of this rank, the frame's counters and the gathered frame."""
        r0, r1, row_stride, rows = bands.split(H, world, rank)      # rank k of N: every N-th tile row of 8 image rows, packed (N = 1: the frame)
        npx = rows * W
        # one contiguous band buffer: sum[3n] | sum2[3n] | count[n] (int32 bits) -> a single gather moves everything.
        # Two of them for N > 1: frame k is rendered into one while the gather of frame k-1 still reads the other, so the
        # collective (on RCCL's stream) overlaps the next frame's kernel instead of extending every step.
        n_band = 2 if world > 1 else 1
        band_bufs = [torch.zeros(bands.band_floats(W, rows), dtype=torch.float32, device=dev) for _ in range(n_band)]
        recv_bufs = [None] * n_band     # rank 0: receive buffers, one set per band buffer
        gathered, in_flight, frame_no, events = [None], [None], [0], []
        params = pt.RenderParams(W, H, r0, r1, 0, spp, MRR, 1e-4, error, 42, 0, row_stride)

        def finish_gather():
            if in_flight[0] is not None:
                work, k, _ = in_flight[0]
                work.wait()             # the launch stream now orders after that gather
                gathered[0] = recv_bufs[k]
                in_flight[0] = None

        def step(timed):
            k = frame_no[0] % n_band
            frame_no[0] += 1
            band = band_bufs[k]
            p_sum, p_sum2, p_cnt = band.data_ptr(), band.data_ptr() + 12 * npx, band.data_ptr() + 24 * npx
            band.zero_()
            st = None
            if timed:
                # Timed steps launch without pt_render_stats (the library then runs the kernel instantiation without its
                # diagnostic counters, as a caller that only wants the frame does), bracketed by HIP events on the launch stream.
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                sc.render_device(params, p_sum, p_sum2, p_cnt, stream=stream.cuda_stream, want_stats=False)
                e1.record(stream)
                events.append((e0, e1))
            else:
                st = sc.render_device(params, p_sum, p_sum2, p_cnt, stream=stream.cuda_stream, want_stats=True)
            if world > 1:   # the frame's one collective (RCCL; gloo on host copies when rehearsing)
                finish_gather()         # at most one gather in flight; it read the OTHER band buffer
                send = band.cpu() if args.rehearse_on_one_gpu else band
                recv_bufs[k], work = bands.gather_bands(send, W, H, dist, rank, world, out=recv_bufs[k], async_op=True)
                in_flight[0] = (work, k, send)      # `send` is kept alive until the gather has been waited for
            return st

        def fence():
            finish_gather()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)

        frame_stats = None
        for _ in range(max(warmup, 1)):   # at least one untimed launch: it supplies the frame's counters
            frame_stats = step(False)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        fence()
        elapsed = time.perf_counter() - t0
        kernel_ms = [e0.elapsed_time(e1) for e0, e1 in events]
        if world > 1:
            rdev = torch.device("cpu") if args.rehearse_on_one_gpu else dev
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            # diagnosis (untimed): every rank's mean kernel time, and ONE gather alone -- all kernels are done, nothing overlaps it
            mine = torch.tensor([sum(kernel_ms) / max(len(kernel_ms), 1)], dtype=torch.float64, device=rdev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            k = (frame_no[0] - 1) % n_band
            send = band_bufs[k].cpu() if args.rehearse_on_one_gpu else band_bufs[k]
            fence()
            tg = time.perf_counter()
            recv_bufs[k], work = bands.gather_bands(send, W, H, dist, rank, world, out=recv_bufs[k], async_op=True)
            in_flight[0] = (work, k, send)
            fence()
            diagnosis.append({"frame": f"{W}x{H} x {spp} spp", "kernel_ms_per_rank": [float(x.item()) for x in every],
                              "gather_alone_ms": (time.perf_counter() - tg) * 1e3, "ms_per_step": elapsed / steps * 1e3,
                              "what": "kernel_ms_per_rank = HIP events around each rank's launches (mean over the timed steps); gather_alone_ms = one "
                                      "gather of the bands to rank 0 with every kernel already done (host clock, barrier to barrier)"})
        frame = gathered[0] if world > 1 else [band_bufs[0]]
        return elapsed, kernel_ms, frame_stats, frame, rows

    def leg(sc, W, H, spp, steps, workload, error=-1.0):
        e, kms, st, _, _ = run_frames(sc, W, H, spp, steps, 1, error)
        return {"workload": workload, "value": W * H * spp * steps / e / 1e6, "unit": "Msamples/s", "steps": steps,
                "ms_per_step": e / steps * 1e3, "kernel_ms": sum(kms) / max(len(kms), 1)}, st

    W, H = bands.frame_for(world)
    elapsed, kernel_ms, frame_stats, frame, rows = run_frames(scene, W, H, args.spp, args.steps, args.warmup)
    c3 = None
    if not args.no_configs3:
        c3_steps = max(1, min(args.steps, 5))
        c3_elapsed, c3_kms, c3_stats, _, c3_rows = run_frames(scene, C3_W, C3_H, C3_SPP, c3_steps, 1)
        c3 = {"workload": f"BASELINE configs[3]: Tor.obj {C3_W}x{C3_H} x {C3_SPP} spp, -MRR {MRR}, -ERR -1, {world} band(s) of "
                          f"{c3_rows} rows" + (f" (every {world}th tile row of 8)" if world > 1 else "") + (", one RCCL gather of 28 B/pixel to rank 0" if world > 1 else ""),
              "value": C3_W * C3_H * C3_SPP * c3_steps / c3_elapsed / 1e6, "unit": "Msamples/s", "scaling": "strong",
              "steps": c3_steps, "ms_per_step": c3_elapsed / c3_steps * 1e3,
              "kernel_ms_rank0": sum(c3_kms) / max(len(c3_kms), 1)}

    legs = {}
    if extra:
        # BASELINE configs[1] on identical work from round to round (the headline moved from 64 to 256 spp in round 2)
        legs["configs1_64spp"], _ = leg(scene, BASE_W, BASE_H, 64, 5, f"BASELINE configs[1]: Tor.obj {BASE_W}x{BASE_H} x 64 spp, -MRR {MRR}, -ERR -1")
        # SURVEY 8(d) "run every config twice": the reference's default, adaptive sampling on (-ERR 0.001, main.cpp:118-125)
        ad, ad_st = leg(scene, BASE_W, BASE_H, SPP, 5, f"Tor.obj {BASE_W}x{BASE_H} x {SPP} spp, -MRR {MRR}, -ERR 0.001 (reference default: "
                        "adaptive sampling skips low-variance pixels on 3 of 4 passes after pass 10)", error=0.001)
        ad["nominal_samples"] = BASE_W * BASE_H * SPP
        ad["samples_traced"] = int(ad_st["samples_traced"])
        ad["value_nominal"] = ad["value"]
        ad["value_traced"] = ad["value"] * ad["samples_traced"] / ad["nominal_samples"]
        ad["what"] = "value = nominal W*H*spp per second (the reference's own accounting); value_traced counts only samples really traced"
        legs["adaptive_default"] = ad
        # BASELINE configs[4]: the replicated scenes, 1080p x 256 spp, driver-timed
        rep = {}
        for inst, name, tri in replicas:
            big_scene = pt.Scene.load_obj(replica_dir + "/", name, device=local)
            steps = 5 if inst == 64 else 3
            r, st = leg(big_scene, BASE_W, BASE_H, SPP, steps, f"BASELINE configs[4]: Tor.obj torus x {inst} in the room = {tri} triangles, "
                        f"{BASE_W}x{BASE_H} x {SPP} spp, -MRR {MRR}, -ERR -1")
            r["triangles"] = tri
            wseg = float(st["wave_segments"]) or None
            r["node_rounds_per_wave_segment"] = st["wave_node_rounds"] / wseg if wseg else None
            r["exact_rounds_per_wave_segment"] = st["wave_exact_iterations"] / wseg if wseg else None
            r["exact_tests_per_segment"] = st["exact_tests"] / st["segments"] if st["segments"] else None
            r["host"] = big_scene.timings()
            if inst == 64:
                v = valu_view(pmc_big, r["kernel_ms"], float(st["segments"]))
                rf = {"bound": "valu_issue", "peak": PEAK_VALU_TLANEOPS, "unit": "Tlane-op/s", **v,
                      "kernel": "pt::integrate_kernel<false,true,false,false,false,0>", "kernel_ms": r["kernel_ms"],
                      "issue": issue_view(pmc_big, r["kernel_ms"], wseg), "counters_source": pmc_big_source,
                      "l1_hit_rate": (1 - pmc_big["TCP_TCC_READ_REQ_sum"] / pmc_big["TCP_TOTAL_CACHE_ACCESSES_sum"])
                      if pmc_big and pmc_big.get("TCP_TOTAL_CACHE_ACCESSES_sum") else None,
                      "l1_accesses_per_cu_cycle": (pmc_big["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256 / (pmc_big["GRBM_GUI_ACTIVE"] / 8))
                      if pmc_big and pmc_big.get("TCP_TOTAL_CACHE_ACCESSES_sum") and pmc_big.get("GRBM_GUI_ACTIVE") else None}
                r["roofline"] = rf
            rep[f"x{inst}"] = r
            if inst == 64:   # the same scene with the reference's default -ERR 0.001: the box-tree kernel's batches of adaptive sampling
                ra, sta = leg(big_scene, BASE_W, BASE_H, SPP, 2, f"the x {inst} scene, {BASE_W}x{BASE_H} x {SPP} spp, -MRR {MRR}, -ERR 0.001 (reference default)", error=0.001)
                ra["samples_traced"] = int(sta["samples_traced"])
                ra["value_traced"] = ra["value"] * ra["samples_traced"] / (BASE_W * BASE_H * SPP)
                ra["what"] = "value = nominal W*H*spp per second (the reference's own accounting)"
                rep[f"x{inst}_adaptive_default"] = ra
            big_scene.close()
        legs["configs4_replica"] = rep

        # BASELINE configs[2]: Tor.obj 1080p x 1024 spp, "the rocprof HBM GB/s run" -- driver-timed, with the HBM bytes of THAT launch
        c2, c2_st = leg(scene, BASE_W, BASE_H, C2_SPP, 3, f"BASELINE configs[2]: Tor.obj {BASE_W}x{BASE_H} x {C2_SPP} spp, -MRR {MRR}, -ERR -1")
        c2_algo = BASE_W * BASE_H * 28 * 2 + n_tri * 56
        c2_traffic = (pmc_c2["FETCH_SIZE"] + pmc_c2["WRITE_SIZE"]) * 1024.0 if pmc_c2 and "FETCH_SIZE" in pmc_c2 and "WRITE_SIZE" in pmc_c2 else None
        c2["hbm"] = {"algorithmic_bytes": c2_algo, "traffic_bytes": c2_traffic,
                     "traffic_over_algorithmic": c2_traffic / c2_algo if c2_traffic else None,
                     "achieved": (c2_traffic or c2_algo) / (c2["kernel_ms"] * 1e-3) / 1e9 if c2["kernel_ms"] > 0 else None,
                     "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": ((c2_traffic or c2_algo) / (c2["kernel_ms"] * 1e-3) / 1e9) / PEAK_HBM_GBS if c2["kernel_ms"] > 0 else None,
                     "counters_source": pmc_c2_source,
                     "fetch_kib": pmc_c2.get("FETCH_SIZE") if pmc_c2 else None, "write_kib": pmc_c2.get("WRITE_SIZE") if pmc_c2 else None}
        if args.save_pmc and pmc_c2:
            json.dump({"kernel": "pt::integrate_kernel<false,false,false,false,false,0>", "width": BASE_W, "height": BASE_H, "spp": C2_SPP, "mrr": MRR,
                       "kernel_source_sha": kernel_source_sha(), "kernel_ms": c2["kernel_ms"], "counters_per_launch": pmc_c2,
                       "hbm": c2["hbm"], "collected_by": "bench.py --save-pmc (configs2_1024spp): " + pmc_c2_source,
                       "note": "FETCH_SIZE / WRITE_SIZE in KiB from separate --pmc passes over pt_render on the 1024-spp frame"},
                      open(re.sub(r"(_pmc_hbm)?\.json$", "", args.save_pmc) + "_c2_pmc_hbm.json", "w"), indent=1)
        legs["configs2_1024spp"] = c2
        # Row N1 of the round-3 review: an OPEN scene under a sky.  Most paths end on their first or second segment; the skybox
        # instantiation regenerates paths (a lane whose path has ended starts its pixel's next pass at once).
        open_scene = pt.Scene.load_obj(replica_dir + "/", "TorOpen.obj", device=local)
        open_scene.set_skybox(os.path.join(replica_dir, "sky.bmp"))
        sk, sk_st = leg(open_scene, BASE_W, BASE_H, SPP, 5, f"Tor.obj without its back wall (266 triangles) under a 256x128 sky bitmap, {BASE_W}x{BASE_H} x {SPP} spp, "
                        f"-MRR {MRR}, -ERR -1: integrate_kernel<true,false,...> with path regeneration")
        n_samples = float(BASE_W * BASE_H * SPP)
        sk["segments_per_sample"] = sk_st["segments"] / n_samples
        sk["misses_per_sample"] = sk_st["misses"] / n_samples
        sk["contributing_per_sample"] = sk_st["contributing"] / n_samples
        sk["live_rays_per_wave_segment"] = sk_st["segments"] / float(sk_st["wave_segments"]) if sk_st["wave_segments"] else None
        sk["live_rays_per_wave_segment_of"] = 64
        sk["without_regeneration"] = {"live_rays_per_wave_segment": 32.7, "value": 5297.0, "spp": 64,
                                      "source": "profiles/r04_open_scene_probe_before.jsonl (the same scene and kernel before regeneration, another box)"}
        v = valu_view(pmc_sky, sk["kernel_ms"], float(sk_st["segments"]))
        sk["roofline"] = {"bound": "valu_issue", "peak": PEAK_VALU_TLANEOPS, "unit": "Tlane-op/s", **v, "kernel_ms": sk["kernel_ms"],
                          "kernel": "pt::integrate_kernel<true,false,false,false,false,0>", "counters_source": pmc_sky_source,
                          "issue": issue_view(pmc_sky, sk["kernel_ms"], float(sk_st["wave_segments"]) or None)}
        if args.cpu_seconds > 0:      # parity of THIS leg: a band of the same frame against the oracle, bit for bit
            import oracle_lib as O
            osc = O.Scene.load(replica_dir + "/", "TorOpen.obj")
            osc.set_skybox(os.path.join(replica_dir, "sky.bmp"))
            r0 = BASE_H // 2 - 8
            cs, cs2, cc, cst = O.render(osc, BASE_W, BASE_H, 8, MRR, rows=(r0, r0 + 16), threads=min(O.lib().orc_max_threads(), host_cores()))
            gs, gs2, gc, _ = open_scene.render_host(BASE_W, BASE_H, 8, MRR, rows=(r0, r0 + 16), want_stats=False)
            sk["accumulators_bit_identical_to_oracle"] = bool(np.array_equal(gs.view(np.uint32), cs.view(np.uint32)) and
                                                              np.array_equal(gs2.view(np.uint32), cs2.view(np.uint32)) and np.array_equal(gc, cc))
            sk["oracle_sample"] = f"rows {r0}-{r0 + 16} x 8 spp, {cst['segments']} segments, {cst['misses']} misses"
        legs["skybox_open"] = sk
        open_scene.close()

    # the C++ host alone on the same frame(s): N = 1 must agree with `value`; for N > 1 it is the direct-RCCL frame path
    cxx = None
    if not args.no_extra_legs:
        if world > 1:
            store = dist.distributed_c10d._get_default_store()
            dist.barrier()
            torch.cuda.synchronize(dev)
            if rank == 0:       # the other ranks wait on the store (on the CPU): their GPUs stay idle for the child
                cxx = {"weak": cxx_frame_leg(world, W, H, args.spp, max(1, min(args.steps, 10)), 1, args.rehearse_on_one_gpu)}
                if not args.no_configs3:
                    cxx["configs3_strong"] = cxx_frame_leg(world, C3_W, C3_H, C3_SPP, max(1, min(args.steps, 5)), 1, args.rehearse_on_one_gpu)
                store.set("cxx_frame_done", "1")
            else:
                store.wait(["cxx_frame_done"], datetime.timedelta(seconds=450))   # two children of at most 150 s each
        else:
            cxx = {"weak": cxx_frame_leg(1, W, H, args.spp, max(1, min(args.steps, 10)), 1, False)}

    if rank == 0:
        k = max(args.steps, 1)
        npx = rows * W
        samples_per_step = W * H * args.spp
        seg = float(frame_stats["segments"])
        kms = sum(kernel_ms) / max(len(kernel_ms), 1)
        # W*H*28 B of accumulators in and out per render call + the scene once
        algo_bytes = npx * 28 * 2 + n_tri * 56
        if pmc is None and world == 1 and args.pmc != "off":
            why = pmc_source
            pmc, pmc_source = pmc_from_file(args.spp, W, H, kms)
            if pmc is None:
                pmc_source = f"{why}; {pmc_source}"
        traffic = None
        if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            # KiB -> bytes.  The guide's x2 on FETCH_SIZE is for wide streaming reads; this kernel's reads are 4-byte
            # strided accumulator loads, calibrated 0.76-1.0 : 1 on their known byte count
            traffic = (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        if args.save_pmc and pmc and pmc_source.startswith("rocprofv3"):
            json.dump({"kernel": "pt::integrate_kernel<false,false,false,false,false,0>", "width": W, "height": H, "spp": args.spp, "mrr": MRR,
                       "kernel_source_sha": kernel_source_sha(), "kernel_ms": kms, "counters_per_launch": pmc,
                       "collected_by": "bench.py --save-pmc: " + pmc_source,
                       "note": "FETCH_SIZE / WRITE_SIZE in KiB from separate --pmc passes; SQ_INSTS_VALU counts wave-instructions"},
                      open(args.save_pmc, "w"), indent=1)
        v = valu_view(pmc, kms, seg)
        wseg = float(frame_stats.get("wave_segments", 0)) or None
        ref_eq = seg * n_tri * FLOP_PER_TEST / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        out = {
            "metric": f"Msamples/sec (WxHxspp/wall) on Tor.obj 1080p x {args.spp} spp",
            "value": samples_per_step * k / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / k * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (models/Tor.obj, 270 triangles, seed 42, counter RNG)" +
                    (" -- REHEARSAL: all ranks on one GPU, gloo gather; not a scaling measurement" if args.rehearse_on_one_gpu else ""),
            "config": {"workload": f"Tor.obj {W}x{H} x {args.spp} spp, -MRR {MRR}, -ERR -1 (adaptive off), -EPS 1e-4; "
                                   f"{world} band(s) of {rows} rows" + (f" (every {world}th tile row of 8)" if world > 1 else "") + (", one RCCL gather of 28 B/pixel to rank 0" if world > 1 else ""),
                       "width": W, "height": H, "spp": args.spp, "max_ray_reflections": MRR, "triangles": n_tri,
                       "parallelism": f"rowband{world}" if world == 1 else f"tilerows_interleaved{world}"},
            "roofline": {"bound": "valu_issue", "achieved": v["achieved"], "peak": PEAK_VALU_TLANEOPS, "unit": "Tlane-op/s",
                         "frac": v["frac"],
                         "traffic": traffic,
                         "kernel": "pt::integrate_kernel<false,false,false,false,false,0>", "kernel_ms": kms,
                         "what": "achieved = executed VALU lane-operations (SQ_INSTS_VALU x 64) / live HIP-event kernel time; "
                                 "peak = 1024 SIMDs x 32 lanes x 2.4 GHz (no FMA: parity forbids contraction); frac counts issued "
                                 "instructions whatever their lane mask, frac_active_lanes = frac x active-lane fraction, useful_fraction = "
                                 f"segments x ({FLOP_FULL_TEST} + {FLOP_SHADING}) flop of the reference's own arithmetic / executed lane-operations",
                         "counters_source": pmc_source, "kernel_source_sha": kernel_source_sha(),
                         "valu_instructions_per_launch": pmc.get("SQ_INSTS_VALU") if pmc else None,
                         "valu_active_lane_fraction": v["valu_active_lane_fraction"],
                         "frac_active_lanes": v["frac_active_lanes"],
                         "useful_fraction": v["useful_fraction"],
                         "issue": issue_view(pmc, kms, wseg),
                         "segments_per_launch": seg, "exact_tests_per_segment": frame_stats["exact_tests"] / seg if seg else None,
                         "reference_equivalent_tflops": ref_eq, "reference_equivalent_over_fp32_peak": ref_eq / PEAK_FP32_VALU_TFLOPS,
                         "flop_per_test": FLOP_PER_TEST,
                         "hbm": {"algorithmic_bytes": algo_bytes, "traffic_bytes": traffic,
                                 "traffic_over_algorithmic": traffic / algo_bytes if traffic else None,
                                 "why": "the timed kernel (two pixels per lane) leaves the accumulators in memory: read-modify-written (three sparse cache "
                                        "lines in and out) when a path reaches an emitter, 1 % of the samples; with adaptive sampling off nothing else touches them",
                                 "achieved": (traffic or algo_bytes) / (kms * 1e-3) / 1e9 if kms > 0 else 0.0,
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": ((traffic or algo_bytes) / (kms * 1e-3) / 1e9) / PEAK_HBM_GBS if kms > 0 else 0.0}},
        }
        if rccl_ranks_seen is not None:
            out["rccl_ranks_seen"] = rccl_ranks_seen
        if diagnosis:
            out["multi_gpu_diagnosis"] = diagnosis
        if c3 is not None:
            out["configs3_strong"] = c3
        out.update(legs)
        if cxx is not None:
            w = cxx["weak"]
            if "value" in w:
                w["over_torch_leg"] = w["value"] / out["value"]
            if "configs3_strong" in cxx and c3 is not None and "value" in cxx["configs3_strong"]:
                cxx["configs3_strong"]["over_torch_leg"] = cxx["configs3_strong"]["value"] / c3["value"]
            out["cxx_frame"] = cxx
        if world == 1:
            # the host-buffer boundary (pt_render_host): same frame, accumulators staged over PCIe both ways; not `value`
            acc = (np.zeros((npx, 3), np.float32), np.zeros((npx, 3), np.float32), np.zeros(npx, np.int32))
            scene.render_host(W, H, args.spp, MRR, accum=acc, want_stats=False)      # warm-up
            th = time.perf_counter()
            scene.render_host(W, H, args.spp, MRR, accum=acc, want_stats=False)
            th = time.perf_counter() - th
            out["pcie_inclusive"] = {"value": samples_per_step / th / 1e6, "unit": "Msamples/s", "ms_per_step": th * 1e3,
                                     "what": "pt_render_host: 116 MB of accumulators host->device and back around the same frame"}
        if world == 1 and os.path.exists(EXE):
            # end to end: the stand-alone front end from process start to the BMP on disk -- exec + dynamic
            # linking, OBJ/MTL parse, HIP start-up, table upload, hierarchy build, render, read-back, resolve (powf), BMP write,
            # exit.  A child process; -T0_NS lets it report the time before main().
            runs = []
            for _ in range(2):   # twice: the runtime's start-up beside a parent that holds the device varies from 0.05 to 0.25 s; both are reported
                with tempfile.TemporaryDirectory() as td:
                    cmd = [EXE, "--W", str(W), "--H", str(H), "-RPP", str(args.spp), "-MRR", str(MRR), "-ERR", "-1", "-UPDATE", "0",
                           "-QUIET", "1", "-SEED", "42", "-MODEL_PATH", MODELS, "-OUT", os.path.join(td, "frame.bmp"), "-TIMING", "1",
                           "-FASTEXIT", "1", "-T0_NS", str(time.time_ns())]
                    te = time.perf_counter()
                    r = subprocess.run(cmd, cwd=td, capture_output=True, text=True)
                    te = time.perf_counter() - te
                    ok = r.returncode == 0 and os.path.getsize(os.path.join(td, "frame.bmp")) == 54 + W * H * 3
                    phases = None
                    for line in r.stderr.splitlines():
                        if line.startswith("{") and "hip_startup_s" in line:
                            phases = json.loads(line)
                runs.append((te if ok else float("inf"), te, ok, phases))
            _, te, ok, phases = min(runs, key=lambda x: x[0])
            unexplained = None
            if phases:
                unexplained = te - phases["pre_main_s"] - phases["main_s"]
            out["end_to_end"] = {"value": samples_per_step / te / 1e6 if ok else None, "unit": "Msamples/s", "seconds": te,
                                 "seconds_of_both_runs": [x[1] for x in runs],
                                 "phases": phases, "exit_and_wait_s": unexplained,
                                 "what": "pt_render (C++ front end) as a child process: process start -> BMP on disk -> exit, the faster of two "
                                         "runs; seconds = pre_main_s + main_s + exit_and_wait_s"}
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"], out["accuracy"] = cpu_baseline(MODELS, args.cpu_seconds, pt, scene)
            out["accuracy"]["vs_reference_stream"] = reference_stream_accuracy(MODELS, pt, scene)
        if args.write_bmp:
            parts = [t.cpu().numpy() for t in frame]
            s, s2, c = bands.assemble(parts, W, H, world)
            bgr, disp = pt.resolve(W, H, s, s2, c)
            pt.write_bmp(args.write_bmp, bgr)
            out["config"]["dispersion_max_min_avg"] = [float(d) for d in disp]
        print(json.dumps(out), flush=True)
    if replica_dir:
        shutil.rmtree(replica_dir, ignore_errors=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
