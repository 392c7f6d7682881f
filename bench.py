#!/usr/bin/env python3
"""Benchmark of the radiance-integrator hot path (BASELINE.json metric: Msamples/s = W*H*spp / wall on Tor.obj).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Without WORLD_SIZE in the environment `--gpus N` starts its own N rank processes (one per GPU, RCCL) and prints rank
0's line; under torch.distributed.run it is one of the ranks.

One "step" = one complete frame: zero the accumulators, trace every sample of every pixel of this rank's row band
(all passes x segments x triangles in one kernel launch through the C ABI), and -- for N > 1 -- the single RCCL
gather of the accumulator bands to rank 0 (issued asynchronously: it overlaps the next frame's kernel, every gather is
complete before the timed region ends).  At N = 1 the frame is the configuration north_star quotes its target on:
models/Tor.obj, 1920x1080, 256 spp, -MRR 8, adaptive sampling off (-ERR -1, so all W*H*spp samples are traced;
`--spp 64` is BASELINE configs[1], `--spp 1024` configs[2]).  For N > 1 the image grows with N
(path-tracing_amd/bands.py: frame_for) so that every GPU keeps a 1080p-sized band: weak scaling, `value`.  Next to it
every run also times BASELINE configs[3] as written -- 3840x2160 x 256 spp cut into N row bands, one gather -- and
reports it as `configs3_strong` (strong scaling: the frame is fixed, 2160/N rows per GPU).

Rank 0 prints ONE JSON line.  Besides the contract's fields it carries
  roofline      the integrator kernel against the roofline that bounds it, vector-ALU issue (the path is
                neither HBM- nor MFMA-bound): achieved = VALU lane-operations the kernel EXECUTES per second
                (SQ_INSTS_VALU x 64, counted by rocprofv3 --pmc on this very command in child processes) over the
                HIP-event kernel time measured live on the launch stream; peak = 78.6 T lane-op/s (1024 SIMDs x 32 lanes
                x 2.4 GHz, no FMA: parity forbids contraction).  traffic = HBM bytes per launch from FETCH_SIZE and
                WRITE_SIZE (separate passes).  reference_equivalent_tflops = the REFERENCE's brute-force work
                (segments x triangles x 31.5 flop) per second -- what the culling hierarchy saves, not a roofline.
  cpu_baseline  the CPU oracle (a port of the reference's algorithm) timed on this box's host cores on a bounded sample
  accuracy      the other half of BASELINE's metric: per-channel RMSE against the CPU path
"""
import argparse
import ctypes as C
import csv
import glob
import hashlib
import importlib
import json
import os
import re
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLOP_PER_TEST = 31.5            # reference's own average over its stage-exit mix
PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz
PEAK_VALU_TLANEOPS = 78.6432    # the same without FMA: one operation per lane per clock
PEAK_HBM_GBS = 8000.0
BASE_W, BASE_H, SPP, MRR = 1920, 1080, 256, 8
C3_W, C3_H, C3_SPP = 3840, 2160, 256      # BASELINE configs[3]
KERNEL_SOURCES = ["pt_kernels.hip", "pt_kernels.hpp", "pt_fastfp.hpp", "pt_scene.cpp", "pt_scene.hpp", "pt_capi.cpp"]


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


def timed_instantiation(kernel_name):
    """integrate_kernel<SKY, BIG, STATS, ENV>: the timed launches are the statistics-free ones (third argument false)."""
    m = re.search(r"integrate_kernel<(\w+),(\w+),(\w+),(\w+)>", kernel_name.replace(" ", ""))
    return bool(m) and m.group(3) == "false"


def kernel_source_sha():
    """Identifies the kernel a PMC summary was taken from: sha256 over the sources the integrator is built from."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "path-tracing_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves (before anything in this process touches the GPU)
# ---------------------------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    import torch
    have = torch.cuda.device_count()          # counting devices does not initialise the GPU
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


# ---------------------------------------------------------------------------------------------------------------------
# Hardware counters of the timed kernel: rocprofv3 --pmc over THIS command, in child processes, before the parent
# touches the GPU.  One counter group per pass, never combined with a trace domain (MI355X_MICROARCH.md "rocprofv3 PMC
# slots": FETCH_SIZE and WRITE_SIZE do not fit one pass).
# ---------------------------------------------------------------------------------------------------------------------
PMC_PASSES = [["SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_SMEM",
               "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE"], ["FETCH_SIZE"], ["WRITE_SIZE"]]


def pmc_live(args, timeout_s=240):
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    counters, t0 = {}, time.perf_counter()
    td = tempfile.mkdtemp(prefix="pt_pmc_")
    try:
        for i, group in enumerate(PMC_PASSES):
            left = timeout_s - (time.perf_counter() - t0)
            if left < 20:
                return None, "time budget of the counter passes exhausted"
            d = os.path.join(td, f"p{i}")
            cmd = [exe, "--pmc", *group, "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-child", "--spp", str(args.spp), "--steps", "1", "--warmup", "1"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=left, cwd=td, env=dict(os.environ, TMPDIR=td))
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 pass {group} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}"
            rows = [x for x in csv.DictReader(open(max(files, key=os.path.getmtime))) if "integrate_kernel" in x["Kernel_Name"]]
            # the timed launches run the instantiation without statistics (third template argument false)
            rows = [x for x in rows if timed_instantiation(x["Kernel_Name"])]
            launches = len({x["Dispatch_Id"] for x in rows})
            if not launches:
                return None, f"no integrate_kernel dispatch in pass {group}"
            for x in rows:
                counters[x["Counter_Name"]] = counters.get(x["Counter_Name"], 0.0) + float(x["Counter_Value"]) / launches
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        return None, f"counter passes failed: {e!r}"
    finally:
        shutil.rmtree(td, ignore_errors=True)
    return counters, f"rocprofv3 --pmc, {len(PMC_PASSES)} passes over this command in {time.perf_counter() - t0:.0f} s"


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
    # accuracy: the same rows x passes through the C ABI on the GPU, then the reference's resolve on both
    gs, gs2, gc, _ = scene.render_host(BASE_W, BASE_H, spp, MRR, rows=(r0, r1))
    g_rgb, _ = pt.resolve_float(BASE_W, r1 - r0, gs, gs2, gc)
    c_rgb, _ = pt.resolve_float(BASE_W, r1 - r0, *cpu_acc)
    d = g_rgb.astype(np.float64) - c_rgb.astype(np.float64)
    g_bgr, c_bgr = pt.quantize(g_rgb, gc), pt.quantize(c_rgb, cpu_acc[2])
    accuracy = {"vs": "cpu_baseline sample (same rows, passes, seed; counter RNG on both sides)",
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
    ok = True
    try:
        R.assert_same_image(r)
    except AssertionError:
        ok = False
    return {"vs": "oracle with ORC_RNG_SEQUENTIAL + libm trig (the reference's minstd_rand0 streams), 128x128, seeds 42..49 x 512 passes",
            "rmse_rgb_float_image": [ch["rmse_image"] for ch in r["channels"]],
            "rmse_predicted_from_variance": [ch["rmse_image_predicted"] for ch in r["channels"]],
            "z_rms": [ch["z_rms"] for ch in r["channels"]], "z_mean": [ch["z_mean"] for ch in r["channels"]],
            "tolerance": R.TOLERANCE, "within_tolerance": ok}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp", type=int, default=SPP, help="samples per pixel of the frame (BASELINE configs: 64 / 256 / 1024)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget; 0 skips it")
    ap.add_argument("--pmc", choices=["live", "file", "off"], default="live",
                    help="hardware counters of the timed kernel: collected now with rocprofv3 in child processes (N = 1), read "
                         "from a committed summary of the same kernel source, or omitted")
    ap.add_argument("--no-configs3", action="store_true", help="skip the 3840x2160 x 256 spp strong-scaling leg")
    ap.add_argument("--save-pmc", default="", help="write the live counters as a summary `--pmc file` can read later (profiles/rNN_pmc_hbm.json)")
    ap.add_argument("--write-bmp", default="", help="resolve rank 0's gathered frame and write it here")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks all render on device 0 and gather over gloo (host copies): exercises the multi-rank "
                         "code path on a one-GPU box; the number it prints is NOT a scaling result")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the run rocprofv3 wraps: launches only
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    pmc, pmc_source = None, "not collected"
    if world == 1 and args.pmc == "live" and not args.pmc_child:
        pmc, pmc_source = pmc_live(args)      # child processes; this process has not touched the GPU yet

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

    models = os.path.join(ROOT, "models") + "/"
    scene = pt.Scene.load_obj(models, "Tor.obj", device=local)
    n_tri = scene.counts()[0]
    stream = torch.cuda.current_stream(dev)

    def run_frames(W, H, spp, steps, warmup):
        """Renders `warmup` untimed and `steps` timed frames of a W x H x spp image cut into `world` row bands; returns
        wall seconds (max over ranks), per-step kernel times of this rank, the frame's counters and the gathered frame."""
        r0, r1 = bands.band_rows(H, world, rank)
        rows = r1 - r0
        npx = rows * W
        # one contiguous band buffer: sum[3n] | sum2[3n] | count[n] (int32 bits) -> a single gather moves everything.
        # Two of them for N > 1: frame k is rendered into one while the gather of frame k-1 still reads the other, so the
        # collective (on RCCL's stream) overlaps the next frame's kernel instead of extending every step.
        n_band = 2 if world > 1 else 1
        band_bufs = [torch.zeros(bands.band_floats(W, rows), dtype=torch.float32, device=dev) for _ in range(n_band)]
        recv_bufs = [None] * n_band     # rank 0: receive buffers, one set per band buffer
        gathered, in_flight, frame_no, events = [None], [None], [0], []
        params = pt.RenderParams(W, H, r0, r1, 0, spp, MRR, 1e-4, -1.0, 42)

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
                scene.render_device(params, p_sum, p_sum2, p_cnt, stream=stream.cuda_stream, want_stats=False)
                e1.record(stream)
                events.append((e0, e1))
            else:
                st = scene.render_device(params, p_sum, p_sum2, p_cnt, stream=stream.cuda_stream, want_stats=True)
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
        frame = gathered[0] if world > 1 else [band_bufs[0]]
        return elapsed, kernel_ms, frame_stats, frame, rows

    if args.pmc_child:
        run_frames(BASE_W, BASE_H, args.spp, args.steps, args.warmup)
        return

    W, H = bands.frame_for(world)
    elapsed, kernel_ms, frame_stats, frame, rows = run_frames(W, H, args.spp, args.steps, args.warmup)
    c3 = None
    if not args.no_configs3:
        c3_steps = max(1, min(args.steps, 5))
        c3_elapsed, c3_kms, c3_stats, _, c3_rows = run_frames(C3_W, C3_H, C3_SPP, c3_steps, 1)
        c3 = {"workload": f"BASELINE configs[3]: Tor.obj {C3_W}x{C3_H} x {C3_SPP} spp, -MRR {MRR}, -ERR -1, {world} row band(s) of "
                          f"{c3_rows} rows" + (", one RCCL gather of 28 B/pixel to rank 0" if world > 1 else ""),
              "value": C3_W * C3_H * C3_SPP * c3_steps / c3_elapsed / 1e6, "unit": "Msamples/s", "scaling": "strong",
              "steps": c3_steps, "ms_per_step": c3_elapsed / c3_steps * 1e3,
              "kernel_ms_rank0": sum(c3_kms) / max(len(c3_kms), 1)}

    if rank == 0:
        k = max(args.steps, 1)
        npx = rows * W
        samples_per_step = W * H * args.spp
        seg = float(frame_stats["segments"])
        kms = sum(kernel_ms) / max(len(kernel_ms), 1)
        n_chunks = max(1, frame_stats.get("n_chunks", 1))
        # W*H*28 B of accumulators in and out per render call + the scene once
        algo_bytes = npx * 28 * 2 + n_tri * 56
        if pmc is None and world == 1 and args.pmc != "off":
            why = pmc_source
            pmc, pmc_source = pmc_from_file(args.spp, W, H, kms)
            if pmc is None:
                pmc_source = f"{why}; {pmc_source}"
        lane_ops = traffic = lane_frac = util = None
        if pmc:
            if "SQ_INSTS_VALU" in pmc:
                lane_ops = pmc["SQ_INSTS_VALU"] * 64.0
                if pmc.get("SQ_THREAD_CYCLES_VALU"):
                    lf = pmc["SQ_THREAD_CYCLES_VALU"] / lane_ops
                    lane_frac = lf if lf <= 1.0 else None
            if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                # KiB -> bytes.  The guide's x2 on FETCH_SIZE is for wide streaming reads; this kernel's reads are 4-byte
                # strided accumulator loads, calibrated 0.76-1.0 : 1 on their known byte count
                traffic = (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        if args.save_pmc and pmc and pmc_source.startswith("rocprofv3"):
            json.dump({"kernel": "pt::integrate_kernel<false,false,false,false>", "width": W, "height": H, "spp": args.spp, "mrr": MRR,
                       "kernel_source_sha": kernel_source_sha(), "kernel_ms": kms, "counters_per_launch": pmc,
                       "collected_by": "bench.py --save-pmc: " + pmc_source,
                       "note": "FETCH_SIZE / WRITE_SIZE in KiB from separate --pmc passes; SQ_INSTS_VALU counts wave-instructions"},
                      open(args.save_pmc, "w"), indent=1)
        achieved = lane_ops / (kms * 1e-3) / 1e12 if lane_ops and kms > 0 else None
        # The second hardware view: the kernel's time follows the number of instructions its waves issue,
        # vector or scalar alike (calibration builds with 200 extra instructions per wave-segment).  Issue slots = 2 cycles per
        # wave64 vector instruction on a 32-lane SIMD + 1 per scalar / LDS / memory / branch instruction, against
        # 1024 SIMDs x kernel time x 2.4 GHz.
        issue = None
        if pmc and all(k in pmc for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD")):
            others = sum(pmc[k] for k in ("SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD"))
            wseg = float(frame_stats.get("wave_segments", 0)) or None
            issue = {"instructions_per_launch": pmc["SQ_INSTS_VALU"] + others,
                     "valu": pmc["SQ_INSTS_VALU"], "salu": pmc["SQ_INSTS_SALU"], "branch": pmc["SQ_INSTS_BRANCH"],
                     "lds": pmc["SQ_INSTS_LDS"], "smem": pmc["SQ_INSTS_SMEM"], "vmem_rd": pmc["SQ_INSTS_VMEM_RD"],
                     "instructions_per_wave_segment": (pmc["SQ_INSTS_VALU"] + others) / wseg if wseg else None,
                     "issue_slot_fraction": (2.0 * pmc["SQ_INSTS_VALU"] + others) / (1024 * kms * 1e-3 * 2.4e9) if kms > 0 else None,
                     "what": "(2 x vector + 1 x every other instruction) / (1024 SIMDs x kernel time x 2.4 GHz); s_nop / s_waitcnt not counted"}
        ref_eq = seg * n_tri * FLOP_PER_TEST / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        out = {
            "metric": "Msamples/sec (WxHxspp/wall) on Tor.obj 1080p",
            "value": samples_per_step * k / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / k * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (models/Tor.obj, 270 triangles, seed 42, counter RNG)" +
                    (" -- REHEARSAL: all ranks on one GPU, gloo gather; not a scaling measurement" if args.rehearse_on_one_gpu else ""),
            "config": {"workload": f"Tor.obj {W}x{H} x {args.spp} spp, -MRR {MRR}, -ERR -1 (adaptive off), -EPS 1e-4; "
                                   f"{world} row band(s) of {rows} rows" + (", one RCCL gather of 28 B/pixel to rank 0" if world > 1 else ""),
                       "width": W, "height": H, "spp": args.spp, "max_ray_reflections": MRR, "triangles": n_tri,
                       "parallelism": f"rowband{world}"},
            "roofline": {"bound": "valu_issue", "achieved": achieved, "peak": PEAK_VALU_TLANEOPS, "unit": "Tlane-op/s",
                         "frac": achieved / PEAK_VALU_TLANEOPS if achieved is not None else None,
                         "traffic": traffic,
                         "kernel": "pt::integrate_kernel<false,false,false,false>", "kernel_ms": kms,
                         "what": "achieved = executed VALU lane-operations (SQ_INSTS_VALU x 64) / live HIP-event kernel time; "
                                 "peak = 1024 SIMDs x 32 lanes x 2.4 GHz (no FMA: parity forbids contraction)",
                         "counters_source": pmc_source, "kernel_source_sha": kernel_source_sha(),
                         "valu_instructions_per_launch": pmc.get("SQ_INSTS_VALU") if pmc else None,
                         "valu_active_lane_fraction": lane_frac,
                         "issue": issue,
                         "segments_per_launch": seg, "exact_tests_per_segment": frame_stats["exact_tests"] / seg if seg else None,
                         "reference_equivalent_tflops": ref_eq, "reference_equivalent_over_fp32_peak": ref_eq / PEAK_FP32_VALU_TFLOPS,
                         "flop_per_test": FLOP_PER_TEST,
                         "hbm": {"algorithmic_bytes": algo_bytes, "traffic_bytes": traffic,
                                 "traffic_over_algorithmic": traffic / algo_bytes if traffic else None,
                                 "chunks_per_tile": n_chunks,
                                 "why": "each tile's accumulators are re-read and re-written once per pass-range chunk of the launch",
                                 "achieved": (traffic or algo_bytes) / (kms * 1e-3) / 1e9 if kms > 0 else 0.0,
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": ((traffic or algo_bytes) / (kms * 1e-3) / 1e9) / PEAK_HBM_GBS if kms > 0 else 0.0}},
        }
        if rccl_ranks_seen is not None:
            out["rccl_ranks_seen"] = rccl_ranks_seen
        if c3 is not None:
            out["configs3_strong"] = c3
        if world == 1:
            # the host-buffer boundary (pt_render_host): same frame, accumulators staged over PCIe both ways; not `value`
            acc = (np.zeros((npx, 3), np.float32), np.zeros((npx, 3), np.float32), np.zeros(npx, np.int32))
            scene.render_host(W, H, args.spp, MRR, accum=acc, want_stats=False)      # warm-up
            th = time.perf_counter()
            scene.render_host(W, H, args.spp, MRR, accum=acc, want_stats=False)
            th = time.perf_counter() - th
            out["pcie_inclusive"] = {"value": samples_per_step / th / 1e6, "unit": "Msamples/s", "ms_per_step": th * 1e3,
                                     "what": "pt_render_host: 116 MB of accumulators host->device and back around the same frame"}
        exe = os.path.join(ROOT, "path-tracing_amd", "bin", "pt_render")
        if world == 1 and os.path.exists(exe):
            # end to end: the stand-alone front end from process start to the BMP on disk -- HIP start-up,
            # OBJ/MTL load, table build, render, resolve (powf), BMP write.  A child process.
            with tempfile.TemporaryDirectory() as td:
                cmd = [exe, "--W", str(W), "--H", str(H), "-RPP", str(args.spp), "-MRR", str(MRR), "-ERR", "-1", "-UPDATE", "0",
                       "-QUIET", "1", "-SEED", "42", "-MODEL_PATH", models, "-OUT", os.path.join(td, "frame.bmp"), "-TIMING", "1"]
                te = time.perf_counter()
                r = subprocess.run(cmd, cwd=td, capture_output=True, text=True)
                te = time.perf_counter() - te
                ok = r.returncode == 0 and os.path.getsize(os.path.join(td, "frame.bmp")) == 54 + W * H * 3
                phases = None
                for line in r.stderr.splitlines():
                    if line.startswith("{") and "hip_startup_s" in line:
                        phases = json.loads(line)
            out["end_to_end"] = {"value": samples_per_step / te / 1e6 if ok else None, "unit": "Msamples/s", "seconds": te,
                                 "phases": phases,
                                 "what": "pt_render (C++ front end) as a child process: process start -> BMP on disk"}
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"], out["accuracy"] = cpu_baseline(models, args.cpu_seconds, pt, scene)
            out["accuracy"]["vs_reference_stream"] = reference_stream_accuracy(models, pt, scene)
        if args.write_bmp:
            parts = [t.cpu().numpy() for t in frame]
            s, s2, c = bands.assemble(parts, W, H, world)
            bgr, disp = pt.resolve(W, H, s, s2, c)
            pt.write_bmp(args.write_bmp, bgr)
            out["config"]["dispersion_max_min_avg"] = [float(d) for d in disp]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
