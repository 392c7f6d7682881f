#!/usr/bin/env python3
"""Benchmark of the radiance-integrator hot path (BASELINE.json metric: Msamples/s = W*H*spp / wall on Tor.obj).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one complete frame: zero the accumulators, trace every sample of every pixel of this rank's row band
(all passes x segments x triangles in one kernel launch through the C ABI), and -- for N > 1 -- the single RCCL
gather of the accumulator bands to rank 0 (issued asynchronously: it overlaps the next frame's kernel, every gather is
complete before the timed region ends).  At N = 1 the frame is BASELINE.json configs[1]: models/Tor.obj,
1920x1080, 64 spp, -MRR 8, adaptive sampling off (-ERR -1, so all W*H*spp samples are traced).  For N > 1 the image
grows with N (path-tracing_amd/bands.py: frame_for) so that every GPU keeps a 1080p-sized band: weak scaling.

Rank 0 prints ONE JSON line.  Besides the contract's fields it carries
  roofline      FP32 vector-ALU roofline of the integrator kernel (the path is neither HBM- nor
                MFMA-bound), HIP-event kernel time measured live on the launch stream, plus the HBM view north_star asks for
  cpu_baseline  the CPU oracle (a port of the reference's algorithm) timed on this box's host cores on a bounded sample
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLOP_PER_TEST = 31.5          # reference's own average over its stage-exit mix
PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz
PEAK_HBM_GBS = 8000.0
BASE_W, BASE_H, SPP, MRR = 1920, 1080, 64, 8


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
    accuracy = {"vs": "cpu_baseline sample (same rows, passes, seed)",
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP, help="samples per pixel of the frame (BASELINE configs: 64 / 256 / 1024)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget; 0 skips it")
    ap.add_argument("--write-bmp", default="", help="resolve rank 0's gathered frame and write it here")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks all render on device 0 and gather over gloo (host copies): exercises the multi-rank "
                         "code path on a one-GPU box; the number it prints is NOT a scaling result")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    pt = importlib.import_module("path-tracing_amd")
    bands = importlib.import_module("path-tracing_amd.bands")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU with torch.distributed.run")
    if not torch.cuda.is_available() or pt.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the integrator has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm

    W, H = bands.frame_for(world)
    r0, r1 = bands.band_rows(H, world, rank)
    rows = r1 - r0
    npx = rows * W
    models = os.path.join(ROOT, "models") + "/"
    scene = pt.Scene.load_obj(models, "Tor.obj", device=local)
    n_tri = scene.counts()[0]

    # one contiguous band buffer: sum[3n] | sum2[3n] | count[n] (int32 bits) -> a single gather moves everything.
    # Two of them for N > 1: frame k is rendered into one while the gather of frame k-1 still reads the other, so the
    # collective (on RCCL's stream) overlaps the next frame's kernel instead of extending every step.
    n_band = 2 if world > 1 else 1
    band_bufs = [torch.zeros(bands.band_floats(W, rows), dtype=torch.float32, device=dev) for _ in range(n_band)]
    recv_bufs = [None] * n_band     # rank 0: receive buffers, one set per band buffer
    gathered = [None]
    in_flight = [None]              # (work handle, index of the receive set) of the gather not yet waited for
    frame_no = [0]
    band = band_bufs[0]
    params = pt.RenderParams(W, H, r0, r1, 0, args.spp, MRR, 1e-4, -1.0, 42)
    stream = torch.cuda.current_stream(dev)

    # Timed steps launch without pt_render_stats (the library then runs the kernel instantiation without its nine
    # diagnostic counters, as a caller that only wants the frame does) and are bracketed by HIP events on the launch
    # stream; the counters of the same deterministic frame (segments, chunks per tile) come from an untimed launch.
    events = []

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
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            scene.render_device(params, p_sum, p_sum2, p_cnt, stream=stream.cuda_stream, want_stats=False)
            e1.record(stream)
            events.append((e0, e1))
            st = None
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
    for _ in range(max(args.warmup, 1)):   # at least one untimed launch: it supplies the frame's counters
        frame_stats = step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    stats = [dict(frame_stats, kernel_ms=e0.elapsed_time(e1)) for e0, e1 in events] or [dict(frame_stats, kernel_ms=0.0)]
    if world > 1:
        rdev = torch.device("cpu") if args.rehearse_on_one_gpu else dev
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        agg = torch.tensor([sum(s["kernel_ms"] for s in stats), float(sum(s["segments"] for s in stats)),
                            float(sum(s["samples_traced"] for s in stats))], dtype=torch.float64, device=rdev)
        kmax = agg[:1].clone()
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        kernel_ms_rank0 = sum(s["kernel_ms"] for s in stats)
    else:
        kernel_ms_rank0 = sum(s["kernel_ms"] for s in stats)

    if rank == 0:
        k = max(args.steps, 1)
        samples_per_step = W * H * args.spp
        # dominant kernel on THIS rank: algorithmic flops per launch / HIP-event launch duration
        seg = sum(s["segments"] for s in stats) / k
        kms = kernel_ms_rank0 / k
        achieved = seg * n_tri * FLOP_PER_TEST / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        # accumulators are read + written once per pass-range chunk of a tile (the launch is cut into chunks for tail
        # balance, DESIGN.md "Scheduling"), scene tables once
        n_chunks = max(1, stats[0].get("n_chunks", 1))
        algo_bytes = npx * 28 * 2 * n_chunks + n_tri * 112
        traffic = valu_util = None
        # PMC summaries written by tools/summarize_pmc.py from rocprofv3 --pmc passes of this same command line
        import glob
        for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json"))) if world == 1 else []:
            j = json.load(open(pmc))
            if j.get("spp") == args.spp and j.get("width") == W and j.get("height") == H:
                traffic = j.get("hbm_bytes_per_launch", traffic)
                valu_util = j.get("valu_issue_utilisation", valu_util)
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
            # achieved = the REFERENCE's work (segments x triangles x 31.5 flop) per second of kernel time; the kernel culls
            # most ray-triangle pairs, so this can exceed the ALU peak -- valu_issue_utilisation_pmc is the hardware view
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": traffic,
                         "kernel": "pt::integrate_kernel<false,false,false>", "kernel_ms": kms,
                         "valu_issue_utilisation_pmc": valu_util, "segments_per_launch": seg,
                         "flop_per_test": FLOP_PER_TEST,
                         "hbm": {"algorithmic_bytes": algo_bytes, "chunks_per_tile": n_chunks, "achieved": algo_bytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0,
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": (algo_bytes / (kms * 1e-3) / 1e9) / PEAK_HBM_GBS if kms > 0 else 0.0}},
        }
        if world == 1:
            # the host-buffer boundary (pt_render_host): same frame, accumulators staged over PCIe both ways; not `value`
            import numpy as np
            acc = (np.zeros((npx, 3), np.float32), np.zeros((npx, 3), np.float32), np.zeros(npx, np.int32))
            scene.render_host(W, H, args.spp, MRR, accum=acc, want_stats=False)      # warm-up
            th = time.perf_counter()
            scene.render_host(W, H, args.spp, MRR, accum=acc, want_stats=False)
            th = time.perf_counter() - th
            out["pcie_inclusive"] = {"value": samples_per_step / th / 1e6, "unit": "Msamples/s", "ms_per_step": th * 1e3,
                                     "what": "pt_render_host: 116 MB of accumulators host->device and back (pageable memory) around the same launch"}
        exe = os.path.join(ROOT, "path-tracing_amd", "bin", "pt_render")
        if world == 1 and os.path.exists(exe):
            # end to end: the stand-alone front end from process start to the BMP on disk -- HIP start-up,
            # OBJ/MTL load, table build, render through pt_render_host, resolve (powf), BMP write.  A child process.
            import subprocess
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                cmd = [exe, "--W", str(W), "--H", str(H), "-RPP", str(args.spp), "-MRR", str(MRR), "-ERR", "-1", "-UPDATE", "0",
                       "-QUIET", "1", "-SEED", "42", "-MODEL_PATH", models, "-OUT", os.path.join(td, "frame.bmp")]
                te = time.perf_counter()
                r = subprocess.run(cmd, cwd=td, capture_output=True, text=True)
                te = time.perf_counter() - te
                ok = r.returncode == 0 and os.path.getsize(os.path.join(td, "frame.bmp")) == 54 + W * H * 3
            out["end_to_end"] = {"value": samples_per_step / te / 1e6 if ok else None, "unit": "Msamples/s", "seconds": te,
                                 "what": "pt_render (C++ front end) as a child process: process start -> BMP on disk"}
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"], out["accuracy"] = cpu_baseline(models, args.cpu_seconds, pt, scene)
        if args.write_bmp:
            parts = [t.cpu().numpy() for t in gathered[0]] if world > 1 else [band_bufs[0].cpu().numpy()]
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
