#!/usr/bin/env python3
"""What every device of an N-device node would have to render: kernel time of each band of a frame, one band after the other on ONE
GPU -- the contiguous row bands of rounds 1-3 and the interleaved split (every N-th tile row of 8 image rows) of round 4.

    python tools/band_balance_probe.py            ->  one JSON line per (frame, N, split): kernel ms per band, slowest, mean

A frame is as slow as its slowest band; slowest / mean is what the split costs a node of identical devices."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

pt = importlib.import_module("path-tracing_amd")
bands = importlib.import_module("path-tracing_amd.bands")


def main():
    sc = pt.Scene.load_obj(os.path.join(ROOT, "models") + "/", "Tor.obj", device=0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    for n in (2, 4, 8):
        for label, (W, H) in (("configs[3], strong", (3840, 2160)), ("weak", bands.frame_for(n))):
            buf = torch.zeros(7 * W * H, dtype=torch.float32, device=dev)
            for kind in ("contiguous", "interleaved"):
                per_band = []
                for r in range(n):
                    if kind == "contiguous":
                        r0, r1 = bands.band_rows(H, n, r)
                        stride, rows = 1, r1 - r0
                    else:
                        r0, r1, stride, rows = bands.split(H, n, r)
                    npx = rows * W
                    ptrs = (buf.data_ptr(), buf.data_ptr() + 12 * npx, buf.data_ptr() + 24 * npx)
                    p = pt.RenderParams(W, H, r0, r1, 0, 256, 8, 1e-4, -1.0, 42, 0, stride)
                    ms = []
                    for _ in range(3):
                        buf.zero_()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(stream)
                        sc.render_device(p, *ptrs, stream=stream.cuda_stream)
                        e1.record(stream)
                        torch.cuda.synchronize(dev)
                        ms.append(e0.elapsed_time(e1))
                    per_band.append(round(sorted(ms)[1], 2))
                mean = sum(per_band) / n
                print(json.dumps({"frame": f"{W}x{H}x256 ({label})", "devices": n, "split": kind, "kernel_ms_per_band": per_band,
                                  "slowest_ms": max(per_band), "mean_ms": round(mean, 2), "slowest_over_mean": round(max(per_band) / mean, 3)}), flush=True)
            del buf


if __name__ == "__main__":
    main()
