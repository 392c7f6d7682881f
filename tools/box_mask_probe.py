#!/usr/bin/env python3
"""Both device forms of the box tree's child test (float: box_children_kept; packed half precision: box_children_kept_h) on random
(node, ray, t_best) items, through the test build's pt_test_box_masks, against their numpy restatements (tests/bvh_emulation.py)
and against the exact test in float64: neither may drop a child the exact test keeps.    python tools/box_mask_probe.py [n] [library name: testhooks]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bvh_emulation as B
pt = importlib.import_module("path-tracing_amd")


def items(n, seed=3):
    rng = np.random.default_rng(seed)
    org = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
    e = rng.integers(-14, -1, n)
    lo_b = rng.integers(0, 250, (n, 3, 8))
    hi_b = np.minimum(255, lo_b + rng.integers(0, 120, (n, 3, 8)))
    count = rng.integers(1, 9, n)
    leaf = rng.random(n) < 0.5
    raw = np.zeros((n, 64), np.uint8)
    raw[:, :12] = org.view(np.uint8).reshape(n, 12)
    meta = ((e + 127).astype(np.uint32) | ((count - 1).astype(np.uint32) << 8) | (leaf.astype(np.uint32) << 11) | (np.uint32(5) << 12))
    raw[:, 12:16] = meta.view(np.uint8).reshape(n, 4)
    raw[:, 16:40] = lo_b.astype(np.uint8).reshape(n, 24)
    raw[:, 40:64] = hi_b.astype(np.uint8).reshape(n, 24)
    step = (2.0 ** e).astype(np.float64)
    centre = org.astype(np.float64) + 127.5 * step[:, None]
    dist = 10.0 ** rng.uniform(-3, 1.6, n) * (rng.random(n) < 0.85)
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
    o = (centre + u * dist[:, None] + rng.normal(size=(n, 3)) * 60 * step[:, None]).astype(np.float32)
    aim = centre + rng.uniform(-140, 140, (n, 3)) * step[:, None] - o
    aim[::9, 1] *= 1e-4
    aim[::31, 2] = 0.0                                   # exactly axis-parallel components
    d = aim / np.linalg.norm(aim, axis=1)[:, None]
    d = (d / np.sqrt((d * d).sum(1))[:, None]).astype(np.float32)
    t_hit = np.linalg.norm(centre - o, axis=1) * rng.uniform(0.3, 3.0, n)
    t_best = np.where(rng.random(n) < 0.3, np.inf, t_hit).astype(np.float32)
    return raw, o, d, t_best


def exact_keep(t, o, d, t_best):
    with np.errstate(divide="ignore", invalid="ignore"):
        o64, d64 = o.astype(np.float64), d.astype(np.float64)
        lo = t["org"].astype(np.float64)[:, :, None] + t["lo"].astype(np.float64) * t["step"].astype(np.float64)[:, None, None]
        hi = t["org"].astype(np.float64)[:, :, None] + t["hi"].astype(np.float64) * t["step"].astype(np.float64)[:, None, None]
        t0 = (lo - o64[:, :, None]) / d64[:, :, None]
        t1 = (hi - o64[:, :, None]) / d64[:, :, None]
        par = (d64 == 0)[:, :, None]
        inside = (o64[:, :, None] >= lo) & (o64[:, :, None] <= hi)
        tn = np.where(par, np.where(inside, -np.inf, np.inf), np.minimum(t0, t1)).max(1)
        tf = np.where(par, np.where(inside, np.inf, -np.inf), np.maximum(t0, t1)).min(1)
        keep = np.maximum(tn, 0) <= np.minimum(tf, t_best.astype(np.float64)[:, None])
    return keep & (np.arange(8)[None, :] < t["count"][:, None])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    raw, o, d, t_best = items(n)
    L = pt.load_library(os.path.join(ROOT, "path-tracing_amd", "lib", f"libpt_{sys.argv[2] if len(sys.argv) > 2 else 'testhooks'}.so"))
    L.pt_test_box_masks.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int32, C.POINTER(C.c_uint32)]
    rays = np.ascontiguousarray(np.concatenate([o, d], 1), np.float32)
    out = np.zeros(2 * n, np.uint32)
    rc = L.pt_test_box_masks(raw.ctypes.data_as(C.c_void_p), rays.ctypes.data_as(C.POINTER(C.c_float)), t_best.ctypes.data_as(C.POINTER(C.c_float)),
                             C.c_float(5e-7), n, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert rc == 0, L.pt_last_error()
    bits = lambda m: ((m[:, None] >> np.arange(8)[None, :]) & 1).astype(bool)
    dev32, dev16, devmix = bits(out[0::2]), bits(out[1::2] & 0xFF), bits(out[1::2] >> 8)
    t = B.decode(raw)
    node = np.arange(n)
    em32 = B.children_kept(t, node, o, d, t_best, 5e-7)
    em16 = B.children_kept_f16(t, node, o, d, t_best)
    ex = exact_keep(t, o, d, t_best)
    print(f"{n} items; kept children per item: exact {ex.sum(1).mean():.3f}  float (device) {dev32.sum(1).mean():.3f}  half (device) {dev16.sum(1).mean():.3f}  "
          f"float (numpy) {em32.sum(1).mean():.3f}  half (numpy) {em16.sum(1).mean():.3f}")
    print("device float drops a child the exact test keeps:", int((ex & ~dev32).sum()), " device half:", int((ex & ~dev16).sum()))
    print("mixed-precision form (v_fma_mix_f32) differs from the float form in", int((devmix != dev32).any(1).sum()), "items (must be 0)")
    print("device vs numpy, items that differ: float", int((dev32 != em32).any(1).sum()), " half", int((dev16 != em16).any(1).sum()))
    for c in range(8):
        print(f"  child {c}: half device keeps {dev16[:, c].mean():.3f}, numpy {em16[:, c].mean():.3f}, exact {ex[:, c].mean():.3f}, wrongly dropped {int((ex[:, c] & ~dev16[:, c]).sum())}")
    return 0 if (ex & ~dev16).sum() == 0 and (ex & ~dev32).sum() == 0 and (devmix == dev32).all() else 1


if __name__ == "__main__":
    sys.exit(main())
