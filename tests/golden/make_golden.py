#!/usr/bin/env python3
"""Generates the committed golden fixtures in this directory.

The reference itself cannot be run in this image (GLM is absent), so these vectors come from
the CPU oracle (oracle/pt_oracle.c) AFTER it has been pinned to the reference's five recorded frames
(tests/test_oracle_known_answers.py).  They freeze the oracle's output so that neither the oracle nor the HIP path
can drift unnoticed, and they let the GPU suite check against data instead of against a second computation.

    python tests/golden/make_golden.py          # rewrites the .npz files next to this script

Files:
  tor_tables.npz          the 270 triangle records (plane, vertices, square as raw float32 bits), material indices,
                          material table -- what Scene::LoadModel + Triangle's constructor produce for models/Tor.obj
  tor_closest_hits.npz    4096 rays (origins, unit directions) taken from real path segments + their closest hit
                          (triangle index, distance bits); includes the misses found
  tor_frame_64x64x16.npz  accumulators (sum, sum2 bits, count) of 64x64, 16 spp, -MRR 8, -ERR -1, seed 42, counter RNG
  tor_frame_40x24x20_adaptive.npz   the same for 40x24, 20 spp, -ERR 0.001 (adaptive sampling on)
  reference_known_answers.json      what SURVEY.md 8(c) records for the reference itself (md5s, statistics, RNG draws)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402


def path_segment_rays(scene, n, seed):
    """Rays as the integrator produces them: camera rays and diffuse bounces off hit points."""
    rng = np.random.default_rng(seed)
    tri, _ = scene.triangles()
    o = np.tile(np.array([0, 0, -20], np.float32), (n, 1))
    d = np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n), np.ones(n)], 1).astype(np.float32)
    inv = np.float32(1) / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32)
    d = d * inv[:, None]
    out_o, out_d = [o.copy()], [d.copy()]
    for _ in range(3):
        idx, t, _ = scene.closest_hits(o, d)
        ok = idx >= 0
        p = o + d * np.where(ok, t, 0)[:, None]
        nrm = tri[np.maximum(idx, 0), 0:3]
        o = np.where(ok[:, None], p + nrm * np.float32(1e-4), o).astype(np.float32)
        r = rng.normal(size=(n, 3)).astype(np.float32)
        r = np.where(((r * nrm).sum(1) < 0)[:, None], -r, r)
        inv = np.float32(1) / np.sqrt((r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1]) + r[:, 2] * r[:, 2], dtype=np.float32)
        d = (r * inv[:, None]).astype(np.float32)
        out_o.append(o.copy()); out_d.append(d.copy())
    return np.concatenate(out_o), np.concatenate(out_d)


def main():
    sc = O.Scene.load(os.path.join(ROOT, "models") + "/", "Tor.obj")
    tri, mat = sc.triangles()
    np.savez_compressed(os.path.join(HERE, "tor_tables.npz"), triangles_bits=tri.view(np.uint32), triangle_material=mat,
                        materials_bits=sc.materials().view(np.uint32))
    o, d = path_segment_rays(sc, 1024, 123)
    idx, t, nan_seen = sc.closest_hits(o, d)
    assert not nan_seen.any()
    np.savez_compressed(os.path.join(HERE, "tor_closest_hits.npz"), origins_bits=o.view(np.uint32), directions_bits=d.view(np.uint32),
                        hit_index=idx, hit_t_bits=t.view(np.uint32))
    for name, (W, H, spp, mrr, err) in {"tor_frame_64x64x16": (64, 64, 16, 8, -1.0),
                                        "tor_frame_40x24x20_adaptive": (40, 24, 20, 8, 0.001)}.items():
        s, s2, c, st = O.render(sc, W, H, spp, mrr, error=err, seed=42, rng=O.RNG_COUNTER, trig=O.TRIG_PORTABLE)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), width=W, height=H, spp=spp, mrr=mrr, error=err, seed=42,
                            sum_bits=s.view(np.uint32), sum2_bits=s2.view(np.uint32), count=c,
                            segments=st["segments"], contributing=st["contributing"], misses=st["misses"],
                            samples_traced=st["samples_traced"])
    known = {
        "source": "SURVEY.md section 8(c): reference sources, THREADS_TO_RUN=1, seed 42, md5 of ../result.bmp",
        "frames": [
            {"args": "--H 256 --W 256 -RPP 4 -MRR 3", "md5": "ed4137839a531d82d4a6614ef3c12b13", "lit_pixels": 762,
             "max_disp": "0.175574", "min_disp": "0.000000", "aver_disp": "0.988375"},
            {"args": "--H 256 --W 256 -RPP 4 -MRR 8 -UPDATE 0", "md5": "a1cf8513956da7501050772509aa14b2",
             "max_disp": "0.641730", "aver_disp": "0.973576"},
            {"args": "--H 64 --W 64 -RPP 4 -MRR 3", "md5": "994782793a83d584cb8f0815a5a65b90", "aver_disp": "0.986328"},
            {"args": "--H 64 --W 64 -RPP 16 -MRR 8 -ERR -1 -UPDATE 0", "md5": "7706ad2c812da31a0ad8efb1837dcf43",
             "max_disp": "0.363144", "aver_disp": "0.912172"},
            {"args": "--H 64 --W 64 -RPP 16 -MRR 8", "md5": "cf4dc5e658211d1116880c0405e6ecd8", "aver_disp": "0.908428"}],
        "minstd_rand0_seed42_raw": [705894, 1126542223, 1579310009, 565444343],
        "Random_floats": ["0.000328707043", "0.524587095", "0.735423505", "0.263305545"],
        "jitter_doubles": ["0.024587101791753829", "-0.23669445921572174"],
        "assets_sha256": {"Tor.obj": "5356ff70022c8b1bde4607bfb9a25459b18bbb6ab48cd845ff0705ea41b3d4de",
                          "Tor.mtl": "f4a1c5b177900f3f20b6de0c055df973232a794f2ef975b23d999d3cd7c8a86c"}}
    json.dump(known, open(os.path.join(HERE, "reference_known_answers.json"), "w"), indent=1)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
