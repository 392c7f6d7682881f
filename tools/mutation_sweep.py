#!/usr/bin/env python3
"""Mutation sweep of the culling hierarchy's conservative margins (GPU box).

For every family of margins (pt_test_set_mutation in libpt_testhooks.so) and a ladder of scale factors, build the
tables with that family scaled and run a fixed parity workload against the CPU oracle: frames of Tor.obj, of a
replicated scene (deep-queue kernel + pair pre-filter) and of a synthetic quad scene, plus explicit adversarial rays.
Prints one JSON line per (family, scale): how many pixels / rays differ.  tests/test_gpu_mutation.py freezes the result:
the suite passes at 1.0 and notices each family at the scale recorded here.

    python tools/mutation_sweep.py > gpurun_out/mutation_sweep.jsonl
"""
import importlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_lib as O  # noqa: E402

pt = importlib.import_module("path-tracing_amd")

# single families, and the barycentric margins together (each of them alone is covered by the others' slack)
FAMILIES = [("sphere_r2",), ("box",), ("box_err",), ("m0",), ("k12",), ("a_max",), ("quad_slack",), ("m0", "k12", "quad_slack"),
            ("m0", "k12", "quad_slack", "a_max")]
LADDER = [1.0, 0.98, 0.9, 0.7, 0.4, 0.1, 0.0]


class Workload:
    """Scenes, oracle answers and rays, prepared once."""

    def __init__(self):
        import make_replicated_scene as M
        self.tmp = tempfile.mkdtemp() + "/"
        models = os.path.join(ROOT, "models") + "/"
        M.generate(os.path.join(ROOT, "models"), self.tmp, "x9.obj", 9)
        self.scenes = [(models, "Tor.obj", (96, 64, 8)), (self.tmp, "x9.obj", (64, 40, 4))]   # small-scene and box-tree paths
        self.oracle = []
        for d, name, (W, H, spp) in self.scenes:
            o = O.Scene.load(d, name)
            frame = O.render(o, W, H, spp, 8, error=-1.0)[:3]
            rng = np.random.default_rng(17)
            rays = self.rays(o, rng, 40_000)
            hits = o.closest_hits(*rays)
            self.oracle.append((o, frame, rays, hits))

    @staticmethod
    def rays(o, rng, n):
        """Path-like rays: from points just off random surfaces into random directions, plus camera rays."""
        tri, _ = o.triangles()
        T = len(tri)
        v = tri[:, 4:13].reshape(T, 3, 3).astype(np.float64)
        nrm = tri[:, 0:3].astype(np.float64)
        a = rng.integers(0, T, n)
        w = rng.dirichlet([1, 1, 1], n)
        src = (v[a] * w[:, :, None]).sum(1) + nrm[a] * 1e-4
        d = rng.normal(size=(n, 3))
        # three quarters of them aimed AT an edge or a vertex of another triangle -- exactly on it, or a hair to either
        # side -- where a margin that is too small bites first; large triangles (walls) are picked as often as small ones
        area = tri[:, 13].astype(np.float64)
        big = np.flatnonzero(area > 0.25 * area.max())
        b = np.where(rng.random(n) < 0.5, big[rng.integers(0, len(big), n)], rng.integers(0, T, n))
        e = rng.random((n, 1)) * (rng.random((n, 1)) < 0.9)          # 10 %: exactly a vertex
        k = rng.integers(0, 3, n)
        p0, p1, p2 = v[b, k], v[b, (k + 1) % 3], v[b, (k + 2) % 3]
        inward = (p2 - (p0 + p1) / 2)
        inward /= np.linalg.norm(inward, axis=1, keepdims=True) + 1e-30
        hair = rng.choice([0.0, 0.0, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5], n)[:, None]
        tgt = p0 * e + p1 * (1 - e) + inward * hair
        d = np.where((np.arange(n) % 4 == 0)[:, None], d, tgt - src)
        src = src.astype(np.float32)
        d = d.astype(np.float32)
        inv = np.float32(1) / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32)
        return src, (d * inv[:, None]).astype(np.float32)

    def run(self, lib):
        """Returns (differing pixels, differing rays) summed over the scenes for the library's current mutation."""
        bad_px = bad_rays = 0
        for (d, name, (W, H, spp)), (o, frame, rays, hits) in zip(self.scenes, self.oracle):
            g = pt.Scene.load_obj(d, name, device=0, library=lib)
            s, s2, c, _ = g.render_host(W, H, spp, 8, error=-1.0)
            bad_px += int(((s.view(np.uint32) != frame[0].view(np.uint32)).any(1) | (c != frame[2])).sum())
            gi, gt = g.trace_rays(*rays)
            ri, rt, nan_seen = hits
            bad_rays += int((((gi != ri) | (gt.view(np.uint32) != rt.view(np.uint32))) & ~nan_seen).sum())
            g.close()
        return bad_px, bad_rays


def main():
    lib = pt.load_library(pt.TESTHOOKS_LIB_PATH)
    w = Workload()
    for fams in FAMILIES:
        for scale in LADDER:
            lib.pt_test_set_mutation(b"reset", 0.0)
            for fam in fams:
                lib.pt_test_set_mutation(fam.encode(), scale)
            px, rays = w.run(lib)
            print(json.dumps({"family": "+".join(fams), "scale": scale, "differing_pixels": px, "differing_rays": rays}), flush=True)
    lib.pt_test_set_mutation(b"reset", 0.0)


if __name__ == "__main__":
    main()
