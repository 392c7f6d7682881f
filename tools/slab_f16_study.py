#!/usr/bin/env python3
"""What would a half-precision slab test of the box tree cost in extra work?  The shipped tree of a replicated scene, path-like
rays (camera rays and bounces off the oracle's hit points), walked on the CPU with the float32 test (tests/bvh_emulation.py:
children_kept) and with the half-precision one (children_kept_f16), both pruned with each ray's final hit distance: node visits,
child boxes kept and (ray, triangle) pairs per ray -- and a check that the half-precision test keeps everything the float32 one
keeps along the chain to the hit.  CPU only.    python tools/slab_f16_study.py [instances=64] [camera rays=4000]
"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_replicated_scene as M
import oracle_lib as O
import bvh_emulation as B
pt = importlib.import_module("path-tracing_amd")

inst = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
d = f"/tmp/f16study_x{inst}/"
os.makedirs(d, exist_ok=True)
M.generate(os.path.join(ROOT, "models"), d, "s.obj", inst)
g = pt.Scene.load_obj(d, "s.obj", device=-1)
o = O.Scene.load(d, "s.obj")
tri, _ = g.triangles()
lay = g.cull_layout()
t = B.decode(lay["bvh"])
rng = np.random.default_rng(1)
x = rng.random(NR) - 0.5; y = rng.random(NR) - 0.5
rd = np.stack([x, y, np.ones(NR)], 1); rd /= np.linalg.norm(rd, axis=1)[:, None]
ro = np.tile(np.array([0, 0, -20.0]), (NR, 1))
allo, alld, allt = [], [], []
for gen in range(5):
    idx, th, _ = o.closest_hits(ro.astype(np.float32), rd.astype(np.float32), threads=8)
    ok = idx >= 0
    allo.append(ro[ok]); alld.append(rd[ok]); allt.append(th[ok])
    P = ro[ok] + rd[ok] * th[ok][:, None]
    N = tri[idx[ok], 0:3].astype(np.float64)
    r = rng.normal(size=P.shape); r /= np.linalg.norm(r, axis=1)[:, None]
    r[(r * N).sum(1) < 0] *= -1
    r[(rd[ok] * N).sum(1) > 0] *= -1
    ro, rd = P + r * 1e-3, r
RO = np.concatenate(allo).astype(np.float32); RD = np.concatenate(alld).astype(np.float32); RT = np.concatenate(allt).astype(np.float32)
RD = (RD / np.sqrt((RD.astype(np.float64) ** 2).sum(1))[:, None]).astype(np.float32)
print(f"x{inst}: {len(tri)} triangles, {len(lay['bvh'])} nodes, {len(RO)} rays")


def walk(kept_fn):
    visits = np.zeros(len(RO)); kept_children = np.zeros(len(RO)); pairs = np.zeros(len(RO))
    front = [(np.zeros(len(RO), np.int64), np.arange(len(RO)))]      # (node per ray, ray index)
    while front:
        node, rays = front.pop()
        if len(rays) == 0:
            continue
        visits[rays] += 1
        k = kept_fn(t, node, RO[rays], RD[rays], RT[rays])
        kept_children[rays] += k.sum(1)
        leaf = t["leaf"][node]
        pairs[rays[leaf]] += k[leaf].sum(1)
        for c in range(8):
            sel = k[:, c] & ~leaf
            if sel.any():
                front.append((t["base"][node[sel]].astype(np.int64) + c, rays[sel]))
    return visits, kept_children, pairs


v32, k32, p32 = walk(lambda t_, n_, o_, d_, tb: B.children_kept(t_, n_, o_, d_, tb, 5e-7))
st = {}
v16, k16, p16 = walk(lambda t_, n_, o_, d_, tb: B.children_kept_f16(t_, n_, o_, d_, tb, st))
print(f"float32 test : visits/ray {v32.mean():.3f}  children kept/ray {k32.mean():.3f}  pairs/ray {p32.mean():.3f}")
print(f"half test    : visits/ray {v16.mean():.3f}  children kept/ray {k16.mean():.3f}  pairs/ray {p16.mean():.3f}   (nodes kept whole: {st.get('bad', 0)})")
print(f"ratio        : visits {v16.mean() / v32.mean():.4f}  pairs {p16.mean() / p32.mean():.4f}")
