#!/usr/bin/env python3
"""Is the box tree of big scenes the lever?  Node visits, child tests and pairs per ray of the SHIPPED tree (decoded from
pt_scene_cull_layout) against a binary surface-area-heuristic tree collapsed to 8-wide nodes of variable depth, on path-like rays
(camera rays and bounces off the oracle's hit points), pruned with each ray's final hit distance.  CPU only (numpy + oracle).
profiles/r03_ab_logs.txt tree01.

    python tools/tree_prototype.py [instances=64] [camera rays=6000]
"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_replicated_scene as M
import oracle_lib as O
import bvh_emulation as B
pt = importlib.import_module("path-tracing_amd")

inst = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
d = f"/tmp/proto_x{inst}/"
os.makedirs(d, exist_ok=True)
M.generate(os.path.join(ROOT, "models"), d, "s.obj", inst)
g = pt.Scene.load_obj(d, "s.obj", device=-1)
o = O.Scene.load(d, "s.obj")
tri, _ = g.triangles()
lay = g.cull_layout()
st = lay["slot_triangle"]
nodes = B.decode(lay["bvh"]); fl = lay["bvh_inner_nodes"]
n_slots_bvh = (len(lay["bvh"]) - fl) * 8
small = st[:n_slots_bvh]; small = small[small >= 0]
print("triangles", len(tri), "in tree", len(small), "nodes", len(lay["bvh"]), "inner nodes", fl)
V = tri[:, 4:13].reshape(-1, 3, 3).astype(np.float64)
tlo, thi = V.min(1) - 1e-4, V.max(1) + 1e-4

# ---- rays: camera rays, then cosine-ish bounces off the hit points
rng = np.random.default_rng(1)
def camera(n):
    x = rng.random(n) - 0.5; y = rng.random(n) - 0.5
    dd = np.stack([x, y, np.ones(n)], 1); dd /= np.linalg.norm(dd, axis=1)[:, None]
    return np.tile(np.array([0, 0, -20.0]), (n, 1)), dd
ro, rd = camera(NR)
allo, alld, allt = [], [], []
for gen in range(4):
    idx, t, _ = o.closest_hits(ro.astype(np.float32), rd.astype(np.float32), threads=8)
    ok = idx >= 0
    allo.append(ro[ok]); alld.append(rd[ok]); allt.append(t[ok].astype(np.float64))
    P = ro[ok] + rd[ok] * t[ok][:, None]
    N = tri[idx[ok], 0:3].astype(np.float64)
    r = rng.normal(size=P.shape); r /= np.linalg.norm(r, axis=1)[:, None]
    flip = (r * N).sum(1) < 0
    r[flip] *= -1
    # reference normals are not flipped toward the ray; bounce into the side the incoming ray came from
    inc = (rd[ok] * N).sum(1) > 0
    r[inc] *= -1
    ro, rd = P + r * 1e-3, r
RO, RD, RT = np.concatenate(allo), np.concatenate(alld), np.concatenate(allt)
print("rays", len(RO))

def slab(lo, hi, ro, rd, tb):
    inv = 1.0 / np.where(np.abs(rd) < 1e-30, 1e-30, rd)
    t0 = (lo - ro) * inv; t1 = (hi - ro) * inv
    tn = np.minimum(t0, t1).max(-1); tf = np.maximum(t0, t1).min(-1)
    return (np.maximum(tn, 0) <= np.minimum(tf, tb))

class Tree:  # generic wide tree: children lists
    def __init__(self):
        self.child = []   # per node: list of (is_leaf_tri, index, lo, hi)
def walk(tree, root, quant=True):
    """returns node visits per ray, leaf-child hits (pairs) per ray"""
    visits = np.zeros(len(RO)); pairs = np.zeros(len(RO)); tests = np.zeros(len(RO))
    stack = [(root, np.arange(len(RO)))]
    while stack:
        n, rays = stack.pop()
        if len(rays) == 0: continue
        visits[rays] += 1
        ch = tree.child[n]
        tests[rays] += len(ch)
        los = np.array([c[2] for c in ch]); his = np.array([c[3] for c in ch])
        if quant:
            nlo = los.min(0); ext = (his.max(0) - nlo).max()
            e = np.ceil(np.log2(max(ext, 1e-30) / 255.0)); step = 2.0 ** e
            los = nlo + np.floor((los - nlo) / step) * step
            his = nlo + np.ceil((his - nlo) / step) * step
        for c, lo, hi in zip(ch, los, his):
            k = slab(lo, hi, RO[rays], RD[rays], RT[rays] * (1 + 1e-6))
            if c[0]: pairs[rays[k]] += c[1]
            else: stack.append((c[1], rays[k]))
    return visits, pairs, tests

# ---- shipped tree
T0 = Tree()
nn = len(lay["bvh"])
for n in range(nn):
    ch = []
    for c in range(int(nodes["count"][n])):
        lo = nodes["org"][n].astype(np.float64) + nodes["lo"][n][:, c].astype(np.float64) * float(nodes["step"][n])
        hi = nodes["org"][n].astype(np.float64) + nodes["hi"][n][:, c].astype(np.float64) * float(nodes["step"][n])
        if nodes["leaf"][n]:
            s = int(nodes["base"][n]) * 8 + c
            if st[s] < 0: continue
            ch.append((True, 1, lo, hi))
        else:
            ch.append((False, int(nodes["base"][n]) + c, lo, hi))
    T0.child.append(ch)
v, p, te = walk(T0, 0, quant=False)
print(f"shipped tree: visits/ray {v.mean():.2f}  pairs/ray {p.mean():.2f}  child tests/ray {te.mean():.1f}  max visits {v.max():.0f}")

# ---- binary SAH (binned), then collapse to 8-wide
cen = (tlo + thi) / 2
def area(lo, hi):
    e = np.maximum(hi - lo, 0); return 2 * (e[..., 0] * e[..., 1] + e[..., 1] * e[..., 2] + e[..., 2] * e[..., 0])
class BNode: pass
def build_bin(ids, leaf_max):
    n = BNode(); n.ids = ids; n.lo = tlo[ids].min(0); n.hi = thi[ids].max(0); n.l = n.r = None
    if len(ids) <= leaf_max: return n
    best = (np.inf, None, None)
    for ax in range(3):
        order = ids[np.argsort(cen[ids, ax], kind="stable")]
        lo_acc = np.minimum.accumulate(tlo[order], 0); hi_acc = np.maximum.accumulate(thi[order], 0)
        lo_rev = np.minimum.accumulate(tlo[order][::-1], 0)[::-1]; hi_rev = np.maximum.accumulate(thi[order][::-1], 0)[::-1]
        k = np.arange(1, len(ids))
        cost = area(lo_acc[:-1], hi_acc[:-1]) * k + area(lo_rev[1:], hi_rev[1:]) * (len(ids) - k)
        if SAHQ:   # prefer cuts at multiples of leaf_max slightly (full leaves)
            pass
        j = int(np.argmin(cost))
        if cost[j] < best[0]: best = (cost[j], order, j + 1)
    _, order, cut = best
    n.l = build_bin(order[:cut], leaf_max); n.r = build_bin(order[cut:], leaf_max)
    return n
SAHQ = False
sys.setrecursionlimit(100000)
for leaf_max in (8, 16):
    t0 = time.time()
    root = build_bin(np.array(sorted(small)), leaf_max)
    # collapse: every wide node takes the binary node's children and keeps replacing the child of largest area by its two
    # children until it has 8 (leaves of the binary tree stay leaves = wide leaf nodes with their triangles as children)
    W = Tree()
    def collapse(bn):
        me = len(W.child); W.child.append(None)
        if bn.l is None:
            if leaf_max > 8:   # wide leaves: 8 child boxes of (up to) leaf_max / 8 triangles each, neighbours along the leaf's longest axis
                ids = np.array(bn.ids); ax = int(np.argmax(bn.hi - bn.lo)); ids = ids[np.argsort(cen[ids, ax], kind="stable")]
                per = (len(ids) + 7) // 8
                W.child[me] = [(True, len(g), tlo[g].min(0), thi[g].max(0)) for g in (ids[i:i + per] for i in range(0, len(ids), per))]
            else:
                W.child[me] = [(True, 1, tlo[t], thi[t]) for t in bn.ids]
            return me
        kids = [bn.l, bn.r]
        while len(kids) < 8:
            cand = [(area(k.lo, k.hi), i) for i, k in enumerate(kids) if k.l is not None]
            if not cand: break
            _, i = max(cand)
            k = kids.pop(i); kids += [k.l, k.r]
        W.child[me] = [(False, collapse(k), k.lo, k.hi) for k in kids]
        return me
    collapse(root)
    nl = sum(1 for c in W.child if c and c[0][0])
    v, p, te = walk(W, 0)
    depth = 0
    print(f"binary SAH leaf<={leaf_max} -> 8-wide: nodes {len(W.child)} (leaves {nl}) build {time.time()-t0:.1f}s: visits/ray {v.mean():.2f}  pairs/ray {p.mean():.2f}  "
          f"child tests/ray {te.mean():.1f}  max visits {v.max():.0f}")
