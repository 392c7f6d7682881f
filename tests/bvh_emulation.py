"""numpy restatement of the kernel's box-tree cull (pt_kernels.hip: box_children_kept) -- test infrastructure.

Float32 arithmetic in the kernel's operation order (fma emulated as one rounding of the exact double result), used by
the CPU suite to check the CONSERVATIVE property of the tables without a GPU: the chain of nodes above the triangle the
reference hits must survive, for the tightest t_best the walk can ever hold (the hit's own distance).
"""
import numpy as np

F = np.float32


def decode(nodes):
    """nodes: uint8 [n, 64] as pt_scene_cull_layout returns them."""
    raw = np.ascontiguousarray(nodes)
    org = raw[:, :12].copy().view(np.float32).reshape(-1, 3)
    meta = raw[:, 12:16].copy().view(np.uint32).reshape(-1)
    step = ((meta & 0xFF).astype(np.uint32) << 23).view(np.float32)
    count = ((meta >> 8) & 7) + 1
    leaf = ((meta >> 11) & 1).astype(bool)
    base = meta >> 12          # inner node: first child node; leaf: first slot / 8
    lo = raw[:, 16:40].reshape(-1, 3, 8)
    hi = raw[:, 40:64].reshape(-1, 3, 8)
    return {"org": org, "step": step, "count": count, "base": base, "leaf": leaf, "lo": lo, "hi": hi}


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def children_kept(t, node, o, d, t_best, err):
    """Kept-children masks [n_rays, 8] of nodes `node` [n_rays] for rays (o, d) [n_rays, 3] float32."""
    # (the kernel takes the plain reciprocal: a zero component gives infinite planes and keeps every child; clamping it to
    # 1e-30 here models a cull that is at least as strict, so what this restatement keeps the kernel keeps as well)
    m = np.maximum(np.abs(d), F(1e-30))
    inv = (F(1) / np.copysign(m, d)).astype(np.float32)
    step = t["step"][node][:, None]
    a = (step * inv).astype(np.float32)
    b = ((t["org"][node] - o).astype(np.float32) * inv).astype(np.float32)
    e2 = F(2) * F(err) * fma(np.full(len(node), F(255)), np.abs(a).max(1), np.abs(b).max(1))
    lo = t["lo"][node].astype(np.float32)      # [n, 3, 8]
    hi = t["hi"][node].astype(np.float32)
    # the kernel picks the entry / exit plane of every slab by the sign of the reciprocal direction
    neg = (inv < 0)[:, :, None]
    near, far = np.where(neg, hi, lo), np.where(neg, lo, hi)
    nb = (b - e2[:, None]).astype(np.float32)      # the allowance goes into the entry planes, once per node
    t0 = fma(np.broadcast_to(a[:, :, None], lo.shape), near, np.broadcast_to(nb[:, :, None], lo.shape))
    t1 = fma(np.broadcast_to(a[:, :, None], hi.shape), far, np.broadcast_to(b[:, :, None], hi.shape))
    t_in = np.maximum(t0.max(1), -e2[:, None])
    t_out = np.minimum(t1.min(1), t_best[:, None])
    keep = ~(t_in > t_out)
    exists = np.arange(8)[None, :] < t["count"][node][:, None]
    return keep & exists


def parents(t, first_leaf):
    """parent node and child position of every node (root: -1)."""
    n = len(t["count"])
    par = np.full(n, -1, np.int64)
    pos = np.zeros(n, np.int64)
    for i in range(n):
        if t["leaf"][i]:
            continue
        for c in range(int(t["count"][i])):
            par[int(t["base"][i]) + c] = i
            pos[int(t["base"][i]) + c] = c
    return par, pos
