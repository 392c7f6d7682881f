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


# ---------------------------------------------------------------------------------------------------------------------
# The half-precision form of the same test (pt_kernels.hip: box_children_kept_h, PT_BOX_F16): two children per packed
# instruction, the child planes' bytes used AS THEY ARE as half-precision subnormals (q * 2^-24), the ray recentred on the point
# where it enters the node's frame and scaled by a power of two so that the frame's extent along the ray is below 2^-10.
# numpy restatement, operation for operation.
# ---------------------------------------------------------------------------------------------------------------------
H = np.float16
M_ALLOW = np.float32(2.0 ** -21 * (1.0 + 2.0 ** -6))   # both roundings of a comparison (half an ulp below 2^-10 each), in T units


def rtz_f16(x):
    """v_cvt_pkrtz_f16_f32: float32 -> float16 rounded toward zero (overflow saturates at the largest finite value; inf stays inf)."""
    x = np.asarray(x, np.float32)
    with np.errstate(over="ignore"):
        h = x.astype(np.float16)
    big = np.isfinite(x) & ~np.isfinite(h)
    h = np.where(big, np.copysign(np.float16(65504.0), x).astype(np.float16), h)
    over = np.abs(h.astype(np.float32)) > np.abs(x)
    toward0 = np.nextafter(h, np.float16(0.0))
    return np.where(over, toward0, h).astype(np.float16)


def ulp_up_magnitude(h):
    """the packed integer add of 1 to the bit pattern: the next half-precision value away from zero (the largest finite -> inf)."""
    b = np.asarray(h, np.float16).view(np.uint16)
    return (b + np.uint16(1)).astype(np.uint16).view(np.float16)


def fma16(a, q, b):
    """v_pk_fma_f16 on one half: the exact a * q + b rounded once to half precision (inf * 0 = NaN as in IEEE)."""
    with np.errstate(invalid="ignore", over="ignore"):
        return (np.asarray(a, np.float64) * np.asarray(q, np.float64) + np.asarray(b, np.float64)).astype(np.float16)


def max16(a, b):   # v_pk_max_f16 with a quiet NaN: the other operand
    return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.maximum(a, b)))


def min16(a, b):
    return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.minimum(a, b)))


def planes_f16(t, node, o, d, err=5e-7):
    """The half-precision planes of box_children_kept_h for (ray, node) items: dict with tn, tf [n, 3, 8] (float16: entry / exit
    plane of every child along every axis, in T units), t_enter, s (float32: T = (t - t_enter) s), m_t (the allowance that went
    into the entry planes) and the pieces the comparison needs."""
    n = len(node)
    f = np.float32
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        inv = (f(1) / d).astype(f)      # (the plain reciprocal, as in the kernel: a zero component gives infinite planes, NaNs, and keeps)
        step = t["step"][node][:, None]
        a = (step * inv).astype(f)
        b = ((t["org"][node] - o).astype(f) * inv).astype(f)
        far = fma(np.full((n, 3), f(255)), a, b)
        lo = np.minimum(b, far)                                   # the frame's near plane along each axis
        t_enter = lo.max(1)
        amin = np.abs(a).min(1)
        # the float32 stage's own rounding (reciprocal, products, differences), as in the float test but with half as much again:
        # the recentring is one more difference
        e2 = (f(3) * f(err) * fma(np.full(n, f(255)), np.abs(a).max(1), np.abs(b).max(1))).astype(f)
        # power of two S24 with amin * S24 in [2^5, 2^6): exponent arithmetic on the bits
        ebits = (amin.view(np.uint32) & np.uint32(0x7F800000)).astype(np.int64)
        s24_bits = np.int64(0x81800000) - ebits
        bad = (ebits == 0) | (ebits == 0x7F800000) | (s24_bits <= 0) | (s24_bits >= 0x7F800000)
        s24 = np.where(bad, 0, s24_bits).astype(np.uint32).view(f)
        s = (s24 * f(2.0 ** -24)).astype(f)
        m_t = fma(e2, s, np.full(n, M_ALLOW * f(err / 5e-7), f))
        ap = (np.abs(a) * s24[:, None]).astype(f)
        dk = (lo - t_enter[:, None]).astype(f)                    # <= 0
        bf = (dk * s[:, None]).astype(f)
        bn = fma(dk, np.broadcast_to(s[:, None], dk.shape), np.broadcast_to(-m_t[:, None], dk.shape))
        a_n = rtz_f16(ap)                                         # >= 0: rounded down
        a_f = ulp_up_magnitude(a_n)                               # rounded up
        b_f = rtz_f16(bf)                                         # <= 0: toward zero = up
        b_n = ulp_up_magnitude(rtz_f16(bn))                       # < 0: one more away from zero = down
        # child planes: near rows (lo bytes for a ray going up the axis, else 255 - hi), far rows likewise; as subnormals
        neg = (inv < 0)[:, :, None]
        lo_b, hi_b = t["lo"][node].astype(np.int64), t["hi"][node].astype(np.int64)     # [n, 3, 8]
        qn = np.where(neg, 255 - hi_b, lo_b).astype(np.float64) * 2.0 ** -24
        qf = np.where(neg, 255 - lo_b, hi_b).astype(np.float64) * 2.0 ** -24
        tn = fma16(a_n[:, :, None], qn, b_n[:, :, None])          # [n, 3, 8] half
        tf = fma16(a_f[:, :, None], qf, b_f[:, :, None])
    return {"tn": tn, "tf": tf, "t_enter": t_enter, "s": s, "m_t": m_t, "bad": bad, "neg": neg[:, :, 0], "inv": inv}


def children_kept_f16(t, node, o, d, t_best, stats=None, err=5e-7):
    """Kept-children masks [n_rays, 8] of the half-precision slab test."""
    n = len(node)
    f = np.float32
    p = planes_f16(t, node, o, d, err)
    tn, tf, t_enter, s, m_t, bad = p["tn"], p["tf"], p["t_enter"], p["s"], p["m_t"], p["bad"]
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        tmin_raw = fma(-t_enter, s, -m_t)
        tmin = rtz_f16(fma(np.abs(tmin_raw), np.full(n, f(-2.0 ** -9)), tmin_raw))
        tb_raw = ((t_best - t_enter).astype(f) * s).astype(f)
        tbest = rtz_f16(fma(np.abs(tb_raw), np.full(n, f(2.0 ** -9)), tb_raw))
        t_in = max16(max16(tn[:, 0], tn[:, 1]), max16(tn[:, 2], tmin[:, None]))
        t_out = min16(min16(tf[:, 0], tf[:, 1]), min16(tf[:, 2], tbest[:, None]))
        diff = (t_out.astype(np.float64) - t_in.astype(np.float64)).astype(np.float16)   # v_pk_add_f16 with a negated operand
        keep = ~np.signbit(diff) | np.isnan(diff)                 # a NaN (inf - inf) has its sign bit clear: kept
    keep = keep | bad[:, None]
    exists = np.arange(8)[None, :] < t["count"][node][:, None]
    if stats is not None:
        stats["bad"] = stats.get("bad", 0) + int(bad.sum())
    return keep & exists
