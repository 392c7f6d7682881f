"""The float-error constants behind the culling margins (pt_scene.cpp: tri_geometry, build_cull_tables, build_bvh), each
written as the inequality it stands for and checked on random triangles / rays: float32 arithmetic in the operation order
of the code it bounds, against float64 on the SAME float32 inputs.  CPU only.

    e_fp      = 48 u diam^2          error of the reference's |S - s1 - s2 - s3| near the triangle        (triangles.h:55-68)
    eps_line  = 8 u (d_max + r_org)  rounding of P* = o + d t* per component                               (triangles.h:55)
    disc_err  = 24 u d_max^2         error of the kernel's |m|^2 - (m.d)^2                                 (pt_kernels.hip: sphere_keep)
    k1, k2                           |t_cull - t_reference| <= (k2 + k1 |t|) / |n.d|                        (cull_reject)
    bvh_err   = 5e-7                 |computed slab t - exact slab t| <= bvh_err (|B| + 255 |A|)            (box_children_kept)
    half-precision planes            entry plane <= exact <= exit plane in the recentred, scaled T of box_children_kept_h   (-DPT_BOX_F16=1)
    m0, a_max                        a point the reference accepts has every barycentric >= -(m0 + a_max e_t), random walls

u = 2^-24.  Every check reports how much of the bound the worst sample used, so a margin that is merely lucky shows up.
"""
import ctypes as C
import importlib

import numpy as np

import oracle_lib as O

pt = importlib.import_module("path-tracing_amd")
U = 2.0 ** -24
F = np.float32


def f32(x):
    return np.asarray(x, np.float32)


def fma32(a, b, c):
    """One rounding of the exact a*b + c (products of float32 are exact in float64; the sum's double rounding is below 2^-53)."""
    return (f32(a).astype(np.float64) * f32(b).astype(np.float64) + f32(c).astype(np.float64)).astype(np.float32)


def random_triangles(rng, n, lo, hi):
    c = rng.uniform(-9, 9, (n, 1, 3))
    v = c + rng.normal(size=(n, 3, 3)) * rng.uniform(lo, hi, (n, 1, 1))
    return f32(v)


def glm_cross(a, b):   # glm::cross, float32, GLM's operand order
    return np.stack([a[:, 1] * b[:, 2] - b[:, 1] * a[:, 2], a[:, 2] * b[:, 0] - b[:, 2] * a[:, 0], a[:, 0] * b[:, 1] - b[:, 0] * a[:, 1]], 1).astype(np.float32)


def glm_length(a):
    return np.sqrt((a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1]) + a[:, 2] * a[:, 2], dtype=np.float32)


def test_area_sum_error_is_within_e_fp():
    """tri_geometry: E_fp = 48 u (diam + 1e-3)^2 bounds the float32 error of S - s1 - s2 - s3 for points within a
    triangle's neighbourhood (the only points for which the reference's last test can pass)."""
    rng = np.random.default_rng(1)
    n = 200_000
    v = random_triangles(rng, n, 0.05, 12.0)
    w = rng.dirichlet([1, 1, 1], n) * rng.choice([1.0, 1.0, 1.05], n)[:, None]
    p = f32((v.astype(np.float64) * w[:, :, None]).sum(1) + rng.normal(size=(n, 3)) * 1e-3)
    S = glm_length(glm_cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]))
    f = [f32(p - v[:, k]) for k in range(3)]
    s1, s2, s3 = glm_length(glm_cross(f[0], f[1])), glm_length(glm_cross(f[0], f[2])), glm_length(glm_cross(f[2], f[1]))
    got = (((S - s1) - s2) - s3).astype(np.float64)
    f64 = [x.astype(np.float64) for x in f]
    S64 = np.linalg.norm(np.cross(v[:, 1].astype(np.float64) - v[:, 0], v[:, 2].astype(np.float64) - v[:, 0]), axis=1)
    ref = S64 - np.linalg.norm(np.cross(f64[0], f64[1]), axis=1) - np.linalg.norm(np.cross(f64[0], f64[2]), axis=1) - np.linalg.norm(np.cross(f64[2], f64[1]), axis=1)
    e = v.astype(np.float64)
    diam = np.max([np.linalg.norm(e[:, 1] - e[:, 0], axis=1), np.linalg.norm(e[:, 2] - e[:, 0], axis=1), np.linalg.norm(e[:, 2] - e[:, 1], axis=1)], 0)
    used = np.abs(got - ref) / (48 * U * (diam + 1e-3) ** 2)
    print("e_fp: worst sample uses", used.max(), "of the bound")
    assert used.max() < 1.0


def test_hit_point_rounding_is_within_eps_line():
    """build_cull_tables: eps_line = 8 u (d_max + r_org) bounds |fl(o + fl(d t)) - (o + d t)| per component."""
    rng = np.random.default_rng(2)
    n = 500_000
    r_org = 21.0
    d_max = 2 * np.sqrt(3) * r_org
    o = f32(rng.uniform(-r_org, r_org, (n, 3)))
    d = rng.normal(size=(n, 3))
    d = f32(d / np.linalg.norm(d, axis=1, keepdims=True))
    t = f32(rng.uniform(0, d_max, n))
    got = (o + (d * t[:, None]).astype(np.float32)).astype(np.float32).astype(np.float64)
    ref = o.astype(np.float64) + d.astype(np.float64) * t.astype(np.float64)[:, None]
    used = np.abs(got - ref).max() / (8 * U * (d_max + r_org))
    print("eps_line: worst sample uses", used, "of the bound")
    assert used < 1.0


def test_sphere_discriminant_error_is_within_disc_err():
    """sphere_keep computes disc = |m|^2 - max(m.d, 0)^2 with fma chains; disc_err = 24 u d_max^2 bounds its error
    (including |d| differing from 1 by a few ulp, which scales (m.d)^2)."""
    rng = np.random.default_rng(3)
    n = 500_000
    r_org = 21.0
    d_max = 2 * np.sqrt(3) * r_org
    c = f32(rng.uniform(-r_org, r_org, (n, 3)))
    o = f32(rng.uniform(-r_org, r_org, (n, 3)))
    d = rng.normal(size=(n, 3)).astype(np.float32)
    inv = F(1) / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32)
    d = (d * inv[:, None]).astype(np.float32)          # Ray's normalisation (ray.h:23)
    m = (c - o).astype(np.float32)
    b = fma32(m[:, 0], d[:, 0], fma32(m[:, 1], d[:, 1], m[:, 2] * d[:, 2]))
    m2 = fma32(m[:, 0], m[:, 0], fma32(m[:, 1], m[:, 1], m[:, 2] * m[:, 2]))
    bb = np.maximum(b, F(0))
    got = fma32(-bb, bb, m2).astype(np.float64)
    m64, d64 = c.astype(np.float64) - o.astype(np.float64), d.astype(np.float64)
    d64 /= np.linalg.norm(d64, axis=1, keepdims=True)           # the exact unit direction the geometry is about
    bt = np.maximum((m64 * d64).sum(1), 0)
    ref = (m64 * m64).sum(1) - bt * bt
    used = np.abs(got - ref).max() / (24 * U * d_max ** 2)
    print("disc_err: worst sample uses", used, "of the bound")
    assert used < 1.0


def test_plane_distance_error_is_within_k1_k2():
    """cull_reject / cull_reject_quad: t_cull = -(fma chain) * rcp(fma chain); the reference's t = -(o.n + w) / (d.n) in
    plain float32.  |t_cull - t_ref| <= (k2 + k1 |t|) / |n.d| with k1 = 40 u, k2 = 12 u m_abs + 8 u r_org,
    m_abs = 2 sqrt(3) r_org; rcp is given a full ulp of error either way."""
    rng = np.random.default_rng(4)
    n = 400_000
    r_org = 21.0
    k1, k2 = 40 * U, 12 * U * 2 * np.sqrt(3) * r_org + 8 * U * r_org
    nn = rng.normal(size=(n, 3))
    nn = f32(nn / np.linalg.norm(nn, axis=1, keepdims=True))
    w = f32(rng.uniform(-r_org, r_org, n))
    o = f32(rng.uniform(-r_org, r_org, (n, 3)))
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d = (d * (F(1) / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32))[:, None]).astype(np.float32)
    num = fma32(o[:, 0], nn[:, 0], fma32(o[:, 1], nn[:, 1], fma32(o[:, 2], nn[:, 2], w)))
    den = fma32(d[:, 0], nn[:, 0], fma32(d[:, 1], nn[:, 1], d[:, 2] * nn[:, 2]))
    worst = 0.0
    for ulp in (-1, 0, 1):                                   # v_rcp_f32: 1 ulp
        rden = (F(1) / den).astype(np.float32)
        rden = np.nextafter(rden, np.where(ulp > 0, np.inf, -np.inf).astype(np.float32)) if ulp else rden
        t_cull = (-num * rden).astype(np.float32).astype(np.float64)
        # the reference: PlaneIntersect (triangles.h:10-13), GLM dot order, IEEE divide
        sd = ((d[:, 0] * nn[:, 0] + d[:, 1] * nn[:, 1]) + d[:, 2] * nn[:, 2]).astype(np.float32)
        t_ref = (-(((o[:, 0] * nn[:, 0] + o[:, 1] * nn[:, 1]) + o[:, 2] * nn[:, 2]) + w) / sd).astype(np.float32).astype(np.float64)
        ok = np.isfinite(t_ref) & (np.abs(t_ref) < 4096 * r_org) & (np.abs(sd) > 1e-6)      # beyond t_guard the cull abstains
        bound = (k2 + k1 * np.abs(t_cull)) / np.abs(den.astype(np.float64))
        worst = max(worst, float((np.abs(t_cull - t_ref)[ok] / bound[ok]).max()))
    print("k1/k2: worst sample uses", worst, "of the bound")
    assert worst < 1.0


def test_slab_arithmetic_error_is_within_bvh_err():
    """box_children_kept: t = fma(A, q, B), A = step * rcp(d), B = (org - o) * rcp(d) against the exact
    (org + q step - o) / d: |error| <= bvh_err (|B| + 255 |A|) with bvh_err = 5e-7 (rcp given a full ulp); the entry
    planes use B - 2E (the allowance subtracted once per node), whose rounding is part of the same budget."""
    rng = np.random.default_rng(5)
    n = 400_000
    org = f32(rng.uniform(-21, 21, n))
    o = f32(rng.uniform(-21, 21, n))
    d = f32(rng.uniform(-1, 1, n))
    d = np.where(np.abs(d) < 1e-30, F(1e-30), d)
    step = f32(2.0 ** rng.integers(-12, 0, n))
    q = f32(rng.integers(0, 256, n))
    worst = 0.0
    for ulp in (-1, 0, 1):
        inv = (F(1) / d).astype(np.float32)
        inv = np.nextafter(inv, np.where(ulp > 0, np.inf, -np.inf).astype(np.float32)) if ulp else inv
        A = (step * inv).astype(np.float32)
        B = ((org - o).astype(np.float32) * inv).astype(np.float32)
        ref = (org.astype(np.float64) + q.astype(np.float64) * step.astype(np.float64) - o.astype(np.float64)) / d.astype(np.float64)
        bound = 5e-7 * (np.abs(B.astype(np.float64)) + 255 * np.abs(A.astype(np.float64)))
        e2 = (F(2) * F(5e-7) * fma32(np.full(n, F(255)), np.abs(A), np.abs(B))).astype(np.float32)
        got_exit = fma32(A, q, B).astype(np.float64)
        got_entry = fma32(A, q, (B - e2).astype(np.float32)).astype(np.float64) + e2.astype(np.float64)   # entry plane, allowance added back
        worst = max(worst, float((np.abs(got_exit - ref) / bound).max()), float((np.abs(got_entry - ref) / bound).max()))
    print("bvh_err: worst sample uses", worst, "of the bound")
    assert worst < 1.0


def test_random_walls_accepted_points_satisfy_the_barycentric_margin():
    """Random LARGE triangles (the class culled by barycentric records): whenever the reference's Triangle::Intersect
    accepts a (ray, triangle) pair, the cull's inequalities hold in the kernel's own float32 arithmetic:
    min barycentric >= -(m0 + a_max e_t) and t >= -e_t."""
    rng = np.random.default_rng(6)
    T = 40
    v = f32(rng.uniform(-9.5, 9.5, (T, 1, 3)) * 0.2 + rng.normal(size=(T, 3, 3)) * 6.0)
    v = np.clip(v, -9.9, 9.9).astype(np.float32)
    tri = np.zeros((T, 14), np.float32)
    tri[:, 4:13] = v.reshape(T, 9)
    e1, e2 = (v[:, 1] - v[:, 0]).astype(np.float32), (v[:, 2] - v[:, 0]).astype(np.float32)
    nrm = glm_cross(e1, e2)
    S = glm_length(nrm)
    inv = (F(1) / np.sqrt((nrm[:, 0] * nrm[:, 0] + nrm[:, 1] * nrm[:, 1]) + nrm[:, 2] * nrm[:, 2], dtype=np.float32))
    nh = (nrm * inv[:, None]).astype(np.float32)
    tri[:, 0:3] = nh
    tri[:, 3] = -((nh[:, 0] * v[:, 0, 0] + nh[:, 1] * v[:, 0, 1]) + nh[:, 2] * v[:, 0, 2])      # Triangle ctor (triangles.h:40-44)
    tri[:, 13] = S
    mats = np.array([[0.5] * 3 + [0] * 3 + [0.5] * 3 + [10]], np.float32)
    g = pt.Scene.create(tri, np.zeros(T, np.int32), mats, device=-1)
    o_scene = O.Scene.from_arrays(tri, np.zeros(T, np.int32), mats)
    t = g.cull_tables()
    st = g.cull_layout()["slot_triangle"]
    large = np.flatnonzero(t["kind"] == 1)
    assert len(large) == 1 and t["n_tri"][large[0]] >= 30            # nearly all of them are "walls"
    c = large[0]
    first, off, qmask = t["first_tri"][c], t["data_off"][c], int(t["level_off"][c][1])
    assert qmask == 0                                                  # random triangles share no plane: single records
    k1, k2, a_max, m0 = (F(t["constants"][k]) for k in ("k1", "k2", "a_max", "m0"))
    L = O.lib()
    best = C.c_float()
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    checked, worst = 0, 0.0
    for k in range(t["n_tri"][c]):
        ti = int(st[first + k])
        rec = t["bary"][off + k]
        if np.isnan(rec[4]):
            continue                                                   # degenerate: never culled
        tv = v[ti].astype(np.float64)
        n_s = 1500
        ee = rng.random((n_s, 1)) * (rng.random((n_s, 1)) < 0.9)
        kk = rng.integers(0, 3, n_s)
        on_edge = tv[kk] * ee + tv[(kk + 1) % 3] * (1 - ee)
        inside = (tv[None] * rng.dirichlet([1, 1, 1], n_s)[:, :, None]).sum(1)
        tgt = np.where((np.arange(n_s) % 3 == 0)[:, None], inside, on_edge)
        org = f32(rng.uniform(-9.5, 9.5, (n_s, 3)))
        d = f32(tgt - org)
        d = (d * (F(1) / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32))[:, None]).astype(np.float32)
        acc = np.array([L.orc_probe_intersect(o_scene.h, ti, fp(org[i]), fp(d[i]), 1e-4, F(np.inf), C.byref(best)) == 4 for i in range(n_s)])
        if not acc.any():
            continue
        o_, d_ = org[acc], d[acc]
        num = fma32(o_[:, 0], rec[0], fma32(o_[:, 1], rec[1], fma32(o_[:, 2], rec[2], rec[3])))
        den = fma32(d_[:, 0], rec[0], fma32(d_[:, 1], rec[1], d_[:, 2] * rec[2]))
        rden = (F(1) / den).astype(np.float32)
        tt = (-num * rden).astype(np.float32)
        p = [fma32(tt, d_[:, x], o_[:, x]) for x in range(3)]
        uu = fma32(p[0], rec[4], fma32(p[1], rec[5], fma32(p[2], rec[6], rec[7])))
        vv = fma32(p[0], rec[8], fma32(p[1], rec[9], fma32(p[2], rec[10], rec[11])))
        ww = ((F(1) - uu) - vv).astype(np.float32)
        e_min = np.minimum(np.minimum(uu, vv), ww)
        et = (fma32(np.full(len(tt), k1), np.abs(tt), np.full(len(tt), k2)) * np.abs(rden)).astype(np.float32)
        mg = fma32(np.full(len(tt), a_max), et, np.full(len(tt), m0))
        assert (e_min >= -mg).all() and (tt >= -et).all(), (ti, float((e_min + mg).min()))
        worst = max(worst, float((-e_min / mg).max()))
        checked += int(acc.sum())
    print("barycentric margin: worst accepted sample uses", worst, "of it;", checked, "accepted pairs")
    assert checked > 10000


def test_half_precision_planes_bracket_the_exact_ones():
    """box_children_kept_h (the packed half-precision slab test, -DPT_BOX_F16=1; tests/bvh_emulation.py: planes_f16 restates it
    operation for operation).  The inequality it stands on: in T = (t - t_enter) S every ENTRY plane it computes lies below the
    exact plane by at least what any EXIT plane of the same node lies below its exact plane (the exit planes carry no allowance:
    both sides' errors are charged to the entry planes) -- for planes inside the window that decides, |T| < 2^-10 (beyond it only
    the order of magnitude matters: DESIGN.md section 5) -- so that exact entry <= exact exit implies computed entry <= computed
    exit: no child the exact test keeps is dropped.  `exact` is (org + q step - o) / d in float64 on the same float32
    inputs and t_enter, S are the kernel's own (a shift and a scale common to everything it compares).  Random frames, rays
    from all around them, with the reciprocal an ulp off either way; reports how much of the entry planes' allowance
    (float32 stage: 3 err (|B| + 255 |A|) S; two half-precision roundings: 2^-21 (1 + 2^-6)) the worst sample used."""
    import bvh_emulation as B
    rng = np.random.default_rng(11)
    n = 60_000
    # one synthetic node per item: random origin, a power-of-two step, eight random child boxes
    org = f32(rng.uniform(-20, 20, (n, 3)))
    step = f32(2.0 ** rng.integers(-14, -1, n))
    lo_b = rng.integers(0, 250, (n, 3, 8))
    hi_b = np.minimum(255, lo_b + rng.integers(0, 120, (n, 3, 8)))
    t = {"org": org, "step": step, "lo": lo_b.astype(np.uint8), "hi": hi_b.astype(np.uint8), "count": np.full(n, 8), "leaf": np.zeros(n, bool),
         "base": np.zeros(n, np.uint32)}
    node = np.arange(n)
    centre = org.astype(np.float64) + 127.5 * step[:, None].astype(np.float64)
    # ray origins: inside the frame, near it, far from it; directions: random, some nearly axis-parallel
    dist = 10.0 ** rng.uniform(-3, 1.6, n) * (rng.random(n) < 0.85)
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
    o = f32(centre + u * dist[:, None] + rng.normal(size=(n, 3)) * 60 * step[:, None])
    aim = centre + rng.uniform(-140, 140, (n, 3)) * step[:, None] - o
    aim[::9, rng.integers(0, 3)] *= 1e-4
    d = aim / np.linalg.norm(aim, axis=1)[:, None]
    d = f32(d / np.sqrt((d * d).sum(1))[:, None])
    used_worst, checked = 0.0, 0
    for ulp in (-1, 0, 1):
        # the kernel's reciprocal is v_rcp_f32, an ulp off at worst: emulate by perturbing the direction so that 1/d moves an ulp
        dd = d if ulp == 0 else np.nextafter(d, np.where((ulp > 0) == (d > 0), 0, np.sign(d) * np.inf).astype(np.float32))
        p = B.planes_f16(t, node, o, dd)
        ok = ~p["bad"] & np.isfinite(p["t_enter"])
        te, S = p["t_enter"].astype(np.float64)[:, None, None], p["s"].astype(np.float64)[:, None, None]
        inv64 = 1.0 / d.astype(np.float64)          # the EXACT ray: the unperturbed direction
        neg = (inv64 < 0)[:, :, None]
        near_q = np.where(neg, hi_b, lo_b).astype(np.float64)
        far_q = np.where(neg, lo_b, hi_b).astype(np.float64)
        base = (org.astype(np.float64) - o.astype(np.float64))[:, :, None]
        exact_n = ((base + near_q * step[:, None, None].astype(np.float64)) * inv64[:, :, None] - te) * S
        exact_f = ((base + far_q * step[:, None, None].astype(np.float64)) * inv64[:, :, None] - te) * S
        tn, tf = p["tn"].astype(np.float64), p["tf"].astype(np.float64)
        win = 2.0 ** -10
        same_sign = (np.sign(dd) == np.sign(d)).all(1) & ok      # (an ulp step across zero would turn the ray around)
        sel_n = same_sign[:, None, None] & (np.abs(exact_n) < win) & np.isfinite(tn)
        sel_f = same_sign[:, None, None] & (np.abs(exact_f) < win) & np.isfinite(tf)
        # per item (one ray, one node): the least slack of an entry plane covers the worst undershoot of an exit plane -- the exit
        # planes carry no allowance of their own, both sides' errors are charged to the entry planes once per node
        big = 1e30
        slack_n = np.where(sel_n, exact_n - tn, big).min((1, 2))
        under_f = np.where(sel_f, exact_f - tf, -big).max((1, 2))
        has = (slack_n < big) & (under_f > -big)
        assert (slack_n[slack_n < big] >= 0).all(), float(slack_n.min())
        assert (slack_n[has] >= np.maximum(under_f[has], 0)).all(), float((under_f[has] - slack_n[has]).max())
        # share of the allowance used: what the entry planes would overshoot without it + what the exit planes undershoot
        m_t = p["m_t"].astype(np.float64)
        over_n = np.where(sel_n, tn + m_t[:, None, None] - exact_n, -big).max((1, 2))
        used = (np.maximum(over_n[has], 0) + np.maximum(under_f[has], 0)) / m_t[has]
        used_worst = max(used_worst, float(used.max()))
        checked += int(sel_n.sum() + sel_f.sum())
    print("half-precision planes:", checked, "planes inside the window; worst entry plane uses", used_worst, "of the allowance")
    assert checked > 1_000_000 and used_worst < 1.0
