"""Mutation tests of the culling hierarchy's conservative margins: the parity workload must NOTICE a cull that is too tight.

The test build (libpt_testhooks.so, -DPT_TEST_HOOKS) can scale each family of margins when it builds the tables; the
shipped library has no such knob.  Workload (tools/mutation_sweep.py): frames of Tor.obj (sphere-tree path) and of a
replicated scene (box-tree path) plus 80 000 explicit rays, three quarters of them aimed exactly at -- or a hair beside --
edges and vertices, all compared bit for bit with the CPU oracle.

What the full sweep showed on the MI355X (profiles/r02_mutation_sweep.jsonl), and what is asserted here:
  * as shipped (every scale 1): zero differing pixels, zero differing rays;
  * bounding spheres (r^2) and leaf boxes (half-extent) are TIGHT: scaled by 0.98 the workload already finds wrong hits;
  * the barycentric margins of the large class (m0, k1/k2, the quad slack, a_max) are proven bounds with room to spare:
    each of them alone can be set to ZERO without a wrong hit in this workload (the others cover it), all of them
    together are noticed from a scale of 0.1 downwards.  The test pins that threshold, so a change that silently eats the
    reserve (or a workload that stops reaching the edges) fails here.
"""
import importlib
import os
import sys

import pytest

pt = importlib.import_module("path-tracing_amd")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def bench():
    import mutation_sweep as M
    assert pt.device_count() >= 1
    lib = pt.load_library(pt.TESTHOOKS_LIB_PATH)
    lib.pt_test_set_mutation(b"reset", 0.0)
    w = M.Workload()
    yield lib, w
    lib.pt_test_set_mutation(b"reset", 0.0)


def _run(bench, **scales):
    lib, w = bench
    lib.pt_test_set_mutation(b"reset", 0.0)
    for fam, v in scales.items():
        lib.pt_test_set_mutation(fam.encode(), float(v))
    try:
        return w.run(lib)
    finally:
        lib.pt_test_set_mutation(b"reset", 0.0)


def test_as_shipped_nothing_differs(bench):
    assert _run(bench) == (0, 0)


@pytest.mark.parametrize("family", ["sphere_r2", "box"])
def test_tight_margins_are_noticed_at_two_percent(bench, family):
    px, rays = _run(bench, **{family: 0.98})
    assert px + rays > 0, f"{family} scaled by 0.98 went unnoticed"
    px, rays = _run(bench, **{family: 0.9})
    assert rays > 100


def test_barycentric_margins_are_noticed_together(bench):
    px, rays = _run(bench, m0=0.1, k12=0.1, quad_slack=0.1, a_max=0.1)
    assert rays > 10
    px, rays = _run(bench, m0=0.0, k12=0.0, quad_slack=0.0)
    assert rays > 1000


@pytest.mark.parametrize("family", ["m0", "k12", "quad_slack"])
def test_each_barycentric_margin_alone_is_covered_by_the_others(bench, family):
    """Not a requirement, a recorded fact: one bound at a time can vanish.  If this starts failing the reserve has shrunk."""
    px, rays = _run(bench, **{family: 0.5})
    assert (px, rays) == (0, 0)
