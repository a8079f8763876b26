"""The grid walk's sphere pre-test (grid_walk.hpp) drops a (ray, triangle) pair whose line passes the triangle's sphere by, WITHOUT running
core/src/geometry/primitives/triangle.rs:11-44 on it.  That is only right if the reference's test — as computed in binary64, with its own
operation order — fails for every such pair, rounding errors included.  internal.hpp: triangle_sphere argues an allowance for those errors; this
test evaluates the reference's test in numpy binary64 (same operations, same order as device_core.hpp: triangle_intersect) on pairs chosen to
make the errors large — rays almost in the triangle's plane (|a| down to the 1e-8 cut-off), origins up to 10^6 away, slivers, triangles up to
10^5 from the origin — and asserts that every pair the test accepts lies inside the pre-test's threshold, computed from the spheres the
product's own host code makes (rmd_probe_triangle_sphere: host only, no GPU).  The GPU suite has the complement: the DIAG build runs every
dropped pair of real renders through the test as well (tests/test_gpu_faults.py)."""
import numpy as np
import pytest


from pretest_pairs import adversarial_pairs, dot as _dot, moeller_trumbore as _moeller_trumbore


@pytest.mark.parametrize("regime", ["near", "extreme"])
def test_no_pair_the_pre_test_drops_passes_the_reference_test(product_lib, regime):
    from raymond_amd import probe

    rng = np.random.default_rng(11 if regime == "near" else 12)
    hits = 0
    worst = 0.0
    for _ in range(3):
        p0, p1, p2, ro, rd, ok = adversarial_pairs(regime, rng, 400_000)
        e1, e2 = p1 - p0, p2 - p0
        hit = _moeller_trumbore(p0, e1, e2, ro, rd)
        centre, r2a, kb = probe.triangle_sphere(np.concatenate([p0, p1, p2], axis=1))
        d = centre - ro
        along, dd = _dot(d, rd), _dot(d, d)
        lhs, rhs = dd - along * along, r2a + kb * dd  # the pre-test passes a pair when lhs <= rhs (grid_walk.hpp: pretest; a grid's kb is its largest)
        m = hit & ok & np.isfinite(lhs)
        hits += int(m.sum())
        dropped = m & ~(lhs <= rhs)
        assert not dropped.any(), (int(dropped.sum()), lhs[dropped][:3], rhs[dropped][:3])
        worst = max(worst, float((lhs[m] / rhs[m]).max()))
    assert hits > 20_000  # the generator does produce accepted pairs, grazing ones among them
    assert 0.5 < worst < 1.0  # ... some of them at a vertex, i.e. near the sphere's surface: the sphere is not wastefully large either
