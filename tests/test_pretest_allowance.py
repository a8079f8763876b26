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


def _dot(a, b):
    return (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]  # cgmath: mul_element_wise().sum()


def _cross(a, b):
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=1)


def _moeller_trumbore(v0, e1, e2, ro, rd):
    eps = 0.00000001
    with np.errstate(all="ignore"):
        h = _cross(rd, e2)
        a = _dot(e1, h)
        f = 1.0 / a
        s = ro - v0
        u = f * _dot(s, h)
        q = _cross(s, e1)
        v = f * _dot(rd, q)
        t = f * _dot(e2, q)
    return ~((a < eps) & (a > -eps)) & ~((u < 0.0) | (u > 1.0)) & ~((v < 0.0) | (u + v > 1.0)) & (t > eps)


@pytest.mark.parametrize("regime", ["near", "extreme"])
def test_no_pair_the_pre_test_drops_passes_the_reference_test(product_lib, regime):
    from raymond_amd import probe

    rng = np.random.default_rng(11 if regime == "near" else 12)
    hits = 0
    worst = 0.0
    for _ in range(3):
        n = 400_000
        scale = 10.0 ** rng.uniform(-3, 1, n)
        far = (1, 5) if regime == "extreme" else (-1, 3)
        p0 = rng.uniform(-5, 5, (n, 3)) * 10.0 ** rng.uniform(far[0], far[1], (n, 1)) + rng.normal(size=(n, 3)) * scale[:, None]
        p1 = p0 + rng.normal(size=(n, 3)) * scale[:, None]
        p2 = p0 + rng.normal(size=(n, 3)) * scale[:, None]
        sliver = rng.uniform(size=n) < 0.3
        p2 = np.where(sliver[:, None], p0 + (p1 - p0) * rng.uniform(0, 1, (n, 1)) + rng.normal(size=(n, 3)) * (scale * 10.0 ** rng.uniform(-9, -2, n))[:, None], p2)
        e1, e2 = p1 - p0, p2 - p0
        nrm = _cross(e1, e2)
        nl = np.sqrt(_dot(nrm, nrm))
        ok = nl > 0
        # a point in or near the triangle, a direction almost in its plane, an origin far back along it (plus a nudge)
        target = p0 + e1 * rng.uniform(-0.2, 1.2, (n, 1)) + e2 * rng.uniform(-0.2, 1.2, (n, 1))
        with np.errstate(all="ignore"):
            inplane = e1 * rng.normal(size=(n, 1)) + e2 * rng.normal(size=(n, 1))
            inplane /= np.sqrt(_dot(inplane, inplane))[:, None]
            tilt = 10.0 ** (rng.uniform(-9.5, -5, n) if regime == "extreme" else rng.uniform(-9, 0, n)) * rng.choice([-1.0, 1.0], n)
            rd = inplane + (nrm / nl[:, None]) * tilt[:, None]
            rd /= np.sqrt(_dot(rd, rd))[:, None]
        dist = 10.0 ** (rng.uniform(1, 6, n) if regime == "extreme" else rng.uniform(-3, 4, n))
        ro = target - rd * dist[:, None] + rng.normal(size=(n, 3)) * (scale * 10.0 ** rng.uniform(-12, -1, n))[:, None]
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
