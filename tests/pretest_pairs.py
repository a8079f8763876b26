"""Adversarial (triangle, ray) pairs for the grid walk's sphere pre-test, and the reference's triangle test in numpy binary64 — shared by the CPU test
of the allowance (tests/test_pretest_allowance.py: the pre-test's formula in numpy) and the GPU test that runs the same pairs through the device's own
arithmetic (tests/test_gpu_reference_pins.py: rmd_probe_pretest_pairs)."""
import numpy as np


def dot(a, b):
    return (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]  # cgmath: mul_element_wise().sum()


def cross(a, b):
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=1)


def moeller_trumbore(v0, e1, e2, ro, rd):
    eps = 0.00000001
    with np.errstate(all="ignore"):
        h = cross(rd, e2)
        a = dot(e1, h)
        f = 1.0 / a
        s = ro - v0
        u = f * dot(s, h)
        q = cross(s, e1)
        v = f * dot(rd, q)
        t = f * dot(e2, q)
    return ~((a < eps) & (a > -eps)) & ~((u < 0.0) | (u > 1.0)) & ~((v < 0.0) | (u + v > 1.0)) & (t > eps)



def adversarial_pairs(regime, rng, n):
    """-> p0, p1, p2, ro, rd, ok: pairs chosen to make the test's rounding errors large — rays almost in the triangle's plane (|a| down to the 1e-8
    cut-off), origins up to 10^6 away, slivers, triangles up to 10^5 from the origin ("extreme") — aimed at points in or near the triangle."""
    scale = 10.0 ** rng.uniform(-3, 1, n)
    far = (1, 5) if regime == "extreme" else (-1, 3)
    p0 = rng.uniform(-5, 5, (n, 3)) * 10.0 ** rng.uniform(far[0], far[1], (n, 1)) + rng.normal(size=(n, 3)) * scale[:, None]
    p1 = p0 + rng.normal(size=(n, 3)) * scale[:, None]
    p2 = p0 + rng.normal(size=(n, 3)) * scale[:, None]
    sliver = rng.uniform(size=n) < 0.3
    p2 = np.where(sliver[:, None], p0 + (p1 - p0) * rng.uniform(0, 1, (n, 1)) + rng.normal(size=(n, 3)) * (scale * 10.0 ** rng.uniform(-9, -2, n))[:, None], p2)
    e1, e2 = p1 - p0, p2 - p0
    nrm = cross(e1, e2)
    nl = np.sqrt(dot(nrm, nrm))
    ok = nl > 0
    # a point in or near the triangle, a direction almost in its plane, an origin far back along it (plus a nudge)
    target = p0 + e1 * rng.uniform(-0.2, 1.2, (n, 1)) + e2 * rng.uniform(-0.2, 1.2, (n, 1))
    with np.errstate(all="ignore"):
        inplane = e1 * rng.normal(size=(n, 1)) + e2 * rng.normal(size=(n, 1))
        inplane /= np.sqrt(dot(inplane, inplane))[:, None]
        tilt = 10.0 ** (rng.uniform(-9.5, -5, n) if regime == "extreme" else rng.uniform(-9, 0, n)) * rng.choice([-1.0, 1.0], n)
        rd = inplane + (nrm / nl[:, None]) * tilt[:, None]
        rd /= np.sqrt(dot(rd, rd))[:, None]
    dist = 10.0 ** (rng.uniform(1, 6, n) if regime == "extreme" else rng.uniform(-3, 4, n))
    ro = target - rd * dist[:, None] + rng.normal(size=(n, 3)) * (scale * 10.0 ** rng.uniform(-12, -1, n))[:, None]
    return p0, p1, p2, ro, rd, ok
