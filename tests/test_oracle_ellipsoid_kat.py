"""Pins the oracle's L-BFGS and ellipsoid shared-normal distances on the reference's own tests:
mundy/math/tests/unit_tests/UnitTestMinimize.cpp:63-105 and
mundy/geom/tests/unit_tests/UnitTestEllipsoidEllipsoid.cpp:65-245 (tolerance 1e-4, "the best precision we can get")."""
import numpy as np
import pytest

ID = [1.0, 0.0, 0.0, 0.0]
TOL = 1e-4  # UnitTestEllipsoidEllipsoid.cpp:53


def test_minimize_simple_functions(oracle):
    # UnitTestMinimize.cpp:63-84
    c, x = oracle.minimize_test(0, [1.0, 1.0])
    assert abs(c) <= 1e-7 and np.all(np.abs(x) <= 1e-7)
    c, x = oracle.minimize_test(1, [1.0, 1.0])
    assert abs(c) <= 1e-7 and abs(x[0] - 2.0) <= 1e-7 and abs(x[1] + 1.0) <= 1e-7


def test_minimize_rosenbrock_42(oracle):
    # UnitTestMinimize.cpp:86-105, tolerance sqrt(1e-7)
    c, x = oracle.minimize_test(2, np.zeros(42))
    tol = np.sqrt(1e-7)
    assert abs(c) <= tol
    np.testing.assert_allclose(x, 1.0, atol=tol)


def euler_to_quat(roll, pitch, yaw):
    # mundy_math/Quaternion.hpp:1455-1470
    c1, c2, c3 = np.cos(0.5 * roll), np.cos(0.5 * pitch), np.cos(0.5 * yaw)
    s1, s2, s3 = np.sin(0.5 * roll), np.sin(0.5 * pitch), np.sin(0.5 * yaw)
    return np.stack([c1 * c2 * c3 + s1 * s2 * s3, s1 * c2 * c3 - c1 * s2 * s3, c1 * s2 * c3 + s1 * c2 * s3,
                     c1 * c2 * s3 - s1 * s2 * c3], axis=-1)


def random_sphere_ellipsoids(rng, n):
    c = rng.uniform(-10, 10, (n, 3))
    q = euler_to_quat(*(rng.random((3, n)) * 2 * np.pi))
    r = rng.uniform(0.1, 10.0, n)
    return c, q, np.repeat(r[:, None], 3, axis=1)


def test_analytical_sphere_cases(oracle):
    # SharedNormalDistanceBetweenEllipsoids.AnalyticalSphereTestCases (UnitTestEllipsoidEllipsoid.cpp:106-145)
    rng = np.random.default_rng(1)
    n = 2000
    c0, q0, r0 = random_sphere_ellipsoids(rng, n)
    c1, q1, r1 = random_sphere_ellipsoids(rng, n)
    out = oracle.distance_ellipsoid_ellipsoid(c0, q0, r0, c1, q1, r1)
    expected = np.linalg.norm(c1 - c0, axis=1) - r0[:, 0] - r1[:, 0]
    np.testing.assert_allclose(out["dist"], expected, atol=TOL, rtol=0)


def test_point_ellipsoid_sphere_cases(oracle):
    # SharedNormalDistanceBetweenEllipsoidAndPoint.AnalyticalSphereTestCases (:65-104)
    rng = np.random.default_rng(2)
    n = 2000
    c, q, r = random_sphere_ellipsoids(rng, n)
    p = rng.uniform(-10, 10, (n, 3))
    dist, cp, nrm = oracle.distance_point_ellipsoid(p, c, q, r)
    np.testing.assert_allclose(dist, np.linalg.norm(p - c, axis=1) - r[:, 0], atol=TOL, rtol=0)


def ellipsoid_cases(oracle):
    """(c0, q0, r0, c1, q1, r1, expected) of AnalyticalEllipsoidTestCases (:147-245)"""
    R = [3.0, 1.0, 2.0]
    cases = [([0, 0, 0], ID, R, [0, 0, 0], ID, R, -2.0), ([0, 0, 0], ID, R, [0, 0, 0], ID, [6.0, 2.0, 4.0], -3.0)]
    for e in (0.2, -0.2, 0.0):
        cases.append(([-3.0 - 0.5 * e, 0, 0], ID, R, [3.0 + 0.5 * e, 0, 0], ID, R, e))
    qy = oracle.quat_from_parallel_transport([[1.0, 0, 0]], [[0, 1.0, 0]])[0].tolist()
    for e in (0.2, -0.2, 0.0):
        cases.append(([0, 3.0 + 1.0 + e, 0], qy, R, [0, 0, 0], ID, R, e))
    return cases


def test_analytical_ellipsoid_cases(oracle):
    cases = ellipsoid_cases(oracle)
    cols = list(zip(*cases))
    out = oracle.distance_ellipsoid_ellipsoid(*[np.array(c, dtype=float) for c in cols[:6]])
    np.testing.assert_allclose(out["dist"], cols[6], atol=TOL, rtol=0)
    # outputs are consistent: unit opposite normals, dist = (cp2 - cp1) . n1
    np.testing.assert_allclose(np.linalg.norm(out["n1"], axis=1), 1.0, atol=1e-12)
    np.testing.assert_allclose(out["n2"], -out["n1"], atol=0)
    np.testing.assert_allclose(np.sum((out["cp2"] - out["cp1"]) * out["n1"], axis=1), out["dist"], atol=1e-14)


def test_rod_ellipsoid_extension_against_a_scan_of_the_centreline(oracle):
    # R-E has no reference implementation (LineSegmentEllipsoid.hpp is an empty stub): the build-side definition is
    # "point - ellipsoid distance of the rod's closest centreline point, minus the rod radius".  Independent check: the
    # point - ellipsoid distance (pinned above on the reference's own cases) scanned along the centreline -- it is a
    # convex function of the arclength for an exterior segment, so a scan plus a ternary refinement brackets its minimum.
    rng = np.random.default_rng(17)
    n = 40
    ec = rng.uniform(-0.5, 0.5, (n, 3))
    eq = rng.normal(size=(n, 4)); eq /= np.linalg.norm(eq, axis=1, keepdims=True)
    er = rng.uniform(0.4, 1.2, (n, 3))
    rq = rng.normal(size=(n, 4)); rq /= np.linalg.norm(rq, axis=1, keepdims=True)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rc = ec + d * rng.uniform(1.6, 3.0, (n, 1))      # centreline stays outside the ellipsoid (largest radius 1.2 ...
    L, r = rng.uniform(0.0, 3.0, n), rng.uniform(0.1, 0.4, n)
    kind = np.concatenate([np.full(n, 1), np.full(n, 2)]).astype(np.int32)
    center, quat = np.concatenate([rc, ec]), np.concatenate([rq, eq])
    shape = np.concatenate([np.stack([r, L, np.zeros(n)], axis=1), er])
    pairs = np.stack([np.arange(n), np.arange(n) + n], axis=1).astype(np.int32)
    out = oracle.contact_mixed(pairs, kind, center, quat, shape)
    seg = oracle.spherocylinder_segments(rc, rq, r, L)   # rows (p0, p1, r, L)
    p0, p1 = seg[:, 0:3], seg[:, 3:6]
    checked = 0
    for i in range(n):
        f = lambda t: oracle.distance_point_ellipsoid((p0[i] + np.asarray(t)[:, None] * (p1[i] - p0[i])),  # noqa: E731
                                                      np.repeat(ec[i:i + 1], len(t), 0), np.repeat(eq[i:i + 1], len(t), 0),
                                                      np.repeat(er[i:i + 1], len(t), 0))[0]
        ts = np.linspace(0.0, 1.0, 33)
        v = f(ts)
        if v.min() <= 0.05:
            continue  # ... but a long rod may still reach it: exterior cases only)
        k = int(np.argmin(v))
        lo, hi = ts[max(k - 1, 0)], ts[min(k + 1, 32)]
        for _ in range(30):
            a, b = lo + (hi - lo) / 3.0, hi - (hi - lo) / 3.0
            fa, fb = f([a, b])
            lo, hi = (lo, b) if fa < fb else (a, hi)
        best = float(f([0.5 * (lo + hi)])[0])
        assert abs(out["sep"][i] - (best - r[i])) < 2e-4, (i, out["sep"][i], best - r[i])
        # contact points: on the centreline, resp. on the ellipsoid surface, joined by the (unit) normal
        t_cp = np.dot(out["cp1"][i] - p0[i], p1[i] - p0[i]) / max(np.dot(p1[i] - p0[i], p1[i] - p0[i]), 1e-300)
        assert -1e-9 <= t_cp <= 1 + 1e-9
        gap = out["cp2"][i] - out["cp1"][i]
        np.testing.assert_allclose(gap / np.linalg.norm(gap), out["normal"][i], atol=5e-3)
        np.testing.assert_allclose(np.linalg.norm(gap) - r[i], out["sep"][i], atol=2e-4)
        checked += 1
    assert checked >= 25
