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


# ---- R-E in closed form (round 3): optimality conditions certify the global optimum of a convex problem ---------------
def _rod_ellipsoid_case(rng, n, degenerate=False):
    ec = rng.uniform(-0.5, 0.5, (n, 3))
    er = rng.uniform(0.3, 1.5, (n, 3))
    if degenerate:
        # axis-aligned frames, equal semi-axes, rods through the centre / in the symmetry planes / along the axes:
        # every special branch of the closest-point case analysis
        eq = np.tile([1.0, 0.0, 0.0, 0.0], (n, 1))
        rq = np.tile([1.0, 0.0, 0.0, 0.0], (n, 1))       # rod axis = z
        er[: n // 4] = er[: n // 4, :1]                   # spheres
        er[n // 4: n // 2, 1] = er[n // 4: n // 2, 0]     # two equal axes
        rc = ec + rng.integers(-2, 3, (n, 3)).astype(float) * rng.choice([0.0, 0.25, 0.5, 1.0, 2.0], (n, 1))
        L = rng.choice([0.0, 0.5, 1.0, 4.0], n)
    else:
        eq = rng.normal(size=(n, 4)); eq /= np.linalg.norm(eq, axis=1, keepdims=True)
        rq = rng.normal(size=(n, 4)); rq /= np.linalg.norm(rq, axis=1, keepdims=True)
        d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        rc = ec + d * rng.uniform(0.0, 3.0, (n, 1))       # from deep inside to well outside
        L = rng.uniform(0.0, 3.0, n)
    r = rng.uniform(0.1, 0.4, n)
    kind = np.concatenate([np.full(n, 1), np.full(n, 2)]).astype(np.int32)
    center, quat = np.concatenate([rc, ec]), np.concatenate([rq, eq])
    shape = np.concatenate([np.stack([r, L, np.zeros(n)], axis=1), er])
    pairs = np.stack([np.arange(n), np.arange(n) + n], axis=1).astype(np.int32)
    return dict(pairs=pairs, kind=kind, center=center, quat=quat, shape=shape, n=n, ec=ec, eq=eq, er=er, rc=rc, rq=rq,
                r=r, L=L)


def check_rod_ellipsoid_optimality(oracle, case, out, tol=1e-9):
    """KKT of min_t s(p(t)) with s the signed distance to the ellipsoid (convex): cp2 on the surface, `normal` = minus
    the surface normal there, cp1 - cp2 = s * n_E, and the slope n_E . (p1 - p0) zero inside (0, 1), >= 0 at t = 0,
    <= 0 at t = 1."""
    n, ec, eq, er, r = case["n"], case["ec"], case["eq"], case["er"], case["r"]
    seg = oracle.spherocylinder_segments(case["rc"], case["rq"], r, case["L"])
    p0, p1 = seg[:, 0:3], seg[:, 3:6]
    conj = eq * np.array([1.0, -1.0, -1.0, -1.0])
    xb = oracle.quat_rotate(conj, out["cp2"] - ec)                      # closest point in the body frame
    np.testing.assert_allclose(np.sum((xb / er) ** 2, axis=1), 1.0, atol=1e-12)
    grad = oracle.quat_rotate(eq, xb / er ** 2)
    n_e = grad / np.linalg.norm(grad, axis=1, keepdims=True)
    np.testing.assert_allclose(out["normal"], -n_e, atol=1e-12)
    s = out["sep"] + r
    np.testing.assert_allclose(out["cp1"] - out["cp2"], s[:, None] * n_e, atol=1e-9)
    d = p1 - p0
    len2 = np.maximum(np.sum(d * d, axis=1), 1e-300)
    t = np.sum((out["cp1"] - p0) * d, axis=1) / len2
    slope = np.sum(n_e * d, axis=1)
    point_like = np.sum(d * d, axis=1) < 1e-24
    assert np.all((t > -1e-12) & (t < 1 + 1e-12))
    inner = (t > 1e-9) & (t < 1 - 1e-9) & ~point_like
    # (inside the ellipsoid the signed distance has kinks -- the medial surface, where the closest point jumps -- and the
    #  deepest point of a centreline usually sits on one: the slope condition is for the smooth, exterior part)
    smooth = inner & (s > 0)
    assert np.all(np.abs(slope[smooth]) < tol * np.sqrt(len2[smooth]) + 1e-9), np.abs(slope[smooth]).max()
    at0 = (t <= 1e-9) & ~point_like
    at1 = (t >= 1 - 1e-9) & ~point_like
    assert np.all(slope[at0] > -1e-9) and np.all(slope[at1] < 1e-9)
    # the minimisation over the centreline against brute force, kinks or not: the exact point - ellipsoid distance
    # (a rod of zero length) at 65 stations is nowhere below what was found
    stations = np.linspace(0.0, 1.0, 65)
    lowest = np.full(n, np.inf)
    shape0 = case["shape"].copy()
    shape0[:n, 1] = 0.0
    for u in stations:
        c = case["center"].copy()
        c[:n] = p0 + u * d
        o = oracle.contact_mixed(case["pairs"], case["kind"], c, case["quat"], shape0)
        lowest = np.minimum(lowest, o["sep"])
    assert np.all(out["sep"] <= lowest + 1e-12), float((out["sep"] - lowest).max())
    assert np.all(out["sep"] >= lowest - 0.05 * np.sqrt(len2) - 1e-12)   # (|slope| <= |d|, stations 1/64 apart)
    return dict(inner=int(inner.sum()), at0=int(at0.sum()), at1=int(at1.sum()), inside=int((s < 0).sum()))


def test_rod_ellipsoid_closed_form_satisfies_the_optimality_conditions(oracle):
    case = _rod_ellipsoid_case(np.random.default_rng(23), 4000)
    out = oracle.contact_mixed(case["pairs"], case["kind"], case["center"], case["quat"], case["shape"])
    stats = check_rod_ellipsoid_optimality(oracle, case, out)
    assert stats["inner"] > 500 and stats["at0"] > 200 and stats["at1"] > 200 and stats["inside"] > 200, stats


def test_rod_ellipsoid_closed_form_on_degenerate_configurations(oracle):
    # exact zeros in the body frame (symmetry planes, axes, the centre), equal semi-axes, zero-length rods
    case = _rod_ellipsoid_case(np.random.default_rng(29), 4000, degenerate=True)
    out = oracle.contact_mixed(case["pairs"], case["kind"], case["center"], case["quat"], case["shape"])
    assert np.all(np.isfinite(out["sep"])) and np.all(np.isfinite(out["normal"])) and np.all(np.isfinite(out["cp2"]))
    # (a rod through the exact centre of an ellipsoid has no unique closest point: the conditions that do not depend
    # on uniqueness still hold)
    stats = check_rod_ellipsoid_optimality(oracle, case, out)
    assert stats["inside"] > 100 and stats["inner"] + stats["at0"] + stats["at1"] > 2000, stats
    # spheres: the closed form of a segment against a sphere
    k = case["n"] // 4
    seg = oracle.spherocylinder_segments(case["rc"][:k], case["rq"][:k], case["r"][:k], case["L"][:k])
    dist = oracle.distance_point_segment(case["ec"][:k], seg[:, 0:3], seg[:, 3:6])[0]
    np.testing.assert_allclose(out["sep"][:k], dist - case["er"][:k, 0] - case["r"][:k], atol=1e-12)


def certify_sphere_ellipsoid_disagreements(oracle, centre, r, ec, eq, er, closed, minimiser, tol=1e-4):
    """The default S-E route is the exact closed form; SURVEY 8f.4 / the reference routine is the 9-start L-BFGS of
    PointEllipsoid.hpp:94-135 (minus r), which minimises the EUCLIDEAN distance |foot point - point| over the surface
    and can stop in a local minimum.  For every pair where the two separations differ by more than `tol`: the closed
    form's surface point must be the closer one, and it must satisfy the optimality conditions of the closest-point
    problem (on the surface, point - foot parallel to the surface normal there, sep = +/-|point - foot| - r with the
    sign of inside / outside) -- i.e. the disagreement is the minimiser's local minimum, pair by pair.  `closed` /
    `minimiser`: dicts with sep [n] and cp [n][3] (the point on the ellipsoid).  Returns the number of such pairs."""
    bad = np.flatnonzero(np.abs(closed["sep"] - minimiser["sep"]) > tol)
    if len(bad) == 0:
        return 0
    p, cpc, cpm = centre[bad], closed["cp"][bad], minimiser["cp"][bad]
    dc_, dm_ = np.linalg.norm(cpc - p, axis=1), np.linalg.norm(cpm - p, axis=1)
    assert np.all(dc_ <= dm_ + 1e-12), "closed form farther than the minimiser: %g" % float((dc_ - dm_).max())
    conj = eq[bad] * np.array([1.0, -1.0, -1.0, -1.0])
    xb = oracle.quat_rotate(conj, cpc - ec[bad])                          # foot point, body frame
    np.testing.assert_allclose(np.sum((xb / er[bad]) ** 2, axis=1), 1.0, atol=1e-12)
    grad = oracle.quat_rotate(eq[bad], xb / er[bad] ** 2)
    n_e = grad / np.linalg.norm(grad, axis=1, keepdims=True)
    yb = oracle.quat_rotate(conj, p - ec[bad])
    inside = np.sum((yb / er[bad]) ** 2, axis=1) < 1.0
    signed = np.where(inside, -dc_, dc_)
    np.testing.assert_allclose(p - cpc, signed[:, None] * n_e, atol=1e-9)   # stationarity: the gap is along the normal
    np.testing.assert_allclose(closed["sep"][bad], signed - r[bad], atol=1e-12)
    # ... and the minimiser's own point is a worse stationary point or none: nowhere closer than the closed form's
    return int(len(bad))


def test_zero_length_rod_is_the_exact_sphere_ellipsoid_distance(oracle):
    # R-E of a rod of zero length = exact signed point - ellipsoid distance minus r; S-E = the reference's L-BFGS
    # point - ellipsoid distance minus r, good to its own 1e-4 (UnitTestEllipsoidEllipsoid.cpp:52-53) -- outside AND
    # inside the ellipsoid (negative distances)
    rng = np.random.default_rng(31)
    case = _rod_ellipsoid_case(rng, 600)
    case["shape"][: case["n"], 1] = 0.0
    out_re = oracle.contact_mixed(case["pairs"], case["kind"], case["center"], case["quat"], case["shape"])
    kind_s = case["kind"].copy()
    kind_s[: case["n"]] = 0
    exact_se = oracle.contact_mixed(case["pairs"], kind_s, case["center"], case["quat"], case["shape"])
    for key in ("sep", "normal", "cp1", "cp2"):   # S-E's default route IS the rod of zero length
        assert np.array_equal(exact_se[key], out_re[key]), key
    with oracle.sphere_ellipsoid_minimiser_route():
        out_se = oracle.contact_mixed(case["pairs"], kind_s, case["center"], case["quat"], case["shape"])
    assert (out_re["sep"] + case["r"] < -0.05).sum() > 30
    np.testing.assert_allclose(out_re["sep"], out_se["sep"], atol=1e-4)
    far = np.abs(out_re["sep"] + case["r"]) > 0.05   # (the normal of a point on the surface is that point's own)
    np.testing.assert_allclose(out_re["normal"][far], out_se["normal"][far], atol=5e-3)


def tiny_coordinate_case():
    """points with one body-frame coordinate 110-300 decades below the others, on spheres, spheroids and a general
    ellipsoid, as a sphere-ellipsoid contact problem (identity frames); and the same with that coordinate exactly 0"""
    tiny = [1e-110, 1e-150, 1e-200, 1e-280, 3e-300]
    radii = [(1.0, 1.0, 1.0), (1.0, 1.0, 0.5), (1.5, 1.0, 1.0), (0.7, 0.7, 1.3), (1.2, 0.9, 0.6)]
    base = [(0.5, 0.5), (0.2, 1.4), (2.0, 0.1), (0.05, 0.02)]
    pts, pts0, ell = [], [], []
    for e in radii:
        for a, b in base:
            for axis in range(3):
                for t in tiny:
                    for sg in (1.0, -1.0):
                        y = [a, b]
                        y.insert(axis, sg * t)
                        y0 = [a, b]
                        y0.insert(axis, 0.0)
                        pts.append(y); pts0.append(y0); ell.append(e)
    n = len(pts)
    pts, pts0, ell = np.array(pts), np.array(pts0), np.array(ell)
    kind = np.concatenate([np.zeros(n), np.full(n, 2)]).astype(np.int32)
    quat = np.tile([1.0, 0.0, 0.0, 0.0], (2 * n, 1))
    shape = np.concatenate([np.stack([np.full(n, 0.1), np.zeros(n), np.zeros(n)], axis=1), ell])
    pairs = np.stack([np.arange(n), np.arange(n) + n], axis=1).astype(np.int32)
    return dict(pairs=pairs, kind=kind, quat=quat, shape=shape, center=np.concatenate([pts, np.zeros((n, 3))]),
                center0=np.concatenate([pts0, np.zeros((n, 3))]), pts=pts, ell=ell)


def test_point_ellipsoid_closed_form_with_a_coordinate_far_below_the_others(oracle):
    # round-3 review: with equal semi-axes (sphere / spheroid: the pole offset m is 0) Newton starts at u = z2, and a
    # body-frame coordinate ~1e-200 next to ones of order 1 overflowed Q = (r z / u)^2 -> NaN -> a garbage closest
    # point.  Coordinates more than 100 decades below the point's largest one now count as zero (symmetry-plane
    # branches): the result must be finite and equal the one for an exact zero to rounding.
    c = tiny_coordinate_case()
    pts, ell = c["pts"], c["ell"]
    out = oracle.contact_mixed(c["pairs"], c["kind"], c["center"], c["quat"], c["shape"])
    ref = oracle.contact_mixed(c["pairs"], c["kind"], c["center0"], c["quat"], c["shape"])
    np.testing.assert_allclose(out["sep"], ref["sep"], rtol=0, atol=1e-13)
    for k in ("normal", "cp1", "cp2"):
        # (inside the focal region of the symmetry plane the closest point is not unique: it goes to the side the tiny
        # coordinate's sign names, for an exact zero to the positive one -- equal up to that mirror image)
        assert np.all(np.isfinite(out[k])), k
        np.testing.assert_allclose(np.abs(out[k]), np.abs(ref[k]), rtol=0, atol=1e-13, err_msg=k)
    sph = np.all(ell == ell[:, :1], axis=1)
    np.testing.assert_allclose(out["sep"][sph], np.linalg.norm(pts[sph], axis=1) - ell[sph, 0] - 0.1, atol=1e-14)


def test_sphere_ellipsoid_default_route_against_the_reference_route_pair_by_pair(oracle):
    # the drop-in's default S-E route (closed form) differs from the routine SURVEY 8f.4 prescribes
    # (PointEllipsoid.hpp:94-135 minus r, selectable as route 1).  Wherever the two disagree by more than the
    # reference's 1e-4, the closed form must be the better answer of the SAME problem -- certified pair by pair.
    # Elongated and flat ellipsoids and points close to the long axes are where the nine-start minimiser stalls.
    rng = np.random.default_rng(77)
    n = 6000
    case = _rod_ellipsoid_case(rng, n)
    case["er"][: n // 2] = rng.uniform(0.02, 4.0, (n // 2, 3))     # needles and flakes for half of them
    case["shape"][n:] = case["er"]
    case["shape"][:n, 1] = 0.0
    kind = case["kind"].copy()
    kind[:n] = 0
    closed = oracle.contact_mixed(case["pairs"], kind, case["center"], case["quat"], case["shape"])
    with oracle.sphere_ellipsoid_minimiser_route():
        mini = oracle.contact_mixed(case["pairs"], kind, case["center"], case["quat"], case["shape"])
    agree = np.abs(closed["sep"] - mini["sep"]) <= 1e-4
    assert agree.mean() >= 0.99, agree.mean()
    assert (~agree).sum() >= 3    # (the certificate below must have something to certify)
    k = certify_sphere_ellipsoid_disagreements(oracle, case["rc"], case["r"], case["ec"], case["eq"], case["er"],
                                               dict(sep=closed["sep"], cp=closed["cp2"]), dict(sep=mini["sep"], cp=mini["cp2"]))
    print("S-E: %d of %d pairs differ by more than 1e-4 between the closed form and the reference route; in every one "
          "the closed form's foot point is the closer one and satisfies the optimality conditions" % (k, n))
