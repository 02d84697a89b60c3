"""GPU parity, geometry: every kernel called through the C ABI (mundy_amd.ops -> libmundy_hip.so) and compared with
the CPU oracle on the same seeded inputs.  Bar: BIT-EXACT (fp64 +,-,*,/,sqrt in the reference's order, device built
with -ffp-contract=off); the reference's own test tolerances (1e-8 AABB, 1e-6 segments) are checked on its KATs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mundy_amd import ops as o
    cnt, arch = o.device_info()
    assert cnt >= 1 and arch.startswith("gfx950"), arch
    return o


def test_aabb_kats_through_c_abi(ops, oracle):
    # UnitTestComputeAABB.cpp:204-232 (spherocylinders) and :179-202 (ellipsoids), tol 1e-8
    from gpu_util import dev, host
    X90 = [1.0 / np.sqrt(2.0), 1.0 / np.sqrt(2.0), 0.0, 0.0]
    ID = [1.0, 0.0, 0.0, 0.0]
    c = dev(np.array([[1, -2, 3]] * 4, dtype=np.float64))
    q = dev(np.array([ID, ID, ID, X90]))
    got = host(ops.compute_aabb_spherocylinders(c, q, dev(np.array([4.0, 0, 2, 2])), dev(np.array([0.0, 4, 4, 3]))))
    exp = [[-3, -6, -1, 5, 2, 7], [1, -2, 1, 1, -2, 5], [-1, -4, -1, 3, 0, 7], [-1, -5.5, 1, 3, 1.5, 5]]
    np.testing.assert_allclose(got, exp, atol=1e-8, rtol=0)
    ce = dev(np.array([[1, -2, 3], [1, -2, 3], [0, 0, 0], [1, -2, 3]], dtype=np.float64))
    qe = dev(np.array([ID, ID, X90, X90]))
    re_ = dev(np.array([[4, 4, 4], [4, 5, 6], [4, 5, 6], [4, 5, 6]], dtype=np.float64))
    exp = [[-3, -6, -1, 5, 2, 7], [-3, -7, -3, 5, 3, 9], [-4, -6, -5, 4, 6, 5], [-3, -8, -2, 5, 4, 8]]
    np.testing.assert_allclose(host(ops.compute_aabb_ellipsoids(ce, qe, re_)), exp, atol=1e-8, rtol=0)
    got = host(ops.compute_aabb_spheres(dev(np.array([[0.0, 0, 0], [1, -2, 3]])), dev(np.array([1.0, 4.0]))))
    np.testing.assert_allclose(got, [[-1, -1, -1, 1, 1, 1], [-3, -6, -1, 5, 2, 7]], atol=1e-8, rtol=0)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 100_003])
def test_per_body_kernels_bit_exact(ops, oracle, n):
    from gpu_util import assert_bits_equal, dev, host, random_rods
    import torch
    rng = np.random.default_rng(n)
    c, q, r, L = random_rods(rng, n, 50.0)
    radii = rng.uniform(0.2, 2.0, (n, 3))
    dc, dq, dr, dL, drad = dev(c), dev(q), dev(r), dev(L), dev(radii)
    assert_bits_equal(host(ops.compute_aabb_spheres(dc, dr)), oracle.compute_aabb_spheres(c, r), "aabb spheres")
    assert_bits_equal(host(ops.compute_aabb_spherocylinders(dc, dq, dr, dL)),
                      oracle.compute_aabb_spherocylinders(c, q, r, L), "aabb rods")
    assert_bits_equal(host(ops.compute_aabb_ellipsoids_conservative(dc, dq, drad)),
                      oracle.compute_aabb_ellipsoids_conservative(c, q, radii), "conservative ellipsoid aabb (extension)")
    assert_bits_equal(host(ops.compute_aabb_ellipsoids(dc, dq, drad)), oracle.compute_aabb_ellipsoids(c, q, radii),
                      "aabb ellipsoids")
    seg = ops.spherocylinder_segments(dc, dq, dr, dL)
    oseg = oracle.spherocylinder_segments(c, q, r, L)
    assert_bits_equal(host(seg), oseg, "segment records")
    assert_bits_equal(host(ops.compute_aabb_segments(seg)), oracle.compute_aabb_spherocylinders(c, q, r, L),
                      "aabb of segment records")
    assert_bits_equal(host(ops.bounding_radius_spherocylinders(dr, dL)), oracle.bounding_radius_spherocylinders(r, L),
                      "bounding radius rods")
    assert_bits_equal(host(ops.bounding_radius_ellipsoids(drad)), oracle.bounding_radius_ellipsoids(radii),
                      "bounding radius ellipsoids")
    torch.cuda.synchronize()


def test_segseg_reference_kats_through_c_abi(ops):
    # UnitTestSegmentSegment.cpp:417-472
    from gpu_util import dev, host
    a1 = dev(np.array([[0.2257294191072674, 0.30159862841764695, 0.12784820133135649],
                       [9.64101615137754, 6, 3.18961417478521]]))
    a2 = dev(np.array([[0.22572948671663273, 0.30159858045792487, 0.1278481814714105],
                       [9.64101615137754, 6, 8.189614174785209]]))
    b1 = dev(np.array([[0.5220039935659887, 0.88764831847472003, -0.2219484914838093],
                       [10.39230484541326, 6, 0.6472696138825587]]))
    b2 = dev(np.array([[0.50288066060587278, 0.66779290982621586, -0.5723507723323677],
                       [10.39230484541326, 6, 5.647269613882559]]))
    dist, cp1, cp2, s, t, sep = [host(x) for x in ops.distance_segment_segment(a1, a2, b1, b2)]
    assert abs(dist[0] - 0.74347757392471259) < 1e-6 and abs(dist[1] - 0.7512886940357237) < 1e-6
    np.testing.assert_allclose(cp2[0], [0.52067221426302679, 0.87233723836682309, -0.24635106326970288], atol=1e-6)
    assert abs(s[0] - 1.0) < 1e-6 and abs(t[0] - 0.069641589451982497) < 1e-6
    rev = host(ops.distance_segment_segment(b1, b2, a1, a2)[0])
    assert abs(rev[1] - dist[1]) < 1e-6


def _segment_cases(rng, n):
    """generic, near-parallel, exactly colinear, degenerate (point) and touching segments"""
    a0, a1 = rng.uniform(-2, 2, (n, 3)), rng.uniform(-2, 2, (n, 3))
    b0, b1 = rng.uniform(-2, 2, (n, 3)), rng.uniform(-2, 2, (n, 3))
    k = n // 8
    b0[:k] = a0[:k] + rng.uniform(-1, 1, (k, 3))            # parallel translate -> colinear branch
    b1[:k] = a1[:k] + (b0[:k] - a0[:k])
    b1[k:2 * k] = b0[k:2 * k] + (a1[k:2 * k] - a0[k:2 * k]) * rng.uniform(0.2, 3, (k, 1)) \
        + 1e-9 * rng.normal(size=(k, 3))                     # nearly parallel
    a1[2 * k:3 * k] = a0[2 * k:3 * k]                        # degenerate first segment
    b1[3 * k:4 * k] = b0[3 * k:4 * k]                        # degenerate second segment
    b0[4 * k:5 * k] = 0.5 * (a0[4 * k:5 * k] + a1[4 * k:5 * k])  # touching: b0 on segment a
    return a0, a1, b0, b1


def test_distance_batches_bit_exact(ops, oracle):
    from gpu_util import assert_bits_equal, dev, host
    rng = np.random.default_rng(42)
    n = 200_000
    a0, a1, b0, b1 = _segment_cases(rng, n)
    got = [host(x) for x in ops.distance_segment_segment(dev(a0), dev(a1), dev(b0), dev(b1))]
    exp = oracle.distance_segment_segment(a0, a1, b0, b1)
    for g, e, name in zip(got, exp, ("dist", "cp1", "cp2", "s", "t", "sep")):
        assert_bits_equal(g, e, "seg-seg " + name)
    p = rng.uniform(-2, 2, (n, 3))
    got = [host(x) for x in ops.distance_point_segment(dev(p), dev(a0), dev(a1))]
    exp = oracle.distance_point_segment(p, a0, a1)
    for g, e, name in zip(got, exp, ("dist", "cp", "t", "sep")):
        assert_bits_equal(g, e, "point-seg " + name)
    r1, r2 = rng.uniform(0.1, 1, n), rng.uniform(0.1, 1, n)
    got = [host(x) for x in ops.distance_sphere_sphere(dev(a0), dev(r1), dev(b0), dev(r2))]
    exp = oracle.distance_sphere_sphere(a0, r1, b0, r2)
    assert_bits_equal(got[0], exp[0], "sphere-sphere dist")
    assert_bits_equal(got[1], exp[1], "sphere-sphere sep")
    # distance(Point, Sphere, sep) / distance(LineSegment, Sphere, cp, t, sep): PointSphere.hpp:69-79,
    # LineSegmentSphere.hpp:88-100 (the segment cases above include degenerate and far-away segments)
    got = [host(x) for x in ops.distance_point_sphere(dev(p), dev(b0), dev(r1))]
    exp = oracle.distance_point_sphere(p, b0, r1)
    for g, e, name in zip(got, exp, ("dist", "sep")):
        assert_bits_equal(g, e, "point-sphere " + name)
    got = [host(x) for x in ops.distance_segment_sphere(dev(a0), dev(a1), dev(p), dev(r2))]
    exp = oracle.distance_segment_sphere(a0, a1, p, r2)
    for g, e, name in zip(got, exp, ("dist", "cp", "t", "sep")):
        assert_bits_equal(g, e, "segment-sphere " + name)


def test_segseg_known_distance_property(ops):
    # manufactured distances of UnitTestSegmentSegment.cpp:223-291 at 10^6 samples (the reference's sample count),
    # tolerance 1e-6 (its TEST_DOUBLE_EPSILON)
    from gpu_util import dev, host
    from test_oracle_geom_kat import known_distance_segments
    rng = np.random.default_rng(99)
    dist_e, a1, a2, b1, b2, a12, b12, u, v, deg = known_distance_segments(rng, 1_000_000)
    dist, cp1, cp2, s, t, sep = [host(x) for x in ops.distance_segment_segment(dev(a1), dev(a2), dev(b1), dev(b2))]
    ok = (np.linalg.norm(np.cross(a2 - a1, b2 - b1), axis=1) ** 2 > 1e-6) | (deg == 5)
    np.testing.assert_allclose(dist[ok], dist_e[ok], atol=1e-6, rtol=0)
    np.testing.assert_allclose(cp1[ok], a12[ok], atol=1e-6, rtol=0)
    np.testing.assert_allclose(cp2[ok], b12[ok], atol=1e-6, rtol=0)
    np.testing.assert_allclose(s[ok], u[ok], atol=1e-6, rtol=0)
    np.testing.assert_allclose(t[ok], v[ok], atol=1e-6, rtol=0)


def test_segseg_intersecting_and_colinear_generators_at_the_reference_sample_count(ops):
    # the other two generators of UnitTestSegmentSegment.cpp (:74-103 intersecting, :129-140 colinear) at its 10^6
    # samples, through the C ABI; tolerances as in the reference (1e-6) / the oracle's own test
    from gpu_util import dev, host
    from test_oracle_geom_kat import check_colinear, check_intersecting, colinear_segments, intersecting_segments
    segseg = lambda *a: [host(x) for x in ops.distance_segment_segment(*[dev(v) for v in a])]
    pointseg = lambda *a: [host(x) for x in ops.distance_point_segment(*[dev(v) for v in a])]
    check_intersecting(segseg, *intersecting_segments(np.random.default_rng(12), 1_000_000))
    check_colinear(segseg, pointseg, *colinear_segments(np.random.default_rng(4), 1_000_000))


@pytest.mark.parametrize("periodic", [False, True])
def test_contact_spheres_bit_exact(ops, oracle, periodic):
    from gpu_util import assert_bits_equal, dev, host
    rng = np.random.default_rng(5)
    n, box = 20_000, np.array([30.0, 31.0, 29.0])
    c = rng.uniform(-5, 35, (n, 3)) if periodic else rng.uniform(0, 30, (n, 3))
    r = rng.uniform(0.5, 1.5, n)
    pairs = rng.integers(0, n, (150_000, 2)).astype(np.int32)
    pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    b = box if periodic else None
    sep, nrm = ops.contact_spheres(dev(pairs), dev(c), dev(r), box=b)
    osep, onrm = oracle.contact_spheres(pairs, c, r, box=b)
    assert_bits_equal(host(sep), osep, "sphere contact sep")
    assert_bits_equal(host(nrm), onrm, "sphere contact normal")


def test_contact_spherocylinders_bit_exact(ops, oracle):
    from gpu_util import assert_bits_equal, dev, host, random_rods
    rng = np.random.default_rng(6)
    n = 30_000
    c, q, r, L = random_rods(rng, n, 25.0)
    q[:100] = q[0]  # a block of exactly parallel rods -> colinear branch inside the contact kernel
    seg = oracle.spherocylinder_segments(c, q, r, L)
    pairs = np.stack([rng.integers(0, n, 200_000), rng.integers(0, n, 200_000)], axis=1).astype(np.int32)
    pairs[:5000] = rng.integers(0, 100, (5000, 2))
    pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    out = ops.contact_spherocylinders(dev(pairs), dev(seg), dev(c))
    exp = oracle.contact_spherocylinders(pairs, seg, c)
    for k in ("sep", "normal", "cp1", "cp2", "ra", "rb", "s", "t"):
        assert_bits_equal(host(out[k]), exp[k], "rod contact " + k)
    # empty list is fine
    e = ops.contact_spherocylinders(dev(pairs[:0]), dev(seg), dev(c))
    assert e["sep"].shape[0] == 0


def test_periodic_metric_bit_exact(ops, oracle):
    # PeriodicScaledMetric::sep / wrap (periodicity.hpp:812-823) and wrap_rigid of centres
    from gpu_util import assert_bits_equal, dev, host
    rng = np.random.default_rng(12)
    box = np.array([3.0, 5.0, 7.5])
    n = 100_000
    p1, p2 = rng.uniform(-40, 40, (n, 3)), rng.uniform(-40, 40, (n, 3))
    p1[:100] = np.round(p1[:100])          # points on cell faces / half-way images
    p2[:100] = p1[:100] + box * rng.integers(-3, 4, (100, 3)) * 0.5
    assert_bits_equal(host(ops.periodic_sep(box, dev(p1), dev(p2))), oracle.periodic_sep(box, p1, p2), "periodic sep")
    w = host(ops.wrap_rigid(box, dev(p1)))
    assert_bits_equal(w, oracle.periodic_wrap(box, p1), "wrap_rigid")
    assert np.all(w >= 0) and np.all(w < box)


def test_triclinic_metric_bit_exact(ops, oracle):
    # PeriodicMetric (periodicity.hpp:233-332): sep / wrap_rigid / shift_image / sphere contacts under a tilted cell,
    # and the diagonal cell of the reference's own test (UnitTestPeriodicity.cpp:627-630)
    from gpu_util import assert_bits_equal, dev, host
    rng = np.random.default_rng(21)
    n = 100_000
    for h in (np.array([[10.0, 2.0, 1.0], [0.0, 9.0, 3.0], [0.0, 0.0, 8.0]]), np.diag([100.0, 100.0, 100.0]),
              np.array([[7.0, -1.5, 0.3], [0.4, 6.0, 2.0], [-0.8, 0.9, 5.0]])):
        assert_bits_equal(ops.unit_cell_inverse(h), oracle.unit_cell_inverse(h), "unit cell inverse")
        p1, p2 = rng.uniform(-40, 40, (n, 3)), rng.uniform(-40, 40, (n, 3))
        p1[:64] = np.round(p1[:64])
        p2[:64] = p1[:64] + (h @ (rng.integers(-3, 4, (64, 3)) * 0.5).T).T   # half-way images: round() ties
        assert_bits_equal(host(ops.periodic_sep(h, dev(p1), dev(p2))), oracle.periodic_sep_triclinic(h, p1, p2), "sep")
        assert_bits_equal(host(ops.wrap_rigid(h, dev(p1))), oracle.periodic_wrap_triclinic(h, p1), "wrap_rigid")
        img = rng.integers(-4, 5, (n, 3)).astype(np.int32)
        assert_bits_equal(host(ops.shift_image(h, dev(p1), dev(img))), oracle.shift_image_triclinic(h, p1, img),
                          "shift_image")
        # sphere contacts with this metric: same kernel as the orthorhombic case, other minimum image
        pairs = np.stack([np.arange(0, n - 1, 2), np.arange(1, n, 2)], 1).astype(np.int32)
        r = rng.uniform(0.3, 1.0, n)
        sep, nrm = ops.contact_spheres(dev(pairs), dev(p1), dev(r), box=h)
        d = oracle.periodic_sep_triclinic(h, p1[pairs[:, 0]], p1[pairs[:, 1]])
        cc = np.sqrt(d[:, 0] * d[:, 0] + (d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]))
        assert_bits_equal(host(sep), cc - r[pairs[:, 0]] - r[pairs[:, 1]], "triclinic sphere sep")
        assert_bits_equal(host(nrm), d * (1.0 / cc)[:, None], "triclinic sphere normal")
    with pytest.raises(Exception):
        ops.periodic_sep(np.zeros((3, 3)), dev(p1), dev(p2))


def test_degenerate_inputs_follow_the_reference(ops, oracle):
    # degenerate bodies as the reference treats them: zero-length rod == sphere-like segment, zero radius, identical
    # segments (distance 0 -> the contact normal is 0/0: NaN on both sides, no guard in the reference, SphereSphere.hpp:66-76)
    from gpu_util import assert_bits_equal, dev, host
    c = np.array([[0.0, 0, 0], [0.0, 0, 0], [3.0, 0, 0], [3.0, 0.0, 0]])
    q = np.array([[1.0, 0, 0, 0]] * 4)
    r = np.array([0.5, 0.5, 0.0, 0.25])
    L = np.array([0.0, 2.0, 2.0, 0.0])
    seg = oracle.spherocylinder_segments(c, q, r, L)
    pairs = np.array([[0, 1], [0, 2], [1, 2], [2, 3], [0, 3]], dtype=np.int32)
    got = ops.contact_spherocylinders(dev(pairs), dev(seg), dev(c))
    exp = oracle.contact_spherocylinders(pairs, seg, c)
    for k in ("sep", "s", "t", "cp1", "cp2"):
        assert_bits_equal(host(got[k]), exp[k], "degenerate " + k)
    gn, en = host(got["normal"]), exp["normal"]
    assert np.array_equal(np.isnan(gn), np.isnan(en)) and np.isnan(en[0]).all()   # coincident centrelines
    np.testing.assert_array_equal(gn[~np.isnan(en)], en[~np.isnan(en)])
    d, sep = ops.distance_sphere_sphere(dev(c[:1]), dev(r[:1]), dev(c[1:2]), dev(r[1:2]))
    od, osep = oracle.distance_sphere_sphere(c[:1], r[:1], c[1:2], r[1:2])
    assert host(d)[0] == od[0] == -1.0 and np.isnan(host(sep)).all() and np.isnan(osep).all()
