"""Pins the oracle's BBPGD (convex.hpp restatement) on the reference's own test problems:
mundy/math/tests/unit_tests/UnitTestConvex.cpp:46-143 (3x3 SPD, three spaces), :145-229 and :416-524 (random strictly
diagonally dominant P-matrix LCPs, N = 3, 7, 200), run as :529-606 do: x0 = 99.99, max_iters = 1000, tol = 1e-6,
asserting converged, num_iters <= max and |x - x*| <= 10 tol.  The reference draws its random entries from rand() /
OpenRAND Philox (absent here); the matrices are regenerated with numpy under the same recipe (input stream unpinned,
harmless: the acceptance criterion is the analytic solution).
"""
import numpy as np
import pytest

A3 = np.array([[2.0, -1.0, 0.0], [-1.0, 2.0, -1.0], [0.0, -1.0, 2.0]])
TOL = 1e-6


def random_lcp(n, seed):
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1.0, 1.0, (n, n))
    off = np.abs(A).sum(axis=1) - np.abs(np.diag(A))
    A[np.arange(n), np.arange(n)] = off + 10.0
    u01 = rng.random(n)
    active = rng.random(n) < 0.5
    x_star = np.where(active, u01 * 0.9 + 0.1, 0.0)
    g_star = np.where(active, 0.0, u01 * 0.9 + 0.1)
    return A, g_star - A @ x_star, x_star


CASES = [
    ("UnconstrainedSPD1", A3, np.array([1.0, 0.0, 1.0]), (0, 0.0, 0.0)),
    ("InactiveBox", A3, np.array([1.0, 0.0, 1.0]), (3, 0.0, 2.0)),
    ("ActiveBox", A3, np.array([9.0, 9.0, 9.0]), (3, 9.0, 10.0)),
]


@pytest.mark.parametrize("name,A,x_star,space", CASES, ids=[c[0] for c in CASES])
def test_analytic_3x3(oracle, name, A, x_star, space):
    q = -A @ x_star
    x, g, r = oracle.solve_cqpp_dense(A, q, space, np.full(3, 99.99), max_iters=1000, tol=TOL)
    assert r["converged"] and r["num_iters"] <= 1000 and r["residual"] <= TOL
    np.testing.assert_allclose(x, x_star, atol=10 * TOL, rtol=0)


@pytest.mark.parametrize("n", [3, 7, 200])
def test_random_lcp(oracle, n):
    A, q, x_star = random_lcp(n, seed=n)
    x, g, r = oracle.solve_cqpp_dense(A, q, (oracle.LOWER_BOUND, 0.0, 0.0), np.full(n, 99.99), max_iters=1000, tol=TOL)
    assert r["converged"] and r["num_iters"] <= 1000
    np.testing.assert_allclose(x, x_star, atol=10 * TOL, rtol=0)
    # complementarity of the returned pair (x, g = A x + q)
    np.testing.assert_allclose(g, A @ x + q, atol=1e-9)
    assert np.all(x >= 0) and np.all(g >= -1e-4) and abs(float(x @ g)) < 1e-3


def test_result_semantics(oracle):
    # already-converged start: zero iterations, grad copied from grad_tmp (convex.hpp:631-635)
    x_star = np.array([1.0, 0.0, 1.0])
    x, g, r = oracle.solve_cqpp_dense(A3, -A3 @ x_star, (0, 0.0, 0.0), x_star.copy(), tol=TOL)
    assert r["converged"] and r["num_iters"] == 0
    np.testing.assert_allclose(g, 0.0, atol=1e-15)
    # max_iters exhausted is not an error: converged = False, num_iters = max (convex.hpp:642-675)
    x, g, r = oracle.solve_cqpp_dense(A3, -A3 @ x_star, (0, 0.0, 0.0), np.full(3, 99.99), max_iters=2, tol=1e-14)
    assert not r["converged"] and r["num_iters"] == 2


def test_vector_kernel_quirks(oracle):
    # |alpha| or |beta| < 1e-15 is treated as exactly zero (convex.hpp:203-220, :228-247)
    x = np.array([1.0, 2.0, 3.0])
    y = np.array([10.0, 20.0, 30.0])
    oracle.axpby(1e-16, x, 2.0, y)
    np.testing.assert_array_equal(y, [20.0, 40.0, 60.0])
    z = np.empty(3)
    oracle.wrapped_axpbyz(1.0, x, -1e-16, y, z, (oracle.LOWER_BOUND, 1.5, 0.0))
    np.testing.assert_array_equal(z, [1.5, 2.0, 3.0])
    # BB1 with the 1e-14 denominator guard (convex.hpp:507-514)
    x0, g0 = np.zeros(3), np.zeros(3)
    assert oracle.bb_step(x0, g0, x, np.zeros(3)) == (1.0 + 4.0 + 9.0) / (1e-15 * 10)
    # projected-gradient residual uses max(0, +g) on the active set, as written (convex.hpp:452-456)
    assert oracle.residual(oracle.RESID_PROJECTED_GRADIENT, np.zeros(2), np.array([-5.0, 3.0]), (1, 0.0, 0.0)) == 3.0
    assert oracle.residual(oracle.RESID_PROJECTED_DIFF, np.array([1.0, 0.0]), np.array([2.0, 5.0]), (1, 0.0, 0.0)) == \
        pytest.approx(2.0, rel=1e-9)


def test_contact_operator_is_dt_DtMD(oracle):
    # A = dt D^T M D assembled densely must equal the matrix-free apply (NgpLcp.cpp:442-548), with and without
    # lever arms; and be symmetric PSD (convex.hpp:358-360)
    rng = np.random.default_rng(0)
    N, C, dt = 12, 30, 5e-3
    pairs = np.stack([rng.integers(0, N, C), rng.integers(0, N, C)], axis=1).astype(np.int32)
    pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    C = len(pairs)
    n = rng.normal(size=(C, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    ra, rb = rng.normal(size=(C, 3)), rng.normal(size=(C, 3))
    mt, mr = rng.uniform(0.5, 2, N), rng.uniform(0.5, 2, N)
    for rot in (False, True):
        D = np.zeros((6 * N, C))
        for c, (i, j) in enumerate(pairs):
            D[6 * i:6 * i + 3, c] = -n[c]
            D[6 * j:6 * j + 3, c] = n[c]
            if rot:
                D[6 * i + 3:6 * i + 6, c] = -np.cross(ra[c], n[c])
                D[6 * j + 3:6 * j + 6, c] = np.cross(rb[c], n[c])
        M = np.zeros(6 * N)
        M[np.arange(N)[:, None] * 6 + np.arange(3)] = mt[:, None]
        M[np.arange(N)[:, None] * 6 + 3 + np.arange(3)] = mr[:, None]
        A = dt * D.T @ (M[:, None] * D)
        x = rng.normal(size=C)
        y = oracle.contact_op_apply(pairs, n, ra if rot else None, rb if rot else None, mt, mr if rot else None, dt, x, N)
        np.testing.assert_allclose(y, A @ x, atol=1e-12)
        assert np.allclose(A, A.T) and np.linalg.eigvalsh(A).min() > -1e-12


def test_contact_lcp_matches_dense_and_threaded(oracle):
    rng = np.random.default_rng(1)
    N = 200
    c = rng.uniform(0, 8, (N, 3))
    r = np.ones(N)
    aabb = oracle.compute_aabb_spheres(c, r)
    lo, hi, R = oracle.grow(aabb, r, 0.3)
    pairs = oracle.search(oracle.SEARCH_SPHERES, lo, hi, c, R)
    sep, nrm = oracle.contact_spheres(pairs, c, r)
    mt = np.full(N, 1.0 / (6 * np.pi * 1.0 * 1e-3))
    x, g, res = oracle.solve_cqpp_contact(pairs, nrm, None, None, mt, None, 5e-3, sep, np.zeros(len(pairs)),
                                          max_iters=5000, tol=1e-6)
    assert res["converged"]
    assert np.all(x >= 0) and g.min() > -1e-5 and abs(float(x @ g)) < 1e-5 * max(1.0, x.sum())
    xt, gt, rt = oracle.solve_cqpp_contact(pairs, nrm, None, None, mt, None, 5e-3, sep, np.zeros(len(pairs)),
                                           max_iters=5000, tol=1e-6, threads=True)
    assert rt["converged"]
    # the LCP solution (g, and the net body forces D x) is unique even when x is not
    # (tol 1e-6: at |x| ~ 60 the projected-diff residual is quantised in steps of ulp(x)/1e-6 ~ 7e-9).  The threaded
    # solve sums forces with atomics, so its path -- 414 to 641 iterations over 200 runs against 514 serial -- and the
    # point inside the tolerance ball where it stops change from run to run: max |dg| was 3e-6 in the median and
    # 1.5e-5 at worst over those runs, hence 40 tol here (the GPU tests hold 20 tol against deterministic sums)
    np.testing.assert_allclose(gt, g, atol=4e-5)


def test_mundy_math_backend_problems(oracle):
    # Convex.MundyMathAnalyticalSolutions (UnitTestConvex.cpp:529-561, :608-615): the in-kernel backend on the three
    # 3x3 analytic problems and RandomLCP<3>, <7>: x0 = 99.99, max_iters 1000, tol 1e-6, |x - x*| <= 10 tol
    for name, A, x_star, space in CASES:
        x, g, it, res, conv = oracle.solve_small_cqpp_batch(A[None], (-A @ x_star)[None], space,
                                                            np.full((1, 3), 99.99), max_iters=1000, tol=TOL)
        assert conv[0] and it[0] <= 1000
        np.testing.assert_allclose(x[0], x_star, atol=10 * TOL, rtol=0)
    for n in (3, 7):
        A, q, x_star = random_lcp(n, seed=100 + n)
        x, g, it, res, conv = oracle.solve_small_cqpp_batch(A[None], q[None], (oracle.LOWER_BOUND, 0.0, 0.0),
                                                            np.full((1, n), 99.99), max_iters=1000, tol=TOL)
        assert conv[0]
        np.testing.assert_allclose(x[0], x_star, atol=10 * TOL, rtol=0)
