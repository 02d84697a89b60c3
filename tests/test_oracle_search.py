"""Neighbour-search oracle: cell list == O(N^2) brute force, both predicates, free and periodic.
The reference's search is third-party (stk::search at GenNeighborLinkers.hpp:658, ArborX in scrap): parity unpinned;
the only reference assertion at this boundary is UnitTestGenNeighborLinks.cpp:73-152 (2 coincident spheres -> 1 link),
restated in test_two_coincident_spheres."""
import numpy as np
import pytest


def _bodies(rng, n, box):
    c = rng.uniform(0, box, (n, 3))
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    r = rng.uniform(0.3, 0.6, n)
    L = rng.uniform(0.5, 2.5, n)
    return c, q, r, L


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("periodic", [False, True])
@pytest.mark.parametrize("symmetric", [False, True])
def test_celllist_equals_bruteforce(oracle, kind, periodic, symmetric):
    rng = np.random.default_rng(10 * kind + periodic)
    box = np.array([9.0, 11.0, 13.0])
    c, q, r, L = _bodies(rng, 600, box)
    aabb = oracle.compute_aabb_spherocylinders(c, q, r, L)
    lo, hi, R = oracle.grow(aabb, oracle.bounding_radius_spherocylinders(r, L), 0.2)
    b = box if periodic else None
    p_cell = oracle.search(kind, lo, hi, c, R, box=b, symmetric=symmetric, method="cell")
    p_brute = oracle.search(kind, lo, hi, c, R, box=b, symmetric=symmetric, method="brute")
    assert len(p_brute) > 500
    np.testing.assert_array_equal(p_cell, p_brute)
    # canonical order: sorted by (i, j)
    key = p_cell[:, 0].astype(np.int64) * 10 ** 6 + p_cell[:, 1]
    assert np.all(np.diff(key) > 0)
    if symmetric:
        s = set(map(tuple, p_cell))
        assert all((j, i) in s for (i, j) in s) and all(i != j for (i, j) in s)
    else:
        assert np.all(p_cell[:, 0] < p_cell[:, 1])


def test_two_coincident_spheres(oracle):
    # UnitTestGenNeighborLinks.cpp:73-152: two coincident spheres -> exactly one link (unique) / both orientations
    c = np.zeros((2, 3))
    r = np.ones(2)
    lo, hi, R = oracle.grow(oracle.compute_aabb_spheres(c, r), r, 0.0)
    for kind in (0, 1):
        assert oracle.search(kind, lo, hi, c, R).tolist() == [[0, 1]]
        assert oracle.search(kind, lo, hi, c, R, symmetric=True).tolist() == [[0, 1], [1, 0]]


def test_touching_is_a_hit_and_edges(oracle):
    # closed predicate (AABB.hpp:420-431): touching boxes / spheres intersect; empty and single inputs give no pairs
    c = np.array([[0.0, 0, 0], [2.0, 0, 0], [4.5, 0, 0]])
    r = np.ones(3)
    lo, hi, R = oracle.grow(oracle.compute_aabb_spheres(c, r), r, 0.0)
    for kind in (0, 1):
        assert oracle.search(kind, lo, hi, c, R).tolist() == [[0, 1]]
    assert len(oracle.search(0, lo[:1], hi[:1], c[:1], R[:1])) == 0
    assert len(oracle.search(1, lo[:0], hi[:0], c[:0], R[:0])) == 0


def test_rebuild_rule(oracle):
    # GenNeighborLinkers.hpp:603-615: rebuild iff some centre moved more than half the buffer
    c = np.zeros((5, 3))
    d = c.copy()
    d[3, 1] = 0.5
    assert not oracle.moved_too_much(d, c, 1.0)
    d[3, 1] = 0.5000001
    assert oracle.moved_too_much(d, c, 1.0)


def test_triclinic_brute_force_states_the_periodic_metric(oracle):
    # the triclinic checker of tests/test_gpu_broadphase.py: PeriodicMetric::sep (periodicity.hpp:304-307) restated in
    # numpy on every pair of a small system; and a diagonal cell is the orthorhombic box
    rng = np.random.default_rng(4)
    n = 400
    cell = np.array([[8.0, 2.5, -1.5], [0.0, 7.0, 3.0], [0.0, 0.0, 6.5]])
    c = rng.uniform(0, 1, (n, 3)) @ cell.T + rng.integers(-1, 2, (n, 3)) @ cell.T
    R = rng.uniform(0.3, 0.9, n)
    lo, hi = c - R[:, None], c + R[:, None]
    got = oracle.search(oracle.SEARCH_SPHERES, lo, hi, c, R, box=cell, method="brute")
    hinv = np.linalg.inv(cell)
    i, j = np.triu_indices(n, 1)
    f = (c[j] - c[i]) @ hinv.T
    s = (f - np.round(f)) @ cell.T
    hit = (s * s).sum(axis=1) <= (R[i] + R[j]) ** 2
    exp = np.stack([i[hit], j[hit]], axis=1).astype(np.int32)
    assert len(exp) > n
    np.testing.assert_array_equal(got, exp)
    box = np.array([8.0, 7.0, 6.5])
    np.testing.assert_array_equal(oracle.search(oracle.SEARCH_AABB, lo, hi, c, R, box=np.diag(box), method="brute"),
                                  oracle.search(oracle.SEARCH_AABB, lo, hi, c, R, box=box, method="brute"))
