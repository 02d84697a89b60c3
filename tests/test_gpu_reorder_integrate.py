"""Body reordering and time integration (8f.3) through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from mundy_amd import ops as o
    return o


def _morton_key(cells):
    """z-most-significant bit interleave of non-negative lattice coordinates (the order zorder_knn::Less gives them,
    mundy_math/zmort.hpp:195-220)"""
    key = np.zeros(len(cells), dtype=object)
    for bit in range(21):
        for axis, shift in ((0, 0), (1, 1), (2, 2)):
            key = key | (((cells[:, axis] >> bit) & 1).astype(object) << (3 * bit + shift))
    return key


@pytest.mark.parametrize("n", [1, 2, 1000, 200_003])
def test_morton_order_is_the_z_order_permutation(ops, oracle, n):
    from gpu_util import dev, host
    rng = np.random.default_rng(n)
    c = rng.uniform(0.0, 97.3, (n, 3))
    c[: min(n, 50)] = np.floor(c[: min(n, 50)])          # centres exactly on lattice planes
    cell, lo = (1.7 if n > 1000 else 7.0), [0.0, 0.0, 0.0]
    perm = host(ops.morton_order(dev(c), lo, cell)).astype(np.int64)
    assert np.array_equal(np.sort(perm), np.arange(n))                    # a permutation
    bits = 4                                   # the lattice resolution rule of mhip_morton_order (mundy_hip.h)
    while bits < 8 and (1 << (3 * (bits + 1))) <= 8 * n:
        bits += 1
    cells = np.clip(np.floor((c - np.array(lo)) * (1.0 / cell)).astype(np.int64), 0, (1 << bits) - 1)
    if n >= 1000:
        assert (cells < (1 << bits) - 1).mean() > 0.2     # the case is not all boundary cells
    key = _morton_key(cells)[perm]
    assert all(key[i] <= key[i + 1] for i in range(n - 1))                # keys non-decreasing along the order
    ties = [i for i in range(n - 1) if key[i] == key[i + 1]]
    assert all(perm[i] < perm[i + 1] for i in ties)                       # ties by index: deterministic
    assert np.array_equal(host(ops.morton_order(dev(c), lo, cell)), perm)
    # consistent with the reference comparator zorder_knn::Less on the lattice coordinates (zmort.hpp:195-220: axes
    # scanned z, y, x with a strict <, so z wins ties of the differing bit level -- the bit interleave above), sampled.
    # (mundy::math::zmorton_less, :228-265, scans x, y, z and lets x win such ties; the two agree on generic floats,
    # as the reference's own test checks, but not on small integers.)
    cp = cells[perm].astype(np.float64)
    for i in rng.integers(0, max(n - 1, 1), size=min(n - 1, 300)):
        if key[i] != key[i + 1]:
            assert oracle.zorder_less(cp[i], cp[i + 1]) and not oracle.zorder_less(cp[i + 1], cp[i])


def test_gather_rows(ops):
    from gpu_util import assert_bits_equal, dev, host
    import torch
    rng = np.random.default_rng(3)
    n = 10_007
    perm = rng.permutation(n).astype(np.int32)
    for width in (1, 3, 4, 6, 8):
        src = rng.normal(size=(n, width))
        assert_bits_equal(host(ops.gather_rows(dev(perm), dev(src))), src[perm], "gather_rows width %d" % width)
    sub = perm[:100]
    assert_bits_equal(host(ops.gather_rows(dev(sub), dev(src))), src[sub], "gather_rows subset")
    assert ops.gather_rows(dev(perm[:0]), dev(src)).shape[0] == 0


def test_integrate_euler_vs_oracle(ops, oracle):
    # x += dt U (NgpLcp.cpp:898) is bit-exact; rotate_quaternion (Quaternion.hpp:1366-1383) goes through sin / cos: 4 ulp
    # of a unit quaternion component against libm, bit-exact against the oracle evaluating the device's sin / cos
    from gpu_util import assert_bits_equal, dev, host
    rng = np.random.default_rng(5)
    n = 50_000
    c = rng.uniform(-50, 50, (n, 3))
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    v = rng.normal(size=(n, 6)) * rng.uniform(1e-3, 30.0, (n, 1))
    v[:100, 3:] = 0.0                       # no rotation: the quaternion must come back untouched
    v[100:200, 3:] *= 1e-17                 # |omega| below the 1e-15 tolerance: same
    for dt in (5e-3, 0.1):
        dc, dq = dev(c), dev(q)
        ops.integrate_euler(dt, dev(v), dc, dq)
        co, qo = oracle.integrate_euler(dt, v, c, q)
        assert_bits_equal(host(dc), co, "Euler update of the centres")
        assert_bits_equal(host(dq)[:200], q[:200], "omega ~ 0 leaves the orientation untouched")
        np.testing.assert_allclose(host(dq), qo, rtol=0, atol=1e-15)    # oracle with libm sin / cos
        with oracle.shared_trig():                                      # oracle with the device's sin / cos: exact
            assert_bits_equal(host(dq), oracle.integrate_euler(dt, v, c, q)[1], "rotate_quaternion")
        np.testing.assert_allclose(np.linalg.norm(host(dq), axis=1), 1.0, atol=4e-16)
    ds = dev(c)
    ops.integrate_euler(5e-3, dev(v), ds)   # spheres: no orientation
    assert_bits_equal(host(ds), oracle.integrate_euler(5e-3, v, c)[0], "Euler update, spheres")


@pytest.mark.parametrize("level", [3, 6, 7])
def test_hilbert_curve_order_equals_the_host_partitioner(ops, level):
    # mhip_curve_order with the key table of mundy::math::hilbert_3d (generated by the recursion of Hilbert.hpp:48-83,
    # pinned on the reference's KATs in tests/test_oracle_zmorton_hilbert_kat.py) == the host-side Hilbert order the
    # domain decomposition uses, permutation for permutation
    from gpu_util import dev, host
    from mundy_amd import distributed as D
    rng = np.random.default_rng(level)
    n = 150_000
    lo, hi = np.array([-3.0, 0.5, 2.0]), np.array([40.0, 61.5, 33.0])
    c = rng.uniform(lo - 1.0, hi + 1.0, (n, 3))          # some points outside the box: clamped to the border cells
    c[:200] = lo + (hi - lo) * rng.integers(0, 1 << level, (200, 3)) / float(1 << level)   # exactly on cell faces
    table = D.hilbert_key_table(level)
    perm = host(ops.curve_order(dev(c), lo, hi, level, dev(table.astype(np.int32))))
    ref = D.hilbert_order(c, lo, hi, level=level)
    np.testing.assert_array_equal(perm, ref)
