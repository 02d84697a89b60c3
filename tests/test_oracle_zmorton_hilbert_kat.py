"""Pins the oracle's Z-order comparator and Hilbert generator on the reference's KATs:
mundy/math/tests/unit_tests/UnitTestZMorton.cpp:217-380 and UnitTestHilbert.cpp:48-387 (fixture hilbert_kat.json,
numbers extracted by tests/golden/make_reference_kats.py)."""
import json
import os

import numpy as np
import pytest

INT_MIN = -2 ** 31
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_float_exp_sig_log(oracle):
    # ZMortonFloatExp.PowerOfTwo / FloatSig.OneOverPowerOfTwo / UIntLogBase2.PowerOfTwo (UnitTestZMorton.cpp:262-300)
    for e in range(-126, 128):
        assert oracle.float_exp(np.float32(2.0) ** np.float32(e), single=True) == e
    for e in range(-1022, 1024):
        assert oracle.float_exp(2.0 ** e) == e
    for i in range(23, 0, -1):
        assert oracle.float_sig(np.float32(1.0) + np.float32(2.0 ** -i), single=True) == 1 << (23 - i)
    for i in range(52, 0, -1):
        assert oracle.float_sig(1.0 + 2.0 ** -i) == 1 << (52 - i)
    for i in range(64):
        assert oracle.uint_log_base2(1 << i) == i
        assert oracle.uint_log_base2((2 << i) - 1) == i


XOR_MSB = [  # UnitTestZMorton.cpp:317-379
    (1.0, 1.0, INT_MIN), (42.6666641235, 42.6666641235, INT_MIN), (1.0, -1.0, INT_MIN),
    (42.6666641235, -42.6666641235, INT_MIN),
    (0.5, 1.0, 0), (0.5, 1.5, 0), (0.5, 0.125, -1), (0.25, 0.125, -2), (1.0, 2.0, 1), (2.0, 4.0, 2),
    (1.0, 1.5, -1), (1.0, 1.75, -1), (1.0, 1.875, -1), (1.0, 1.375, -2), (1.0, 1.125, -3), (1.75, 1.875, -3),
    (0.5, 0.5625, -4), (0.21875, 0.234375, -6), (16.0, 18.0, 1), (24.0, 26.0, 1), (28.0, 30.0, 1), (56.0, 60.0, 2),
    (112.0, 120.0, 3), (80.0, 88.0, 3), (160.0, 176.0, 4), (384.0, 448.0, 6), (1.0, 1.0 + 1.1921e-7, -23),
    (1.0, 1.0 + 3.5763e-7, -22), (1.0, 1.0 + 2.3842e-7, -22), (1.0, 1.0 + 4.7684e-7, -21),
]


@pytest.mark.parametrize("p,q,e", XOR_MSB)
def test_float_xor_msb(oracle, p, q, e):
    assert oracle.float_xor_msb(p, q) == e
    assert oracle.float_xor_msb(float(np.float32(p)), float(np.float32(q)), single=True) == e


def test_float_xor_msb_double_only(oracle):
    assert oracle.float_xor_msb(1.0, 1.0 + 2.2204460e-16) == -52
    assert oracle.float_xor_msb(1.0, 1.0 + 4.4408921e-16) == -51


def _sort_zorder_ref(points):
    """recursive bounding-box split sort of UnitTestZMorton.cpp:99-165 (the reference's own test oracle)"""
    pts = [tuple(p) for p in points]
    d = len(pts[0])
    dtype = points.dtype.type
    bound = dtype(2.0) * dtype(2.0) ** dtype(np.ceil(np.log2(np.abs(points).max())))

    def rec(items, k, lo, hi):
        if len(items) <= 1:
            return items
        split = dtype(0.5) * (lo[k] + hi[k])
        lower = [p for p in items if lo[k] <= p[k] < split]
        upper = [p for p in items if split < p[k] <= hi[k]]
        on = [p for p in items if p[k] == split]
        if split > 0:
            upper = on + upper  # points on the plane go to the upper half-space when split > 0
        else:
            lower = lower + on
        k1 = (k + d - 1) % d
        lhi, ulo = list(hi), list(lo)
        lhi[k], ulo[k] = split, split
        return rec(lower, k1, lo, lhi) + rec(upper, k1, ulo, hi)

    return np.array(rec(pts, d - 1, [-bound] * d, [bound] * d), dtype=points.dtype)


@pytest.mark.parametrize("d", [2, 3, 4, 6])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_zorder_less_sort_equals_bbox_split_sort(oracle, d, dtype):
    # Less.Random{2,3,4,6}D (UnitTestZMorton.cpp:197-215), 2000 points per case
    rng = np.random.default_rng(d)
    pts = rng.uniform(-8.0, 8.0, (2000, d)).astype(dtype)
    order = oracle.zorder_argsort(pts)
    np.testing.assert_array_equal(pts[order], _sort_zorder_ref(pts))


def test_zmorton_less_vs_zorder_less(oracle):
    # the two comparators agree whenever no two axes tie on the XOR-MSB exponent; positive octant,
    # distinct exponents per axis
    rng = np.random.default_rng(2)
    n_checked = 0
    for _ in range(2000):
        p, q = rng.uniform(0.0, 8.0, 3), rng.uniform(0.0, 8.0, 3)
        e = [oracle.float_xor_msb(p[k], q[k]) for k in range(3)]
        if len(set(e)) == 3:
            assert oracle.zmorton_less(p, q) == oracle.zorder_less(p, q)
            n_checked += 1
    assert n_checked > 100
    assert oracle.zmorton_less([-1.0, 2.0, 3.0], [1.0, 2.0, 3.0]) is True   # sign difference on x decides
    assert oracle.zmorton_less([5.0, 2.0, 3.0], [1.0, 2.0, -3.0]) is False  # z sign difference outranks x


@pytest.fixture(scope="module")
def hilbert_kat():
    with open(os.path.join(GOLD, "hilbert_kat.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("s", [2, 4, 8])
def test_hilbert_3d_positions(oracle, hilbert_kat, s):
    np.testing.assert_allclose(oracle.hilbert_3d(s), hilbert_kat["Cube%d" % s]["positions"], atol=1e-12)


@pytest.mark.parametrize("links", [8, 9])
def test_hilbert_directors(oracle, hilbert_kat, links):
    kat = hilbert_kat["DirectorLinks%d" % links]
    pos, dirs = oracle.hilbert_positions_and_directors(links)
    np.testing.assert_allclose(pos[: len(kat["positions"])], kat["positions"], atol=1e-12)
    np.testing.assert_allclose(dirs[: len(kat["directors"])], kat["directors"], atol=1e-12)
