"""The oracle's two summation modes (mundy_oracle.hpp, SumMode) and its rod-axis form of the spherocylinder operator.

SUM_SERIAL is the reference's arithmetic (Kokkos-Serial order).  SUM_COMPENSATED carries every sum that feeds the
Barzilai-Borwein step as a double-double pair and rounds once -- the definition of the sums on the device path.  Checked
here, on the CPU: the compensated sums do not depend on the order of the contacts (bitwise), agree with the serial ones
to rounding, equal math.fsum (the correctly rounded exact sum) on adversarial inputs, and leave the solver's answers
where they were; the rod-axis operator equals the vector-arm one to rounding.
"""
import math

import numpy as np
import pytest


def _rod_problem(oracle, n, seed, buffer=0.1):
    from mundy_amd import synth
    b = synth.spherocylinders(n, seed=seed)
    c = b["center"]
    aabb = oracle.compute_aabb_spherocylinders(c, b["quat"], b["radius"], b["length"])
    brad = oracle.bounding_radius_spherocylinders(b["radius"], b["length"])
    lo, hi, R = oracle.grow(aabb, brad, buffer)
    pairs = oracle.search(1, lo, hi, c, R)
    seg = oracle.spherocylinder_segments(c, b["quat"], b["radius"], b["length"])
    out = oracle.contact_spherocylinders(pairs, seg, c)
    mt, mr = synth.dry_mobility(b["radius"], bounding_radius=brad)
    return dict(N=n, pairs=pairs, sep=out["sep"], normal=out["normal"], ra=out["ra"], rb=out["rb"], mt=mt, mr=mr,
                s=out["s"], t=out["t"], seg=seg)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_compensated_dots_are_correctly_rounded(oracle):
    rng = np.random.default_rng(0)
    # terms spread over 30 orders of magnitude with heavy cancellation: plain sums lose everything, pairs do not
    n = 20001
    x = rng.normal(size=n) * 10.0 ** rng.uniform(-15, 15, n)
    y = np.zeros(n)
    exact = math.fsum((x - y) ** 2)
    with oracle.compensated_sums():
        got = oracle.diff_dot2(x, y)
        perm = rng.permutation(n)
        got_p = oracle.diff_dot2(x[perm], y[perm])
    assert got == exact and got_p == exact
    x2, y2 = rng.normal(size=n), rng.normal(size=n)
    a = x2 * 10.0 ** rng.uniform(-8, 8, n)
    b = y2 * 10.0 ** rng.uniform(-8, 8, n)
    terms = (a - 0.0) * (b - 0.0)
    exact4 = math.fsum(terms)
    with oracle.compensated_sums():
        got4 = oracle.diff_dot4(a, np.zeros(n), b, np.zeros(n))
        perm = rng.permutation(n)
        got4p = oracle.diff_dot4(a[perm], np.zeros(n), b[perm], np.zeros(n))
    assert got4 == got4p
    assert abs(got4 - exact4) <= 2.0 ** -52 * abs(exact4)  # cancelling sum: within an ulp, and order independent
    # the mode is restored on exit, and the serial sum is the plain left-to-right one
    acc = 0.0
    for v in (x - y) ** 2:
        acc += v
    assert oracle.diff_dot2(x, y) == acc


@pytest.mark.parametrize("rod", [False, True])
def test_compensated_operator_is_order_independent(oracle, rod):
    P = _rod_problem(oracle, 1500, seed=4)
    rng = np.random.default_rng(1)
    C = len(P["pairs"])
    x = rng.uniform(0, 1, C) * (rng.random(C) < 0.4)
    kw = dict(rod=(P["s"], P["t"], P["seg"])) if rod else {}
    perm = rng.permutation(C)
    Q = {k: np.ascontiguousarray(P[k][perm]) for k in ("pairs", "normal", "ra", "rb", "s", "t")}
    kwq = dict(rod=(Q["s"], Q["t"], P["seg"])) if rod else {}
    y_serial = oracle.contact_op_apply(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3, x, P["N"], **kw)
    with oracle.compensated_sums():
        y = oracle.contact_op_apply(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3, x, P["N"], **kw)
        yq = oracle.contact_op_apply(Q["pairs"], Q["normal"], Q["ra"], Q["rb"], P["mt"], P["mr"], 5e-3, x[perm], P["N"],
                                     **kwq)
    assert np.array_equal(_bits(yq), _bits(y[perm]))          # any contact order, same bits
    np.testing.assert_allclose(y, y_serial, rtol=1e-12, atol=1e-13 * np.abs(y_serial).max())
    # serial sums do depend on the order (this is what moved the iteration count before)
    yq_serial = oracle.contact_op_apply(Q["pairs"], Q["normal"], Q["ra"], Q["rb"], P["mt"], P["mr"], 5e-3, x[perm],
                                        P["N"], **kwq)
    assert not np.array_equal(_bits(yq_serial), _bits(y_serial[perm]))


def test_rod_axis_form_equals_vector_arms(oracle):
    P = _rod_problem(oracle, 2000, seed=9)
    rng = np.random.default_rng(2)
    x = rng.uniform(0, 1, len(P["pairs"]))
    yv = oracle.contact_op_apply(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3, x, P["N"])
    yr, vel = oracle.contact_op_apply(P["pairs"], P["normal"], None, None, P["mt"], P["mr"], 5e-3, x, P["N"],
                                      rod=(P["s"], P["t"], P["seg"]), body_velocity=True)
    np.testing.assert_allclose(yr, yv, rtol=1e-12, atol=1e-12 * np.abs(yv).max())
    assert vel.shape == (P["N"], 6) and np.abs(vel[:, 3:]).max() > 0


@pytest.mark.parametrize("rod", [False, True])
def test_solver_same_answer_in_both_modes_and_count_independent_of_order(oracle, rod):
    P = _rod_problem(oracle, 1200, seed=13)
    C = len(P["pairs"])
    tol = 1e-6
    kw = dict(rod=(P["s"], P["t"], P["seg"])) if rod else {}
    args = (P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3, P["sep"], np.zeros(C))
    xs, gs, rs = oracle.solve_cqpp_contact(*args, max_iters=10000, tol=tol, **kw)
    with oracle.compensated_sums():
        xc, gc, rc = oracle.solve_cqpp_contact(*args, max_iters=10000, tol=tol, **kw)
        perm = np.random.default_rng(3).permutation(C)
        Q = {k: np.ascontiguousarray(P[k][perm]) for k in ("pairs", "normal", "ra", "rb", "s", "t", "sep")}
        kwq = dict(rod=(Q["s"], Q["t"], P["seg"])) if rod else {}
        xp, gp, rp = oracle.solve_cqpp_contact(Q["pairs"], Q["normal"], Q["ra"], Q["rb"], P["mt"], P["mr"], 5e-3,
                                               Q["sep"], np.zeros(C), max_iters=10000, tol=tol, **kwq)
    assert rs["converged"] and rc["converged"] and rp["converged"]
    np.testing.assert_allclose(gc, gs, atol=20 * tol)          # the LCP's gradient is unique
    # permuting the contacts permutes the iterates and nothing else: same count, same bits
    assert rp["num_iters"] == rc["num_iters"] and rp["residual"] == rc["residual"]
    assert np.array_equal(_bits(xp), _bits(xc[perm])) and np.array_equal(_bits(gp), _bits(gc[perm]))
