"""CPU: the oracle's rod-compressed operator (ContactOpRod) against its vector-arm form (ContactOp, the reference's
contact points) on aligned rods, where the distance routine's raw parameters leave [0, 1]."""
import numpy as np
import pytest

import aligned_rods as ar


def test_raw_parameter_stays_unclamped_and_contact_arclength_is_clamped(oracle):
    # rows a14 / a15 keep the reference's raw parameter; the contact assembly (a16) reports the contact point's
    P = ar.problem(oracle, ar.two_rods(), buffer=1.0)
    seg = P["seg"]
    d, _, _, s, t, _ = oracle.distance_segment_segment(seg[:1, 0:3], seg[:1, 3:6], seg[1:, 0:3], seg[1:, 3:6])
    assert s[0] == 1.0 and t[0] == -0.25                              # LineSegmentLineSegment.hpp:236-265
    assert P["s"][0] == 1.0 and P["t"][0] == 0.0                      # arclengths of cp1 = p1, cp2 = q0
    np.testing.assert_array_equal(P["ra"][0], [0.0, 0.0, 1.0])
    np.testing.assert_array_equal(P["rb"][0], [0.0, 0.0, -1.0])       # not (t - 1/2) u = (0, 0, -1.5)
    u = seg[:, 3:6] - seg[:, 0:3]
    np.testing.assert_array_equal((P["s"][0] - 0.5) * u[0], P["ra"][0])
    np.testing.assert_array_equal((P["t"][0] - 0.5) * u[1], P["rb"][0])


def test_rod_operator_clamps_whatever_it_is_given(oracle):
    # the operator itself defines the arm from the clamped arclength: raw parameters give the same operator
    P = ar.problem(oracle, ar.two_rods(), buffer=1.0)
    x = np.array([1.0])
    yv = oracle.contact_op_apply(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3, x, 2)
    for t in (P["t"], np.array([-0.25])):
        yr = oracle.contact_op_apply(P["pairs"], P["normal"], None, None, P["mt"], P["mr"], 5e-3, x, 2,
                                     rod=(P["s"], t, P["seg"]))
        np.testing.assert_allclose(yr, yv, rtol=1e-14)


@pytest.mark.parametrize("make", [lambda: ar.nematic(3000, 7), lambda: ar.nematic(3000, 8, axis=(1.0, 2.0, 3.0)),
                                  lambda: ar.half_nematic(3000, 9)])
def test_rod_form_equals_vector_form_on_aligned_rods(oracle, make):
    P = ar.problem(oracle, make())
    C = len(P["pairs"])
    assert C > 3000
    assert np.all((P["s"] >= 0) & (P["s"] <= 1) & (P["t"] >= 0) & (P["t"] <= 1))
    ends = (P["s"] == 0) | (P["s"] == 1) | (P["t"] == 0) | (P["t"] == 1)
    assert ends.mean() > 0.25                                         # colinear-branch pairs are there in numbers
    x = np.random.default_rng(0).uniform(0, 1, C)
    rod = (P["s"], P["t"], P["seg"])
    yv = oracle.contact_op_apply(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3, x, P["N"])
    yr, vr = oracle.contact_op_apply(P["pairs"], P["normal"], None, None, P["mt"], P["mr"], 5e-3, x, P["N"], rod=rod,
                                     body_velocity=True)
    np.testing.assert_allclose(yr, yv, rtol=0, atol=1e-12 * np.abs(yv).max())
    vv = ar.body_velocity_vector_form(P, x)
    np.testing.assert_allclose(vr, vv, rtol=0, atol=1e-12 * np.abs(vv).max())
    sol = {}
    for name, r in (("vector", None), ("rod", rod)):
        with oracle.compensated_sums():
            sol[name] = oracle.solve_cqpp_contact(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3,
                                                  P["sep"], np.zeros(C), max_iters=20000, tol=1e-6, rod=r)
    (xv, gv, rv), (xr, gr, rr) = sol["vector"], sol["rod"]
    assert rv["converged"] and rr["converged"]
    # the yardstick is the vector-arm (reference) problem: the rod-form solution must solve IT to the tolerance
    tol, lcp = 1e-6, (oracle.LOWER_BOUND, 0.0, 0.0)
    gx = oracle.contact_op_apply(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3, xr, P["N"]) + P["sep"]
    assert oracle.residual(oracle.RESID_PROJECTED_DIFF, xr, gx, lcp) < tol
    vv, vr = ar.body_velocity_vector_form(P, xv), ar.body_velocity_vector_form(P, xr)
    # body velocities are the unique part of the solution (A = dt D^T M D is only semi-definite in the multipliers)
    np.testing.assert_allclose(vr, vv, rtol=0, atol=1e-5 * np.abs(vv).max())
    if rv["num_iters"] == rr["num_iters"]:
        # two associations of the same arms, same path through the BB iteration: the iterates differ by rounding that
        # ~150 steps amplify (observed 3e-10 of multipliers up to 18, 8e-9 of velocities up to 200)
        np.testing.assert_allclose(xr, xv, rtol=0, atol=1e-8)
        np.testing.assert_allclose(gr, gv, rtol=0, atol=1e-9)
        np.testing.assert_allclose(vr, vv, rtol=0, atol=1e-9 * np.abs(vv).max())
