"""GPU parity on ALIGNED rods (nematic packings, growing colonies: scrap/.../Bacteria.cpp): the rod-compressed contact
operator -- the default of the spherocylinder pipeline -- against the reference's contact points as vector arms
(scrap/.../SpherocylinderSpherocylinderLinker.cpp:218-247), where the colinear branch of the segment-segment distance
hands back an UNCLAMPED parameter (PointLineSegment.hpp:156-166, LineSegmentLineSegment.hpp:236-265).
Bars: contact records and the operator in rod association BIT-EXACT against the oracle; against the vector-arm form
1e-12 of the result scale for the operator and the body velocities; LCP: the solution solves the vector-arm problem to
the tolerance, and equals the oracle's rod-form solve (same sums definition) in iteration count."""
import numpy as np
import pytest

import aligned_rods as ar

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import torch
    assert torch.cuda.is_available()
    from mundy_amd import ops as o
    return o


def _gpu_contacts(ops, P):
    from gpu_util import dev
    return ops.contact_spherocylinders(dev(P["pairs"]), dev(P["seg"]), dev(P["center"]), want_points=True)


def _rod_op(ops, P, s=None, t=None, dt=5e-3):
    from gpu_util import dev
    return ops.ContactOperator(dev(P["pairs"]), dev(P["normal"]), dev(P["mt"]), dt, mob_rot=dev(P["mr"]),
                               rod=(dev(P["s"] if s is None else s), dev(P["t"] if t is None else t), dev(P["seg"])))


def _vec_op(ops, P, dt=5e-3):
    from gpu_util import dev
    return ops.ContactOperator(dev(P["pairs"]), dev(P["normal"]), dev(P["mt"]), dt, ra=dev(P["ra"]), rb=dev(P["rb"]),
                               mob_rot=dev(P["mr"]))


def test_two_end_to_end_rods(ops, oracle):
    # the round-2 review's case: raw t = -0.25, contact point = the end of the rod
    from gpu_util import assert_bits_equal, dev, host
    P = ar.problem(oracle, ar.two_rods(), buffer=1.0)
    seg = P["seg"]
    raw = ops.distance_segment_segment(dev(seg[:1, 0:3]), dev(seg[:1, 3:6]), dev(seg[1:, 0:3]), dev(seg[1:, 3:6]))
    assert float(raw[3][0]) == 1.0 and float(raw[4][0]) == -0.25          # rows a14 / a15 keep the raw parameter
    G = _gpu_contacts(ops, P)
    for k in ("sep", "normal", "ra", "rb", "s", "t", "cp1", "cp2"):
        assert_bits_equal(host(G[k]), P[k], k)
    assert float(G["s"][0]) == 1.0 and float(G["t"][0]) == 0.0
    np.testing.assert_array_equal(host(G["rb"])[0], [0.0, 0.0, -1.0])
    x = dev(np.array([1.0]))
    yv = oracle.contact_op_apply(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3, np.array([1.0]), 2)
    for t in (P["t"], np.array([-0.25])):                                     # the operator clamps what it is given
        op = _rod_op(ops, P, t=t)
        np.testing.assert_allclose(host(op.apply(x)), yv, rtol=1e-14)
        lam, g, res = ops.solve_lcp(op, dev(P["sep"]), dev(np.zeros(1)), ops.PGDConfig(max_iters=100, tol=1e-10))
        # one contact: lambda = -sep / A  with the VECTOR-arm A (3.29 in the review's units, not 2.61)
        assert res.converged
        np.testing.assert_allclose(host(lam), -P["sep"] / yv, rtol=1e-12)
        op.close()


@pytest.mark.parametrize("make,n", [(lambda n: ar.nematic(n, 7), 20000),
                                    (lambda n: ar.nematic(n, 8, axis=(1.0, 2.0, 3.0)), 20000),
                                    (lambda n: ar.half_nematic(n, 9), 20000)])
def test_rod_operator_on_aligned_packing_equals_vector_arm_form(ops, oracle, make, n):
    from gpu_util import assert_bits_equal, dev, host
    P = ar.problem(oracle, make(n))
    C, N = len(P["pairs"]), P["N"]
    ends = (P["s"] == 0) | (P["s"] == 1) | (P["t"] == 0) | (P["t"] == 1)
    assert C > n and ends.mean() > 0.25
    # contact records: bit-exact, arclengths of the contact points in [0, 1]
    G = _gpu_contacts(ops, P)
    for k in ("sep", "normal", "ra", "rb", "s", "t", "cp1", "cp2"):
        assert_bits_equal(host(G[k]), P[k], k)
    u = P["seg"][:, 3:6] - P["seg"][:, 0:3]
    i, j = P["pairs"][:, 0], P["pairs"][:, 1]
    np.testing.assert_allclose((P["s"] - 0.5)[:, None] * u[i], P["ra"], rtol=0, atol=1e-12)
    np.testing.assert_allclose((P["t"] - 0.5)[:, None] * u[j], P["rb"], rtol=0, atol=1e-12)
    # operator and body velocities: rod form vs the vector-arm oracle at 1e-12; vs the oracle's rod association bitwise
    rng = np.random.default_rng(0)
    xh = rng.uniform(0, 1, C)
    op = _rod_op(ops, P)
    y = host(op.apply(dev(xh)))
    yv = oracle.contact_op_apply(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3, xh, N)
    np.testing.assert_allclose(y, yv, rtol=0, atol=1e-12 * np.abs(yv).max())
    vv = ar.body_velocity_vector_form(P, xh)
    np.testing.assert_allclose(host(op.body_velocity()), vv, rtol=0, atol=1e-12 * np.abs(vv).max())
    with oracle.compensated_sums():
        yr = oracle.contact_op_apply(P["pairs"], P["normal"], None, None, P["mt"], P["mr"], 5e-3, xh, N,
                                     rod=(P["s"], P["t"], P["seg"]))
    assert_bits_equal(y, yr, "A x, rod association")
    # the GPU's own vector-arm operator agrees too
    opv = _vec_op(ops, P)
    np.testing.assert_allclose(host(opv.apply(dev(xh))), y, rtol=0, atol=1e-12 * np.abs(yv).max())
    # LCP
    tol, lcp = 1e-6, (oracle.LOWER_BOUND, 0.0, 0.0)
    cfg = ops.PGDConfig(max_iters=20000, tol=tol)
    lam, g, res = ops.solve_lcp(op, dev(P["sep"]), dev(np.zeros(C)), cfg)
    assert res.converged
    lam_h = host(lam)
    vel = host(op.body_velocity())
    with oracle.compensated_sums():
        xo, go, ro = oracle.solve_cqpp_contact(P["pairs"], P["normal"], None, None, P["mt"], P["mr"], 5e-3, P["sep"],
                                               np.zeros(C), max_iters=20000, tol=tol, rod=(P["s"], P["t"], P["seg"]))
        xv, gv, rv = oracle.solve_cqpp_contact(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3,
                                               P["sep"], np.zeros(C), max_iters=20000, tol=tol)
    assert ro["converged"] and rv["converged"]
    assert abs(res.num_iters - ro["num_iters"]) <= 2, (res.num_iters, ro["num_iters"])
    np.testing.assert_allclose(host(g), go, rtol=0, atol=20 * tol)
    # ... and solves the reference's (vector-arm) problem to the tolerance
    gx = oracle.contact_op_apply(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3, lam_h, N) + P["sep"]
    assert oracle.residual(oracle.RESID_PROJECTED_DIFF, lam_h, gx, lcp) < tol * (1 + 1e-6)
    vref = ar.body_velocity_vector_form(P, xv)
    np.testing.assert_allclose(vel, ar.body_velocity_vector_form(P, lam_h), rtol=0, atol=1e-12 * np.abs(vref).max())
    np.testing.assert_allclose(vel, vref, rtol=0, atol=1e-5 * np.abs(vref).max())
    if res.num_iters == rv["num_iters"]:   # same path through the BB iteration: rounding level (see the CPU test)
        np.testing.assert_allclose(lam_h, xv, rtol=0, atol=1e-8)
        np.testing.assert_allclose(vel, vref, rtol=0, atol=1e-9 * np.abs(vref).max())
    op.close()
    opv.close()


def test_cold_tier_on_aligned_packing_is_bit_identical(ops, oracle):
    # the tier's wake-up bound relies on |arm coefficient| <= 1/2 (convex.hip, k_body drift bookkeeping): aligned
    # rods with raw parameters outside [0, 1] used to break it.  Tiered (forced: mode 3) == untiered, bit for bit,
    # also when the operator is handed the RAW parameters of the distance routine.
    import torch
    from gpu_util import dev
    P = ar.problem(oracle, ar.nematic(40000, 11))
    C = len(P["pairs"])
    assert C >= 65536, C
    seg = P["seg"]
    i, j = P["pairs"][:, 0], P["pairs"][:, 1]
    raw = ops.distance_segment_segment(dev(seg[i, 0:3]), dev(seg[i, 3:6]), dev(seg[j, 0:3]), dev(seg[j, 3:6]))
    s_raw, t_raw = raw[3].cpu().numpy(), raw[4].cpu().numpy()
    assert ((s_raw < 0) | (s_raw > 1) | (t_raw < 0) | (t_raw > 1)).mean() > 0.05
    q = dev(P["sep"])
    out = {}
    for name, (s, t) in (("clamped", (None, None)), ("raw", (s_raw, t_raw))):
        for mode in (0, 3):
            op = _rod_op(ops, P, s=s, t=t)
            op.set_tiering(mode)
            st = tuple(dev(np.zeros(C)) for _ in range(4))
            _, _, res = ops.solve_lcp(op, q, None, ops.PGDConfig(max_iters=20000, tol=1e-6), state=st)
            stats = op.tier_stats()
            assert res.converged
            assert (stats["tiered_iterations"] > 0) == (mode == 3), stats
            out[name, mode] = (st, res, op.body_velocity().clone())
            op.close()
    ref = out["clamped", 0]
    for key, (st, res, vel) in out.items():
        assert (res.num_iters, res.residual) == (ref[1].num_iters, ref[1].residual), key
        for a, b in zip(st, ref[0]):
            assert torch.equal(a, b), key
        assert torch.equal(vel, ref[2]), key
