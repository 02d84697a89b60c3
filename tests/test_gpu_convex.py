"""GPU parity, BBPGD: S1 vector kernels, the contact operator and the fused / unfused / dense drivers against the CPU
oracle and the reference's own test problems (UnitTestConvex.cpp).  Tolerances: element-wise kernels and the operator
apply are BIT-EXACT; the sums that feed the BB step are double-double on the device (order independent) and are
compared with the oracle's compensated mode (same definition of the rounding): equal values, equal iteration counts
(asserted within 2); against the oracle's plain serial sums they differ by rounding only (rel 1e-12).  Solutions within
the reference's 10*tol (UnitTestConvex.cpp:559)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import torch
    assert torch.cuda.is_available()
    from mundy_amd import ops as o
    return o


def test_vector_kernels(ops, oracle):
    from gpu_util import assert_bits_equal, dev, host
    rng = np.random.default_rng(0)
    for n in (1, 255, 256, 257, 1_000_003):
        x, y = rng.normal(size=n), rng.normal(size=n)
        for alpha, beta in ((1.5, -0.5), (1e-16, 2.0), (3.0, 1e-16), (1e-17, -1e-17)):
            yo = y.copy()
            oracle.axpby(alpha, x, beta, yo)
            dy = dev(y)
            ops.axpby(alpha, dev(x), beta, dy)
            assert_bits_equal(host(dy), yo, "axpby")
            for space in ((0, 0.0, 0.0), (1, 0.1, 0.0), (2, 0.0, 0.2), (3, -0.3, 0.3)):
                zo = np.empty(n)
                oracle.wrapped_axpbyz(alpha, x, beta, y, zo, space)
                dz = dev(np.empty(n))
                ops.wrapped_axpbyz(alpha, dev(x), beta, dev(y), dz, space)
                assert_bits_equal(host(dz), zo, "wrapped_axpbyz")
        x2, y2 = rng.normal(size=n), rng.normal(size=n)
        assert ops.diff_dot(dev(x), dev(y)) == pytest.approx(oracle.diff_dot2(x, y), rel=1e-12)
        assert ops.diff_dot(dev(x), dev(x2), dev(y), dev(y2)) == pytest.approx(oracle.diff_dot4(x, x2, y, y2), rel=1e-11, abs=1e-9)
        with oracle.compensated_sums():   # double-double on both sides: the same correctly rounded sums, any order
            assert ops.diff_dot(dev(x), dev(y)) == oracle.diff_dot2(x, y)
            assert ops.diff_dot(dev(x), dev(x2), dev(y), dev(y2)) == oracle.diff_dot4(x, x2, y, y2)
            assert ops.bb_step(dev(x), dev(y), dev(x2), dev(y2)) == oracle.bb_step(x, y, x2, y2)
        for kind in (0, 1):
            for space in ((1, 0.0, 0.0), (3, -0.5, 0.5)):
                # max is order independent: exact
                assert ops.residual(kind, dev(x), dev(y), space) == oracle.residual(kind, x, y, space)
        assert ops.bb_step(dev(x), dev(y), dev(x2), dev(y2)) == pytest.approx(oracle.bb_step(x, y, x2, y2), rel=1e-9)


def test_vector_kernels_unaligned_views(ops, oracle):
    # the streaming kernels take 16-byte accesses when they can; views that start on an odd element (8-byte aligned
    # only) take the scalar path -- same element-wise results either way
    from gpu_util import assert_bits_equal, dev, host
    rng = np.random.default_rng(5)
    for n in (2, 7, 1000, 4097):
        x, y = rng.normal(size=n + 1), rng.normal(size=n + 1)
        for ox, oy in ((1, 0), (0, 1), (1, 1)):
            xs, ys = x[ox:ox + n], y[oy:oy + n]
            yo = ys.copy()
            oracle.axpby(0.75, xs, -1.25, yo)
            dx, dy = dev(x), dev(y)
            ops.axpby(0.75, dx[ox:ox + n], -1.25, dy[oy:oy + n])
            assert_bits_equal(host(dy)[oy:oy + n], yo, "axpby on an offset view")
            assert_bits_equal(host(dy)[:oy], y[:oy], "elements before the view untouched")
            assert_bits_equal(host(dy)[oy + n:], y[oy + n:], "elements after the view untouched")
            zo = np.empty(n)
            oracle.wrapped_axpbyz(1.0, xs, -0.5, ys, zo, (1, 0.0, 0.0))
            dz = dev(np.full(n + 1, 7.0))
            ops.wrapped_axpbyz(1.0, dev(x)[ox:ox + n], -0.5, dev(y)[oy:oy + n], dz[1:1 + n], (1, 0.0, 0.0))
            assert_bits_equal(host(dz)[1:], zo, "wrapped_axpbyz on offset views")
            assert host(dz)[0] == 7.0


A3 = np.array([[2.0, -1.0, 0.0], [-1.0, 2.0, -1.0], [0.0, -1.0, 2.0]])


@pytest.mark.parametrize("x_star,space", [([1.0, 0.0, 1.0], (0, 0.0, 0.0)), ([1.0, 0.0, 1.0], (3, 0.0, 2.0)),
                                          ([9.0, 9.0, 9.0], (3, 9.0, 10.0))])
def test_reference_analytic_problems_dense(ops, oracle, x_star, space):
    # UnitTestConvex.cpp:239-414 via :563-606: x0 = 99.99, max_iters 1000, tol 1e-6
    from gpu_util import dev, host
    x_star = np.array(x_star)
    q = -A3 @ x_star
    cfg = ops.PGDConfig(max_iters=1000, tol=1e-6)
    x, g, res = ops.solve_cqpp(dev(A3), dev(q), space, dev(np.full(3, 99.99)), cfg)
    assert res.converged and res.num_iters <= 1000
    np.testing.assert_allclose(host(x), x_star, atol=1e-5, rtol=0)
    xo, go, ro = oracle.solve_cqpp_dense(A3, q, space, np.full(3, 99.99), max_iters=1000, tol=1e-6)
    assert res.num_iters == ro["num_iters"]            # n = 3: one wave, same sums -> same trajectory
    np.testing.assert_allclose(host(x), xo, atol=1e-12)


@pytest.mark.parametrize("n", [3, 7, 200])
def test_reference_random_lcp_dense(ops, oracle, n):
    # UnitTestConvex.cpp:416-524, :617-625
    from gpu_util import dev, host
    from test_oracle_convex_kat import random_lcp
    A, q, x_star = random_lcp(n, seed=n)
    cfg = ops.PGDConfig(max_iters=1000, tol=1e-6)
    x, g, res = ops.solve_lcp(dev(A), dev(q), dev(np.full(n, 99.99)), cfg)
    assert res.converged and res.num_iters <= 1000
    np.testing.assert_allclose(host(x), x_star, atol=1e-5, rtol=0)
    with oracle.compensated_sums():
        xo, go, ro = oracle.solve_cqpp_dense(A, q, (1, 0.0, 0.0), np.full(n, 99.99), max_iters=1000, tol=1e-6)
    assert res.num_iters == ro["num_iters"]
    np.testing.assert_allclose(host(g), A @ host(x) + q, atol=1e-9)
    with pytest.raises(ValueError, match="dimension mismatch"):
        ops.solve_lcp(dev(A[:, :-1].copy()), dev(q), dev(np.zeros(n)), cfg)


def _sphere_problem(oracle, n, seed, phi=0.3, buffer=0.3):
    from mundy_amd import synth
    s = synth.spheres(n, volume_fraction=phi, seed=seed)
    c, r = s["center"], s["radius"]
    lo, hi, R = oracle.grow(oracle.compute_aabb_spheres(c, r), r, buffer)
    pairs = oracle.search(0, lo, hi, c, R)
    sep, nrm = oracle.contact_spheres(pairs, c, r)
    mt, _ = synth.dry_mobility(r)
    return dict(N=n, pairs=pairs, sep=sep, normal=nrm, ra=None, rb=None, mt=mt, mr=None)


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


def _rod_problem_arclength(oracle, n, seed, buffer=0.1):
    # same system; the GPU operator is built from (s, t, seg) (mhip_contact_op_create_rods) while the oracle keeps
    # the reference's (ra, rb) vectors
    return dict(_rod_problem(oracle, n, seed, buffer), rod=True)


def _gpu_op(ops, P, dt=5e-3):
    from gpu_util import dev
    opt = lambda a: None if a is None else dev(a)  # noqa: E731
    if P.get("rod"):
        return ops.ContactOperator(dev(P["pairs"]), dev(P["normal"]), dev(P["mt"]), dt, mob_rot=dev(P["mr"]),
                                   rod=(dev(P["s"]), dev(P["t"]), dev(P["seg"])))
    return ops.ContactOperator(dev(P["pairs"]), dev(P["normal"]), dev(P["mt"]), dt, ra=opt(P["ra"]), rb=opt(P["rb"]),
                               mob_rot=opt(P["mr"]))


@pytest.mark.parametrize("maker,n", [(_sphere_problem, 3000), (_rod_problem, 3000), (_rod_problem_arclength, 3000)])
def test_contact_operator_apply(ops, oracle, maker, n):
    # the GPU operator against the serial scatter/mobility/gather of NgpLcp.cpp:442-530.  Per-contact terms are the
    # same expressions; the per-body sums run as a fixed G-lane tree instead of serially, so the bar is rounding
    # level (1e-12 of the result scale) -- and bitwise reproducibility run to run, which the serial reference
    # (atomics on a parallel backend) does not have.
    from gpu_util import assert_bits_equal, dev, host
    P = maker(oracle, n, seed=3)
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 1, len(P["pairs"]))
    op = _gpu_op(ops, P)
    y = host(op.apply(dev(x)))
    yo = oracle.contact_op_apply(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3, x, P["N"])
    np.testing.assert_allclose(y, yo, rtol=1e-12, atol=1e-12 * np.abs(yo).max())
    assert_bits_equal(host(op.apply(dev(x))), y, "A x twice")
    # pairs in arbitrary (unsorted, shuffled) order give the same operator
    perm = rng.permutation(len(x))
    Q = dict(P, pairs=np.ascontiguousarray(P["pairs"][perm]), normal=np.ascontiguousarray(P["normal"][perm]))
    if P["ra"] is not None:
        Q.update({k: np.ascontiguousarray(P[k][perm]) for k in ("ra", "rb", "s", "t")})
    y2 = host(_gpu_op(ops, Q).apply(dev(x[perm])))
    np.testing.assert_allclose(y2, yo[perm], rtol=1e-12, atol=1e-12 * np.abs(yo).max())
    op.close()


@pytest.mark.parametrize("maker,n", [(_sphere_problem, 3000), (_rod_problem, 3000), (_rod_problem_arclength, 3000)])
def test_contact_operator_refresh_equals_rebuild(ops, oracle, maker, n):
    # a step that reuses the neighbour list keeps the operator and refreshes its geometry (mhip_contact_op_refresh*):
    # same pairs, perturbed normals / arms / segments.  The refreshed operator must act like one built from scratch on
    # the new geometry (same terms; the order inside a body's list is the one fixed at create, so rounding level) and
    # solve the LCP to the same point.
    from gpu_util import dev, host
    P = maker(oracle, n, seed=5)
    rng = np.random.default_rng(1)
    nrm2 = P["normal"] + 0.05 * rng.normal(size=P["normal"].shape)
    nrm2 /= np.linalg.norm(nrm2, axis=1, keepdims=True)
    Q = dict(P, normal=nrm2, sep=P["sep"] + 0.01 * rng.normal(size=P["sep"].shape))
    if P["ra"] is not None:
        Q["ra"] = P["ra"] + 0.02 * rng.normal(size=P["ra"].shape)
        Q["rb"] = P["rb"] + 0.02 * rng.normal(size=P["rb"].shape)
    if P.get("rod"):
        seg2 = P["seg"].copy()
        seg2[:, :6] += 0.02 * rng.normal(size=(len(seg2), 6))
        Q.update(seg=seg2, s=np.clip(P["s"] + 0.03 * rng.normal(size=P["s"].shape), 0, 1),
                 t=np.clip(P["t"] + 0.03 * rng.normal(size=P["t"].shape), 0, 1))
    op = _gpu_op(ops, P)
    x = dev(rng.uniform(0, 1, len(P["pairs"])))
    y_old = host(op.apply(x))
    opt = lambda a: None if a is None else dev(a)  # noqa: E731
    if P.get("rod"):
        op.refresh(dev(Q["normal"]), rod=(dev(Q["s"]), dev(Q["t"]), dev(Q["seg"])))
    else:
        op.refresh(dev(Q["normal"]), ra=opt(Q["ra"]), rb=opt(Q["rb"]))
    fresh = _gpu_op(ops, Q)
    y, yf = host(op.apply(x)), host(fresh.apply(x))
    assert np.abs(y - y_old).max() > 1e-6 * np.abs(y_old).max()       # the geometry did change
    np.testing.assert_allclose(y, yf, rtol=1e-12, atol=1e-12 * np.abs(yf).max())
    cfg = ops.PGDConfig(max_iters=20000, tol=1e-6)
    xa, ga, ra_ = ops.solve_lcp(op, dev(Q["sep"]), dev(np.zeros(len(Q["sep"]))), cfg)
    xb, gb, rb_ = ops.solve_lcp(fresh, dev(Q["sep"]), dev(np.zeros(len(Q["sep"]))), cfg)
    assert ra_.converged and rb_.converged
    np.testing.assert_allclose(host(ga), host(gb), atol=20 * 1e-6)
    op.close()
    fresh.close()


def test_rod_operator_body_velocity_matches_vector_form(ops, oracle):
    # the rod-compressed operator keeps (U, W x u) in its velocity rows; body_velocity() hands back (U, W), equal to
    # the vector-arm operator's to rounding (arm (s - 1/2) u vs (p0 + s u) - c: one ulp of the centre coordinate)
    from gpu_util import assert_bits_equal, dev, host
    P = _rod_problem(oracle, 3000, seed=5)
    x = dev(np.random.default_rng(1).uniform(0, 1, len(P["pairs"])))
    opv, opr = _gpu_op(ops, P), _gpu_op(ops, dict(P, rod=True))
    yv, yr = host(opv.apply(x)), host(opr.apply(x))
    vv, vr = host(opv.body_velocity()), host(opr.body_velocity())
    scale = np.abs(vv).max(axis=0)
    assert np.all(scale > 0)
    np.testing.assert_allclose(vr, vv, rtol=0, atol=1e-12 * scale.max())
    np.testing.assert_allclose(yr, yv, rtol=1e-12, atol=1e-12 * np.abs(yv).max())
    assert_bits_equal(host(opr.body_velocity()), vr, "body velocity twice")
    opv.close()
    opr.close()


@pytest.mark.parametrize("maker,n", [(_sphere_problem, 4000), (_rod_problem, 4000), (_rod_problem_arclength, 4000)])
def test_fused_bbpgd_matches_oracle_and_unfused(ops, oracle, maker, n):
    from gpu_util import dev, host
    P = maker(oracle, n, seed=11)
    C = len(P["pairs"])
    tol = 1e-6
    cfg = ops.PGDConfig(max_iters=10000, tol=tol)
    op = _gpu_op(ops, P)
    x, g, res = ops.solve_lcp(op, dev(P["sep"]), dev(np.zeros(C)), cfg)
    xu, gu, resu = ops.solve_lcp(op, dev(P["sep"]), dev(np.zeros(C)), cfg, fused=False)
    rod = (P["s"], P["t"], P["seg"]) if P.get("rod") else None   # the association the rod operator evaluates
    with oracle.compensated_sums():
        xo, go, ro = oracle.solve_cqpp_contact(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3,
                                               P["sep"], np.zeros(C), max_iters=10000, tol=tol, rod=rod)
    assert res.converged and resu.converged and ro["converged"]
    assert res.residual <= tol and resu.residual <= tol
    # same algorithm, same definition of every sum's rounding (double-double on both sides): the same trajectory --
    # SURVEY 8c's "identical iteration count ... else +/- few"
    print("iterations fused %d unfused %d oracle %d" % (res.num_iters, resu.num_iters, ro["num_iters"]))
    assert abs(res.num_iters - ro["num_iters"]) <= 2
    assert abs(resu.num_iters - ro["num_iters"]) <= 2
    x, g = host(x), host(g)
    if res.num_iters == ro["num_iters"]:
        np.testing.assert_allclose(x, xo, rtol=0, atol=1e-13 * max(1.0, np.abs(xo).max()))   # the same iterate
    # LCP conditions at the reference's acceptance level (10 tol): x >= 0, g >= -10 tol, x_i g_i small
    assert x.min() >= 0.0 and g.min() >= -10 * tol
    assert np.max(np.abs(np.minimum(x, g))) <= 10 * tol
    # g = A x + q is unique for the LCP (x need not be): compare gradients, and impulses where they are active
    np.testing.assert_allclose(g, go, atol=20 * tol)
    np.testing.assert_allclose(host(gu), go, atol=20 * tol)
    # g really is A x + q for the returned x
    yo = oracle.contact_op_apply(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3, x, P["N"])
    np.testing.assert_allclose(g, yo + P["sep"], atol=1e-9)
    op.close()


@pytest.mark.parametrize("maker,n", [(_sphere_problem, 30000), (_rod_problem, 12000), (_rod_problem_arclength, 20000)])
def test_iterates_do_not_depend_on_work_mapping_or_contact_order(ops, oracle, maker, n):
    # every sum that reaches an iterate is a double-double pair rounded once: the tile -> XCD mapping (which workgroup
    # writes which block partial), the lanes that share a body's list, and the order of the contact list itself leave
    # the iterates -- and so the BBPGD iteration count -- bit for bit where they were (629 ... 834
    # iterations on one input depending on the mapping, with plain sums)
    from gpu_util import assert_bits_equal, dev, host
    P = maker(oracle, n, seed=31)
    C = len(P["pairs"])
    assert C > 8 * 8 * 256            # enough 256-contact tiles for the permuted windows to exist
    cfg = ops.PGDConfig(max_iters=10000, tol=1e-6)
    ref = None
    for xcd_tile, lanes in ((32, 4), (0, 4), (2, 2), (8, 8), (5, 16)):
        op = _gpu_op(ops, P)
        op.set_work_mapping(xcd_tile, lanes)
        x, g, res = ops.solve_lcp(op, dev(P["sep"]), dev(np.zeros(C)), cfg)
        assert res.converged
        if ref is None:
            ref = (host(x), host(g), res.num_iters, res.residual)
        else:
            assert (res.num_iters, res.residual) == ref[2:], (xcd_tile, lanes, res.num_iters, ref[2])
            assert_bits_equal(host(x), ref[0], "x under mapping %s" % ((xcd_tile, lanes),))
            assert_bits_equal(host(g), ref[1], "g under mapping %s" % ((xcd_tile, lanes),))
        op.close()
    perm = np.random.default_rng(7).permutation(C)
    Q = dict(P, **{k: np.ascontiguousarray(P[k][perm]) for k in ("pairs", "normal", "sep")})
    for k in ("ra", "rb", "s", "t"):
        if P.get(k) is not None:
            Q[k] = np.ascontiguousarray(P[k][perm])
    op = _gpu_op(ops, Q)
    x, g, res = ops.solve_lcp(op, dev(Q["sep"]), dev(np.zeros(C)), cfg)
    assert res.num_iters == ref[2]
    assert_bits_equal(host(x), ref[0][perm], "x with the contacts shuffled")
    assert_bits_equal(host(g), ref[1][perm], "g with the contacts shuffled")
    op.close()
    with pytest.raises(ValueError):
        _gpu_op(ops, P).set_work_mapping(0, 3)


@pytest.mark.parametrize("maker,n", [(_sphere_problem, 4000), (_rod_problem, 3000), (_rod_problem_arclength, 3000)])
def test_scrap_variant_matches_oracle(ops, oracle, maker, n):
    # resolve_collisions of scrap/lcp_spheres/NgpLcp.cpp:558-759 (BB1/BB2 alternation, Dai-Fletcher residual)
    from gpu_util import dev, host
    P = maker(oracle, n, seed=21)
    C = len(P["pairs"])
    tol = 1e-5
    op = _gpu_op(ops, P)
    lam, g, res = ops.resolve_collisions(op, dev(P["sep"]), dev(np.zeros(C)), 5e-3, max_allowable_overlap=tol)
    rod = (P["s"], P["t"], P["seg"]) if P.get("rod") else None
    with oracle.compensated_sums():
        lo, go, ro = oracle.scrap_resolve_collisions(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3,
                                                     P["sep"], np.zeros(C), max_allowable_overlap=tol, rod=rod)
    assert res.max_abs_projected_sep < tol and ro["max_abs_projected_sep"] < tol
    assert abs(res.ite_count - ro["ite_count"]) <= 2
    lam, g = host(lam), host(g)
    assert lam.min() >= 0 and g.min() > -tol
    np.testing.assert_allclose(g, go, atol=20 * tol)
    assert res.max_displacement == pytest.approx(ro["max_speed"] * 5e-3, rel=1e-3)
    # and it solves the same LCP as the convex.hpp solver
    x2, g2, r2 = ops.solve_lcp(op, dev(P["sep"]), dev(np.zeros(C)), ops.PGDConfig(max_iters=10000, tol=tol))
    np.testing.assert_allclose(g, host(g2), atol=40 * tol)
    # nonzero initial guess exercises the first-step quirk (:639) identically on both sides
    lam0 = np.abs(np.sin(np.arange(C))) * 0.01
    lam_b, g_b, res_b = ops.resolve_collisions(op, dev(P["sep"]), dev(lam0), 5e-3, max_allowable_overlap=tol)
    with oracle.compensated_sums():
        lo_b, go_b, ro_b = oracle.scrap_resolve_collisions(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"],
                                                           5e-3, P["sep"], lam0, max_allowable_overlap=tol, rod=rod)
    assert abs(res_b.ite_count - ro_b["ite_count"]) <= 2
    np.testing.assert_allclose(host(g_b), go_b, atol=20 * tol)
    op.close()


def test_in_kernel_small_problems_bit_exact(ops, oracle):
    # MundyMathBackend.  Same scalar arithmetic in the same order -> bit-exact against the oracle, including
    # iteration counts; the reference's own problems (UnitTestConvex.cpp:608-615) are in the batch.
    from gpu_util import assert_bits_equal, dev, host
    from test_oracle_convex_kat import random_lcp
    rng = np.random.default_rng(3)
    for n in (1, 3, 7, 16):
        batch = 3000
        As, qs, xs = [], [], []
        for b in range(batch):
            A, q, x_star = random_lcp(n, seed=1000 * n + b) if n > 1 else (np.array([[2.0]]), np.array([-1.0]), np.array([0.5]))
            As.append(A); qs.append(q); xs.append(x_star)
        A, q, x_star = np.array(As), np.array(qs), np.array(xs)
        x0 = np.full((batch, n), 99.99)
        cfg = ops.PGDConfig(max_iters=1000, tol=1e-6)
        for space in ((1, 0.0, 0.0), (3, 0.05, 0.6)):
            x, g, it, res, conv = ops.solve_small_cqpp_batch(dev(A), dev(q), space, dev(x0), cfg)
            xo, go, ito, reso, convo = oracle.solve_small_cqpp_batch(A, q, space, x0, max_iters=1000, tol=1e-6)
            assert_bits_equal(host(x), xo, "small x n=%d" % n)
            assert_bits_equal(host(g), go, "small g n=%d" % n)
            np.testing.assert_array_equal(host(it), ito.astype(np.int32))
            assert_bits_equal(host(res), reso, "small residual")
            np.testing.assert_array_equal(host(conv), convo)
            if space[0] == 1:
                assert convo.all()
                np.testing.assert_allclose(host(x), x_star, atol=1e-5, rtol=0)
    with pytest.raises(ValueError, match="size"):
        ops.solve_small_cqpp_batch(dev(np.zeros((1, 17, 17))), dev(np.zeros((1, 17))), (1, 0.0, 0.0), dev(np.zeros((1, 17))))


def test_state_vector_postconditions(ops, oracle):
    # what the caller-owned state holds on return (convex.hpp:614-666): converged -> x final / x_tmp previous iterate;
    # converged at init -> grad == grad_tmp; max_iters hit -> x == x_tmp
    import torch
    from gpu_util import dev, host
    P = _sphere_problem(oracle, 1500, seed=2)
    C = len(P["pairs"])
    op = _gpu_op(ops, P)
    q = dev(P["sep"])
    for max_iters in (7, 8, 10000):  # odd / even parity of the ping-pong, and convergence
        st = tuple(dev(np.zeros(C)) for _ in range(4))
        x, g, res = ops.solve_lcp(op, q, None, ops.PGDConfig(max_iters=max_iters, tol=1e-6), state=st)
        xo, go, ro = oracle.solve_cqpp_contact(P["pairs"], P["normal"], None, None, P["mt"], None, 5e-3, P["sep"],
                                               np.zeros(C), max_iters=max_iters, tol=1e-6)
        assert res.converged == ro["converged"]
        if not res.converged:
            assert res.num_iters == max_iters == ro["num_iters"]
            assert torch.equal(st[0], st[2]) and torch.equal(st[1], st[3])
            np.testing.assert_allclose(host(st[0]), xo, rtol=1e-6, atol=1e-9)
        else:
            assert not torch.equal(st[0], st[2])
            # x_tmp is the previous iterate: one more projected step from it reproduces x
            y = op.apply(st[0])
            np.testing.assert_allclose(host(y) + P["sep"], host(st[1]), atol=1e-9)
    # already converged initial guess: zero iterations, grad = grad_tmp
    st = (x.clone(), dev(np.zeros(C)), dev(np.zeros(C)), dev(np.zeros(C)))
    _, _, res0 = ops.solve_lcp(op, q, None, ops.PGDConfig(max_iters=100, tol=1e-5), state=st)
    assert res0.converged and res0.num_iters == 0
    assert torch.equal(st[1], st[3]) and torch.equal(st[0], st[2])
    op.close()


@pytest.mark.parametrize("maker,n", [(_sphere_problem, 6000), (_rod_problem, 4000)])
def test_incidence_build_paths_give_the_same_solve(ops, oracle, maker, n):
    # round 4: a pair list of the broad phase's form (rows sorted by the lower body, i < j) takes the fast incidence build
    # -- only the transpose half counted / filled with atomics and sorted --, any other list the general one.  The same
    # contacts through both: (a) as they are (fast path), (b) with every body index multiplied by 3 -- bodies without any
    # contact between the rows, still the fast path --, (c) multiplied by 5000 -- runs of empty rows beyond the fast
    # path's limit: the general path --, (d) rows in reversed order -- not sorted: the general path.  Every sum is
    # rounded once, so all four give the same iteration count and the same multipliers, bit for bit.
    from gpu_util import assert_bits_equal, dev, host
    P = maker(oracle, n, seed=17)
    C = len(P["pairs"])
    cfg = ops.PGDConfig(max_iters=10000, tol=1e-6)

    def spread(k):
        Q = dict(P)
        Q["N"] = P["N"] * k
        Q["pairs"] = np.ascontiguousarray(P["pairs"] * k)
        for key in ("mt", "mr"):
            if P.get(key) is not None:
                a = np.ones(Q["N"])
                a[::k] = P[key]
                Q[key] = a
        return Q

    def solve(Q, order=None):
        if order is not None:
            Q = dict(Q, **{key: np.ascontiguousarray(Q[key][order]) for key in ("pairs", "normal", "sep", "ra", "rb")
                           if Q.get(key) is not None})
        op = _gpu_op(ops, Q)
        x, g, res = ops.solve_lcp(op, dev(Q["sep"]), dev(np.zeros(C)), cfg)
        out = (host(x), host(g), res.num_iters, bool(res.converged))
        op.close()
        return out

    ref = solve(P)
    assert ref[3] and ref[2] > 20
    for name, got in (("bodies x 3", solve(spread(3))), ("bodies x 5000", solve(spread(5000)))):
        assert got[2:] == ref[2:], (name, got[2], ref[2])
        assert_bits_equal(got[0], ref[0], "x, " + name)
        assert_bits_equal(got[1], ref[1], "g, " + name)
    rev = np.arange(C)[::-1].copy()
    got = solve(P, rev)
    assert got[2:] == ref[2:], ("reversed rows", got[2], ref[2])
    assert_bits_equal(got[0][rev], ref[0], "x, reversed rows")
    assert_bits_equal(got[1][rev], ref[1], "g, reversed rows")


def test_empty_and_invalid(ops):
    import torch
    from gpu_util import dev
    z = lambda *s, dt=torch.float64: torch.zeros(s, dtype=dt, device="cuda")  # noqa: E731
    op = ops.ContactOperator(z(0, 2, dt=torch.int32), z(0, 3), z(5), 1e-3)
    x, g, res = ops.solve_lcp(op, z(0), z(0))
    assert res.converged and res.num_iters == 0
    op.close()
    bad = dev(np.array([[0, 9]], dtype=np.int32))
    with pytest.raises(ValueError, match="outside"):
        ops.ContactOperator(bad, z(1, 3), z(5), 1e-3)
    with pytest.raises(ValueError, match="GPU"):
        ops.axpby(1.0, torch.zeros(3, dtype=torch.float64), 1.0, torch.zeros(3, dtype=torch.float64))


def test_full_size_lcp_properties_1M_rods(ops):
    # BASELINE.json configs[2] size (10^6 spherocylinders, phi = 0.4): size-independent properties of the solve
    import torch
    from gpu_util import dev
    from mundy_amd import pipeline, synth
    b = synth.spherocylinders(1_000_000)
    tol = 1e-5
    st = pipeline.ContactStepper("spherocylinder", dev(b["center"]), dev(b["radius"]), dev(b["quat"]),
                                 dev(b["length"]), search_buffer=0.1, cfg=ops.PGDConfig(max_iters=3000, tol=tol))
    s = st.step(integrate=False)
    assert s.converged and s.num_contacts > 5_000_000
    lam, sep = st.lam, st.contacts["sep"]
    g = st.op.apply(lam) + sep                              # linearised separation after the step
    assert float(lam.min()) >= 0.0
    assert float(g.min()) >= -10 * tol                      # no residual overlap beyond the tolerance
    assert float(torch.minimum(lam, g).abs().max()) <= 10 * tol   # complementarity
    assert int((lam > 0).sum()) > 1_000_000
    # linearity of the operator: A(2x) == 2 A x bit for bit (power-of-two scaling), A(x + y) ~ A x + A y
    y = st.op.apply(lam)
    assert torch.equal(st.op.apply(2.0 * lam), 2.0 * y)
    z = torch.rand_like(lam)
    torch.testing.assert_close(st.op.apply(lam + z), y + st.op.apply(z), rtol=1e-9, atol=1e-9)
    # symmetry: <z, A x> == <x, A z>
    a, bb = float((z * y).sum()), float((lam * st.op.apply(z)).sum())
    assert abs(a - bb) <= 1e-9 * max(1.0, abs(a))


@pytest.mark.parametrize("rigid", [False, True])
def test_high_degree_bodies_beyond_the_activity_mask(ops, oracle, rigid):
    # a few big spheres each touched by ~150 small ones: incidence lists longer than the 64 slots the per-body activity
    # masks cover (the tail is always walked), next to ordinary low-degree bodies
    from gpu_util import dev, host
    from mundy_amd import synth
    rng = np.random.default_rng(8)
    big = np.array([[0.0, 0.0, 0.0], [14.0, 0.0, 0.0], [0.0, 14.0, 3.0]])
    centers, radii = [big], [np.full(3, 5.0)]
    for b in big:
        d = rng.normal(size=(150, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        centers.append(b + d * (5.0 + 0.3 - rng.uniform(0.0, 0.08, (150, 1))))   # small spheres pressed into the big one
        radii.append(np.full(150, 0.3))
    c, r = np.concatenate(centers), np.concatenate(radii)
    n = len(c)
    lo, hi, R = oracle.grow(oracle.compute_aabb_spheres(c, r), r, 0.2)
    pairs = oracle.search(0, lo, hi, c, R)
    deg = np.bincount(pairs.ravel(), minlength=n)
    assert deg.max() > 100 and np.median(deg) < 20
    sep, nrm = oracle.contact_spheres(pairs, c, r)
    mt, mr = synth.dry_mobility(r)
    ra = rb = None
    if rigid:   # surface lever arms: torques vanish for central forces, the 6-DOF kernels still run
        ra, rb = r[pairs[:, 0]][:, None] * nrm, -r[pairs[:, 1]][:, None] * nrm
    C = len(pairs)
    tol = 1e-7
    opt = lambda a: None if a is None else dev(a)  # noqa: E731
    op = ops.ContactOperator(dev(pairs), dev(nrm), dev(mt), 5e-3, ra=opt(ra), rb=opt(rb), mob_rot=dev(mr) if rigid else None)
    x, g, res = ops.solve_lcp(op, dev(sep), dev(np.zeros(C)), ops.PGDConfig(max_iters=20000, tol=tol))
    xo, go, ro = oracle.solve_cqpp_contact(pairs, nrm, ra, rb, mt, mr if rigid else None, 5e-3, sep, np.zeros(C),
                                           max_iters=20000, tol=tol)
    assert res.converged and ro["converged"]
    np.testing.assert_allclose(host(g), go, atol=20 * tol)
    x = host(x)
    assert x.min() >= 0 and np.abs(np.minimum(x, host(g))).max() <= 10 * tol
    # g is A x + q for the returned x: checks the body sweep of the final iterate against the plain apply
    np.testing.assert_allclose(host(op.apply(dev(x))) + sep, host(g), atol=1e-10)
    op.close()


def test_hub_bodies_with_thousands_of_contacts(ops, oracle):
    # round 4: the fast incidence build sorts a body's targets in one thread; a list of thousands (here two hubs touched by
    # 2 500 small spheres each, one hub with the LOWEST index -- all sources --, one with the HIGHEST -- all targets) must
    # take the general path with its workgroup radix sort instead.  Same checks as for the ~150-contact bodies above.
    from gpu_util import dev, host
    from mundy_amd import synth
    rng = np.random.default_rng(9)
    hubs = np.array([[0.0, 0.0, 0.0], [60.0, 0.0, 0.0]])
    small = []
    for b in hubs:
        d = rng.normal(size=(2500, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        small.append(b + d * (20.0 + 0.3 - rng.uniform(0.0, 0.05, (2500, 1))))
    c = np.concatenate([hubs[:1], small[0], small[1], hubs[1:]])
    r = np.concatenate([[20.0], np.full(5000, 0.3), [20.0]])
    n = len(c)
    lo, hi, R = oracle.grow(oracle.compute_aabb_spheres(c, r), r, 0.1)
    pairs = oracle.search(0, lo, hi, c, R)
    deg = np.bincount(pairs.ravel(), minlength=n)
    assert deg[0] > 2000 and deg[-1] > 2000 and np.median(deg) < 20
    sep, nrm = oracle.contact_spheres(pairs, c, r)
    mt, _ = synth.dry_mobility(r)
    C = len(pairs)
    tol = 1e-7
    op = ops.ContactOperator(dev(pairs), dev(nrm), dev(mt), 5e-3, priority=dev(sep))
    x, g, res = ops.solve_lcp(op, dev(sep), dev(np.zeros(C)), ops.PGDConfig(max_iters=20000, tol=tol))
    with oracle.compensated_sums():
        xo, go, ro = oracle.solve_cqpp_contact(pairs, nrm, None, None, mt, None, 5e-3, sep, np.zeros(C), max_iters=20000,
                                               tol=tol, threads=False)
    assert res.converged and ro["converged"] and abs(res.num_iters - ro["num_iters"]) <= 2
    np.testing.assert_allclose(host(g), go, atol=20 * tol)
    x = host(x)
    assert x.min() >= 0 and np.abs(np.minimum(x, host(g))).max() <= 10 * tol
    np.testing.assert_allclose(host(op.apply(dev(x))) + sep, host(g), atol=1e-10)
    op.close()


def test_staged_api_whole_range_equals_fused_solve(ops, oracle):
    # the staged entry points driven by hand on one rank, with the whole-range mhip_bbpgd_stage_constraint wrapper:
    # same kernels in the same order as the fused driver -> identical iterates
    import ctypes as C
    import torch
    from gpu_util import dev, host
    from mundy_amd import capi
    lib = capi.load()
    P = _rod_problem_arclength(oracle, 3000, seed=17)
    nc = len(P["pairs"])
    cfg = ops.PGDConfig(max_iters=5000, tol=1e-6)
    op = _gpu_op(ops, P)
    x_ref, g_ref, r_ref = ops.solve_lcp(op, dev(P["sep"]), dev(np.zeros(nc)), cfg)
    sep = dev(P["sep"])
    x, g, xt, gt = (torch.zeros(nc, dtype=torch.float64, device="cuda") for _ in range(4))
    local3 = torch.empty(5, dtype=torch.float64, device="cuda")  # MHIP_BBPGD_REDUCTION_WIDTH
    sp = capi.Space(ops.SPACE_LOWER_BOUND, 0.0, 0.0)
    pc = capi.PgdConfig(cfg.max_iters, cfg.tol, cfg.residual_kind)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    capi.check(lib.mhip_bbpgd_stage_begin(op._h, p(sep), C.byref(sp), C.byref(pc), p(x), p(g), p(xt), p(gt), None))
    res, done = capi.SolveResult(), C.c_int(0)
    for it in range(cfg.max_iters + 1):
        init = 1 if it == 0 else 0
        capi.check(lib.mhip_bbpgd_stage_body(op._h, init, None))
        capi.check(lib.mhip_bbpgd_stage_constraint(op._h, init, p(local3), None))
        capi.check(lib.mhip_bbpgd_stage_finalize(op._h, init, p(local3), 1, None))   # one rank: its own record
        capi.check(lib.mhip_bbpgd_stage_poll(op._h, C.byref(res), C.byref(done), None))
        if done.value:
            break
    capi.check(lib.mhip_bbpgd_stage_end(op._h, C.byref(res), None))
    assert res.converged and res.num_iters == r_ref.num_iters and res.residual == r_ref.residual
    assert np.array_equal(host(x), host(x_ref)) and np.array_equal(host(g), host(g_ref))
    op.close()


def test_fill_and_deep_copy(ops):
    import ctypes as C
    import torch
    from mundy_amd import capi
    lib = capi.load()
    for n in (0, 1, 1000, 1_000_001):
        t = torch.full((n + 2,), -1.0, dtype=torch.float64, device="cuda")
        capi.check(lib.mhip_fill(n, C.c_void_p(t.data_ptr() + 8), 2.5, None))
        assert bool((t[1:n + 1] == 2.5).all()) and float(t[0]) == -1.0 and float(t[-1]) == -1.0
        u = torch.zeros(n + 2, dtype=torch.float64, device="cuda")
        capi.check(lib.mhip_deep_copy(n, C.c_void_p(u.data_ptr() + 8), C.c_void_p(t.data_ptr() + 8), None))
        assert torch.equal(u[1:n + 1], t[1:n + 1]) and float(u[0]) == 0.0 and float(u[-1]) == 0.0


@pytest.mark.parametrize("drift_source", [1, 2])
@pytest.mark.parametrize("maker,n", [(_sphere_problem, 60000), (_rod_problem_arclength, 12000), (_rod_problem, 12000)])
def test_cold_tier_leaves_every_bit_where_it_was(ops, oracle, maker, n, drift_source):
    # the fused solve keeps inactive contacts in a cold tier from the first convergence poll on (renumbered hot-first,
    # the tail swept only through drift bounds): x, g and the previous iterate, the iteration count and the body
    # velocities equal the untiered solve bit for bit -- for a run to convergence, for iteration caps (61, 62) that end
    # the solve inside the tiers at either parity, and when the solve is made to leave the tiers mid-way (what it does
    # before a BB step outside [0, finite]); afterwards the operator is back in the caller's numbering.  Both sources
    # of the drift bound (round 4: the difference of a body's two rows, the default up to 1.75e6 bodies; rounds 2-3: the
    # change of the force kept in registers, what larger systems take) -- it decides who sleeps, never a bit of an iterate
    import torch
    from gpu_util import dev
    P = maker(oracle, n, seed=41)
    C = len(P["pairs"])
    assert 65536 <= C < 1_500_000, C       # (below the size from which the tier is on by default: mode 3 = any size)
    q = dev(P["sep"])
    results = {}
    for mode in (0, 3, 2):
        op = _gpu_op(ops, P)
        op.set_tiering(mode)
        op.set_drift_source(drift_source)
        out = []
        for max_iters in (10000, 61, 62):
            st = tuple(dev(np.zeros(C)) for _ in range(4))
            x, g, res = ops.solve_lcp(op, q, None, ops.PGDConfig(max_iters=max_iters, tol=1e-6), state=st)
            stats = op.tier_stats()
            out.append((st, res, op.body_velocity().clone(), stats))
            if mode == 0:
                assert stats["tiered_iterations"] == 0
            elif mode == 3:
                # (drift bookkeeping from the poll at 8 iterations, tiers from the poll at 24 or 56)
                assert stats["tiered_iterations"] >= min(res.num_iters, max_iters) - 56 > 0, stats
                assert stats["renumberings"] >= 1, stats
                if max_iters == 10000:
                    assert res.converged and stats["mean_hot_fraction"] < 0.85, stats
            else:
                assert stats["tiered_iterations"] == 1, stats
            # back in the caller's numbering: the operator applied to the solution reproduces the gradient
            y = op.apply(st[0])
            np.testing.assert_allclose((y + q).cpu().numpy(), st[1].cpu().numpy(), atol=1e-9)
        results[mode] = out
        op.close()
    op = _gpu_op(ops, P)                     # default mode at this size: every contact swept
    ops.solve_lcp(op, q, dev(np.zeros(C)), ops.PGDConfig(max_iters=100, tol=1e-6))
    assert op.tier_stats()["tiered_iterations"] == 0
    op.close()
    for mode in (3, 2):
        for (st0, r0, v0, _), (st1, r1, v1, _) in zip(results[0], results[mode]):
            assert (r0.num_iters, r0.converged, r0.residual) == (r1.num_iters, r1.converged, r1.residual)
            for a, b in zip(st0, st1):
                assert torch.equal(a, b)
            assert torch.equal(v0, v1)
