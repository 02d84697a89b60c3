"""BUILD EXTENSION, parity unpinned: Coulomb friction as a cone complementarity problem.  The reference has no
frictional solver, so nothing here is checked against MuNDy; these tests pin the CPU statement of the
extension on mathematical properties, and on the reference-pinned frictionless solve for mu = 0."""
import numpy as np
import pytest


def _rod_system(oracle, n, seed, buffer=0.1):
    from mundy_amd import synth
    b = synth.spherocylinders(n, seed=seed, volume_fraction=0.3)
    c = b["center"]
    aabb = oracle.compute_aabb_spherocylinders(c, b["quat"], b["radius"], b["length"])
    brad = oracle.bounding_radius_spherocylinders(b["radius"], b["length"])
    lo, hi, R = oracle.grow(aabb, brad, buffer)
    pairs = oracle.search(1, lo, hi, c, R)
    seg = oracle.spherocylinder_segments(c, b["quat"], b["radius"], b["length"])
    out = oracle.contact_spherocylinders(pairs, seg, c)
    mt, mr = synth.dry_mobility(b["radius"], bounding_radius=brad)
    n_ = out["normal"]
    ras = out["ra"] + b["radius"][pairs[:, 0]][:, None] * n_      # arms to the contact points on the surfaces
    rbs = out["rb"] - b["radius"][pairs[:, 1]][:, None] * n_
    return dict(N=n, pairs=pairs, sep=out["sep"], normal=n_, ra=out["ra"], rb=out["rb"], ras=ras, rbs=rbs, mt=mt, mr=mr)


def cone_checks(p, g, normal, mu, tol):
    """p in K, g in K* = {mu |g_t| <= g.n}, p . g ~ 0 -- the cone complementarity conditions, at the solver's tolerance"""
    pn = (p * normal).sum(1)
    pt = np.linalg.norm(p - pn[:, None] * normal, axis=1)
    gn = (g * normal).sum(1)
    gt = np.linalg.norm(g - gn[:, None] * normal, axis=1)
    assert np.all(pt <= mu * pn + 1e-12 * (1 + np.abs(pn))), "impulse outside the friction cone"
    assert np.all(mu * gt <= gn + 20 * tol), "gradient outside the dual cone"
    assert np.max(np.abs((p * g).sum(1))) <= 20 * tol * max(1.0, np.abs(p).max()), "complementarity"


def test_cone_projection_properties(oracle):
    rng = np.random.default_rng(0)
    n = 20000
    v = rng.normal(size=(n, 3)) * rng.uniform(0.1, 10, (n, 1))
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    for mu in (0.0, 0.3, 1.0, 5.0):
        w = oracle.project_cone(v, nrm, mu)
        a = (w * nrm).sum(1)
        t = np.linalg.norm(w - a[:, None] * nrm, axis=1)
        assert np.all(a >= -1e-14) and np.all(t <= mu * a + 1e-12 * (1 + a))                   # lands in K
        np.testing.assert_allclose(oracle.project_cone(w, nrm, mu), w, atol=1e-12)             # idempotent
        r = v - w                                                                              # Moreau: r in the polar cone, r . w = 0
        np.testing.assert_allclose((r * w).sum(1), 0.0, atol=1e-10)
        ra = (r * nrm).sum(1)
        rt = np.linalg.norm(r - ra[:, None] * nrm, axis=1)
        assert np.all(mu * rt <= -ra + 1e-10 * (1 + np.abs(ra)))
        inside = np.linalg.norm(v - (v * nrm).sum(1)[:, None] * nrm, axis=1) <= mu * (v * nrm).sum(1)
        np.testing.assert_array_equal(w[inside], v[inside])                                    # points of K are fixed
    w0 = oracle.project_cone(v, nrm, 0.0)                                                      # mu = 0: max(v.n, 0) n
    np.testing.assert_allclose(w0, np.maximum((v * nrm).sum(1), 0.0)[:, None] * nrm, atol=1e-14)


def test_mu_zero_is_the_frictionless_lcp(oracle):
    P = _rod_system(oracle, 1500, seed=4)
    C = len(P["pairs"])
    tol = 1e-6
    x, g, r = oracle.solve_cqpp_contact(P["pairs"], P["normal"], P["ra"], P["rb"], P["mt"], P["mr"], 5e-3, P["sep"],
                                        np.zeros(C), max_iters=20000, tol=tol)
    # surface arms differ from the centreline arms by a multiple of n: no effect on a frictionless solve
    p, gv, rf = oracle.solve_friction_contact(P["pairs"], P["normal"], P["ras"], P["rbs"], P["mt"], P["mr"], 5e-3,
                                              P["sep"], 0.0, max_iters=20000, tol=tol)
    assert r["converged"] and rf["converged"]
    lam = (p * P["normal"]).sum(1)
    np.testing.assert_allclose(p, lam[:, None] * P["normal"], atol=1e-12)
    np.testing.assert_allclose((gv * P["normal"]).sum(1), g, atol=20 * tol)
    assert abs(rf["num_iters"] - r["num_iters"]) <= max(5, 0.25 * r["num_iters"])


@pytest.mark.parametrize("mu", [0.3, 1.0])
def test_friction_solution_satisfies_the_cone_complementarity_conditions(oracle, mu):
    P = _rod_system(oracle, 1200, seed=9)
    tol = 1e-6
    p, g, r = oracle.solve_friction_contact(P["pairs"], P["normal"], P["ras"], P["rbs"], P["mt"], P["mr"], 5e-3,
                                            P["sep"], mu, max_iters=50000, tol=tol)
    assert r["converged"]
    cone_checks(p, g, P["normal"], mu, tol)
    pn = (p * P["normal"]).sum(1)
    pt = np.linalg.norm(p - pn[:, None] * P["normal"], axis=1)
    assert (pt > 1e-6).sum() > 10          # friction is really engaged somewhere
    assert pn.max() > 0


@pytest.mark.parametrize("mu", [0.0, 0.3, 1.0])
def test_apgd_reaches_the_same_solution_in_fewer_sweeps(oracle, mu):
    # BUILD EXTENSION: APGD (Mazhar, Heyn, Negrut, Tasora 2015) on the same cone complementarity problem and with the
    # same stopping rule as the BBPGD form: the cone conditions hold, the gradient (the unique part of the solution)
    # agrees with BBPGD's, and with friction it gets there in fewer operator applications
    P = _rod_system(oracle, 2500, seed=13)
    tol = 1e-6
    args = (P["pairs"], P["normal"], P["ras"], P["rbs"], P["mt"], P["mr"], 5e-3, P["sep"], mu)
    pb, gb, rb = oracle.solve_friction_contact(*args, max_iters=50000, tol=tol)
    pa, ga, ra = oracle.solve_friction_contact(*args, max_iters=50000, tol=tol, method="apgd")
    assert rb["converged"] and ra["converged"] and ra["residual"] <= tol
    cone_checks(pa, ga, P["normal"], mu, tol)
    np.testing.assert_allclose(ga, gb, atol=40 * tol)
    print("mu=%g: sweeps apgd %d, bbpgd %d" % (mu, ra["num_iters"], rb["num_iters"]))
    if mu > 0:
        assert ra["num_iters"] < rb["num_iters"]
    # an iteration cap ends it unconverged with the last accepted iterate (still in the cone)
    pc, gc, rc = oracle.solve_friction_contact(*args, max_iters=7, tol=tol, method="apgd")
    assert not rc["converged"] and rc["num_iters"] == 7
    pn = (pc * P["normal"]).sum(1)
    pt = np.linalg.norm(pc - pn[:, None] * P["normal"], axis=1)
    assert np.all(pt <= mu * pn + 1e-12 * (1 + np.abs(pn)))
