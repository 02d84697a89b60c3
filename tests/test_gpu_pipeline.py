"""The whole step through the host-side pipeline (mundy_amd/pipeline.py) against the same step on the CPU oracle:
BASELINE.json configs[1] (100k spheres, frictionless LCP, one GPU, fp64) and a short multi-step rod trajectory."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import torch
    assert torch.cuda.is_available()
    from mundy_amd import ops, pipeline, synth
    return ops, pipeline, synth


def test_config2_100k_spheres_lcp(mods, oracle):
    ops, pipeline, synth = mods
    from gpu_util import assert_bits_equal, dev, host
    s = synth.spheres(100_000)                       # r = 1, phi = 0.40
    c, r = s["center"], s["radius"]
    tol = 1e-5
    st = pipeline.ContactStepper("sphere", dev(c), dev(r), search_buffer=0.25, search_kind=ops.SEARCH_SPHERES,
                                 cfg=ops.PGDConfig(max_iters=10000, tol=tol))
    res = st.step(integrate=False)
    lo, hi, R = oracle.grow(oracle.compute_aabb_spheres(c, r), r, 0.25)
    pairs = oracle.search(oracle.SEARCH_SPHERES, lo, hi, c, R)
    np.testing.assert_array_equal(host(st.links.pairs), pairs)          # bit-exact neighbour indices / counts
    osep, onrm = oracle.contact_spheres(pairs, c, r)
    assert_bits_equal(host(st.contacts["sep"]), osep, "sep")
    assert_bits_equal(host(st.contacts["normal"]), onrm, "normal")
    mt, _ = synth.dry_mobility(r)
    with oracle.compensated_sums():   # the bit-parity build (no FMA contraction) with the device's definition of the sums
        xo, go, ro = oracle.solve_cqpp_contact(pairs, onrm, None, None, mt, None, 5e-3, osep, np.zeros(len(pairs)),
                                               max_iters=10000, tol=tol, threads=False, fast=False)
    # (the serial oracle: the OpenMP one sums forces with atomics and stops somewhere else in the tolerance ball every
    # run -- 363 to 472 iterations and up to 15 tol away from the serial gradient on this very problem)
    assert res.converged and ro["converged"]
    # order-independent sums on both sides: the same trajectory (identical iteration count, else +/- few)
    assert abs(res.num_iters - ro["num_iters"]) <= 2
    g = host(st.op.apply(st.lam) + st.contacts["sep"])
    print("config 1: iterations gpu %d oracle %d, max |g - g_oracle| = %.3g" % (res.num_iters, ro["num_iters"], np.abs(g - go).max()))
    np.testing.assert_allclose(g, go, atol=20 * tol)                      # fp64 tolerance on the constraint gradient
    lam = host(st.lam)
    assert lam.min() >= 0 and np.abs(np.minimum(lam, g)).max() <= 10 * tol
    # impulses per body (D lam) are unique even where individual multipliers are not
    F = np.zeros((len(r), 3)); Fo = np.zeros((len(r), 3))
    for arr, x in ((F, lam), (Fo, xo)):
        np.add.at(arr, pairs[:, 0], -x[:, None] * onrm)
        np.add.at(arr, pairs[:, 1], x[:, None] * onrm)
    assert np.abs(F - Fo).max() <= 1e-3 * max(1.0, np.abs(Fo).max())


def test_three_step_rod_trajectory(mods, oracle):
    # positions after three full steps (neighbour list -> contacts -> LCP -> Euler with quaternion update) track the
    # oracle's trajectory; the LCP is solved tightly so the comparison is about the path, not solver slack
    ops, pipeline, synth = mods
    from gpu_util import dev, host
    b = synth.spherocylinders(4000, volume_fraction=0.25)
    tol, dt, buf = 1e-6, 5e-3, 0.3   # 1e-6: the projected-diff residual is quantised at ~ulp(x)/1e-6 ~ 1e-8
    st = pipeline.ContactStepper("spherocylinder", dev(b["center"]), dev(b["radius"]), dev(b["quat"]),
                                 dev(b["length"]), dt=dt, search_buffer=buf, cfg=ops.PGDConfig(max_iters=20000, tol=tol))
    c, q = b["center"].copy(), b["quat"].copy()
    r, L = b["radius"], b["length"]
    brad = oracle.bounding_radius_spherocylinders(r, L)
    mt, mr = synth.dry_mobility(r, bounding_radius=brad)
    for step in range(3):
        s = st.step(integrate=True, force_rebuild=True)
        aabb = oracle.compute_aabb_spherocylinders(c, q, r, L)
        lo, hi, R = oracle.grow(aabb, brad, buf)
        pairs = oracle.search(oracle.SEARCH_AABB, lo, hi, c, R)
        seg = oracle.spherocylinder_segments(c, q, r, L)
        con = oracle.contact_spherocylinders(pairs, seg, c)
        x, g, ro = oracle.solve_cqpp_contact(pairs, con["normal"], con["ra"], con["rb"], mt, mr, dt, con["sep"],
                                             np.zeros(len(pairs)), max_iters=20000, tol=tol, threads=False, fast=False)
        assert s.converged and ro["converged"]
        # body velocities from the multipliers, then the reference's Euler + rotate_quaternion update
        F = np.zeros((len(r), 3)); T = np.zeros((len(r), 3))
        f = x[:, None] * con["normal"]
        np.add.at(F, pairs[:, 0], -f); np.add.at(F, pairs[:, 1], f)
        np.add.at(T, pairs[:, 0], -np.cross(con["ra"], f)); np.add.at(T, pairs[:, 1], np.cross(con["rb"], f))
        U, W = mt[:, None] * F, mr[:, None] * T
        c = c + dt * U
        w = np.linalg.norm(W, axis=1)
        mv = w >= 1e-15
        sw, cw = np.sin(0.5 * w * dt), np.cos(0.5 * w * dt)
        winv = np.where(mv, 1.0 / np.where(mv, w, 1.0), 0.0)
        sq, p = q[:, 0], q[:, 1:]
        xyz = (sq * sw * winv)[:, None] * W + cw[:, None] * p + (sw * winv)[:, None] * np.cross(W, p)
        qw = sq * cw - np.sum(W * p, axis=1) * sw * winv
        qn = np.concatenate([qw[:, None], xyz], axis=1)
        qn /= np.linalg.norm(qn, axis=1, keepdims=True)
        q = np.where(mv[:, None], qn, q)
        if step == 0:
            np.testing.assert_array_equal(host(st.links.pairs), pairs)
        # the two solves agree to ~tol in the constraint gradient; bodies move O(0.1-1) per step in this overlapping
        # start, so 2e-4 absolute is a 1e-3..1e-4 relative check of the whole path (a sign or index slip is O(1))
        assert np.abs(host(st.center) - c).max() <= 2e-4, (step, np.abs(host(st.center) - c).max())
        assert np.abs(np.abs(np.sum(host(st.quat) * q, axis=1)) - 1.0).max() <= 1e-6, step
        c, q = host(st.center).copy(), host(st.quat).copy()   # re-anchor: compare per-step maps, not drift
    # overlaps are gone (to the linearisation) after the steps
    seg = oracle.spherocylinder_segments(c, q, r, L)
    pairs = oracle.search(oracle.SEARCH_AABB, *oracle.grow(oracle.compute_aabb_spherocylinders(c, q, r, L), brad, 0.0)[:2], c,
                          brad)
    assert oracle.contact_spherocylinders(pairs, seg, c)["sep"].min() > -0.05


def test_mixed_shape_stepper(mods):
    # BASELINE configs[4] through the stepper: spheres + spherocylinders + ellipsoids in one system, two full steps
    # (AABB -> neighbour list -> class-binned narrow phase -> LCP -> Euler / quaternion update)
    import torch
    from gpu_util import dev
    ops, pipeline, synth = mods
    b = synth.mixed_bodies(12_000, volume_fraction=0.25)
    tol = 1e-5
    st = pipeline.ContactStepper("mixed", dev(b["center"]), None, dev(b["quat"]), search_buffer=0.1,
                                 cfg=ops.PGDConfig(max_iters=20000, tol=tol), kinds=dev(b["kind"]), shape=dev(b["shape"]))
    worst0 = None
    for k in range(2):
        c_before = st.center.clone()
        s = st.step()
        assert s.converged and s.num_contacts > 10_000
        lam, sep = st.lam, st.contacts["sep"]
        g = st.op.apply(lam) + sep
        assert float(lam.min()) >= 0.0 and float(g.min()) >= -10 * tol
        assert float(torch.minimum(lam, g).abs().max()) <= 10 * tol
        if k == 0:
            worst0 = float(sep.min())
            assert worst0 < -0.05                       # the packing starts with real overlaps
        assert torch.isfinite(st.center).all() and torch.isfinite(st.quat).all()
        torch.testing.assert_close(st.quat.norm(dim=1), torch.ones_like(st.quat[:, 0]), rtol=0, atol=1e-12)
        assert float((st.center - c_before).abs().max()) > 0.0
    # bodies were pushed apart: after a step the deepest remaining linearised overlap is the tolerance, and the
    # recomputed geometric overlap shrank
    st.compute_aabb()
    st.generate_neighbor_links(force=True)
    assert float(st.compute_contacts()["sep"].min()) > worst0


@pytest.mark.parametrize("kind", ["sphere", "spherocylinder"])
def test_steps_with_no_contacts_and_single_bodies(mods, kind):
    # dilute systems: an empty neighbour list (nothing to solve: zero iterations, converged, bodies do not move), one
    # body, and two far bodies -- every stage must accept C = 0
    import torch
    from gpu_util import dev
    ops, pipeline, synth = mods
    for n in (1, 2, 300):
        if kind == "sphere":
            s = synth.spheres(n, volume_fraction=1e-4)
            st = pipeline.ContactStepper("sphere", dev(s["center"]), dev(s["radius"]), search_buffer=0.1)
        else:
            b = synth.spherocylinders(n, volume_fraction=1e-4)
            st = pipeline.ContactStepper("spherocylinder", dev(b["center"]), dev(b["radius"]), dev(b["quat"]),
                                         dev(b["length"]), search_buffer=0.1)
        c0 = st.center.clone()
        for _ in range(2):
            r = st.step()
            assert r.num_contacts == 0 and r.converged and r.num_iters == 0
        assert torch.equal(st.center, c0)
        assert st.op.body_velocity().abs().max().item() == 0.0 if n else True


def test_reordering_leaves_the_physics_unchanged(mods, oracle):
    # Morton and Hilbert reordering are permutations of the bodies: the same contacts (as a set of body pairs mapped
    # back through the permutation), the same LCP solution to solver tolerance, the same step
    import torch
    from gpu_util import dev, host
    ops, pipeline, synth = mods
    b = synth.spherocylinders(20_000, seed=3)
    tol = 1e-6

    def run(curve):
        st = pipeline.ContactStepper("spherocylinder", dev(b["center"]), dev(b["radius"]), dev(b["quat"]),
                                     dev(b["length"]), search_buffer=0.1, cfg=ops.PGDConfig(max_iters=20000, tol=tol))
        perm = None
        if curve:
            perm = host(st.reorder_bodies(cell_size=3.0, lo=[0.0, 0.0, 0.0], curve=curve, hi=[b["box"]] * 3, level=6))
        s = st.step()
        assert s.converged
        ids = np.arange(20_000) if perm is None else perm.astype(np.int64)
        p = ids[host(st.links.pairs).astype(np.int64)]
        p.sort(axis=1)
        order = np.lexsort((p[:, 1], p[:, 0]))
        g = host(st.op.apply(st.lam) + st.contacts["sep"])
        centers = np.empty((20_000, 3))
        centers[ids] = host(st.center)
        return p[order], g[order], centers

    p0, g0, c0 = run(None)
    for curve in ("morton", "hilbert"):
        p, g, c = run(curve)
        np.testing.assert_array_equal(p, p0)
        np.testing.assert_allclose(g, g0, atol=20 * tol)
        np.testing.assert_allclose(c, c0, atol=1e-4)


def test_trajectory_with_neighbour_list_reuse(mods, oracle):
    # A time loop as the reference's apps run it (NgpLcp.cpp:835-920): the neighbour list is rebuilt only when a body
    # has moved more than half the search buffer since the last build (GenNeighborLinkers.hpp:603-615), otherwise
    # reused.  The invariant that makes reuse safe: every pair that overlaps at the start of a step is in the list the
    # step uses -- checked against the oracle's search on the step's starting positions -- and steps with and without
    # a rebuild must both occur.  The overlapping random start relaxes: fewer iterations, shallower overlaps.
    ops, pipeline, synth = mods
    from gpu_util import dev, host
    b = synth.spherocylinders(6000, volume_fraction=0.2, seed=21)
    tol, dt, buf = 1e-6, 5e-3, 0.4
    st = pipeline.ContactStepper("spherocylinder", dev(b["center"]), dev(b["radius"]), dev(b["quat"]),
                                 dev(b["length"]), dt=dt, search_buffer=buf, cfg=ops.PGDConfig(max_iters=20000, tol=tol))
    r, L = b["radius"], b["length"]
    brad = oracle.bounding_radius_spherocylinders(r, L)
    rebuilt, iters, deepest = [], [], []
    for step in range(14):
        c0, q0 = host(st.center).copy(), host(st.quat).copy()
        s = st.step(integrate=True)
        assert s.converged, step
        rebuilt.append(bool(s.rebuilt))
        iters.append(s.num_iters)
        aabb = oracle.compute_aabb_spherocylinders(c0, q0, r, L)
        lo, hi, R = oracle.grow(aabb, brad, 0.0)
        pairs = oracle.search(oracle.SEARCH_AABB, lo, hi, c0, R)
        con = oracle.contact_spherocylinders(pairs, oracle.spherocylinder_segments(c0, q0, r, L), c0)
        touching = pairs[con["sep"] < 0.0].astype(np.int64)
        deepest.append(float(con["sep"].min()) if len(pairs) else 0.0)
        used = host(st.links.pairs).astype(np.int64)
        key = lambda p: p[:, 0] * len(r) + p[:, 1]  # noqa: E731
        missing = np.setdiff1d(key(touching), key(used))
        assert len(missing) == 0, (step, rebuilt, len(missing))
        c, q = host(st.center), host(st.quat)
        assert np.isfinite(c).all() and np.isfinite(q).all()
        np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-12)
    print("rebuilt", rebuilt, "iterations", iters, "deepest overlap at step start", ["%.3g" % d for d in deepest])
    assert rebuilt[0] and any(rebuilt[1:]) and not all(rebuilt[1:]), rebuilt
    assert iters[-1] < iters[0] and deepest[-1] > 0.1 * deepest[0]


def test_contact_cutoff_option_compacts_the_constraint_set(mods, oracle):
    # BUILD OPTION (off by default): only the candidate pairs within `contact_cutoff` of touching become constraints --
    # wavefront ballot / prefix-sum compaction of the narrow-phase output.  The kept set equals the numpy selection, the
    # solve reaches the full problem's solution (g is unique; the dropped pairs satisfy g >= 0, which is their LCP
    # condition with lambda = 0), with far fewer constraints per sweep.
    ops, pipeline, synth = mods
    from gpu_util import dev, host
    b = synth.spherocylinders(8000, seed=5)
    tol = 1e-6
    # a relaxed packing (what a running simulation sees): the raw random packing overlaps so deeply that bodies move by
    # more than a rod radius within the one step that separates them, and no useful cutoff survives the check
    pre = pipeline.ContactStepper("spherocylinder", dev(b["center"]), dev(b["radius"]), dev(b["quat"]), dev(b["length"]),
                                  search_buffer=0.3, cfg=ops.PGDConfig(max_iters=20000, tol=tol))
    for _ in range(4):
        pre.step(integrate=True, force_rebuild=True)
    c0, q0 = pre.center.clone(), pre.quat.clone()
    mk = lambda **kw: pipeline.ContactStepper("spherocylinder", c0.clone(), dev(b["radius"]), q0.clone(),  # noqa: E731
                                              dev(b["length"]), search_buffer=0.3,
                                              cfg=ops.PGDConfig(max_iters=20000, tol=tol), **kw)
    full, cut = mk(), mk(contact_cutoff=0.02)
    sf, sc = full.step(integrate=False), cut.step(integrate=False)
    assert sf.converged and sc.converged and cut.cutoff_fallbacks == 0
    sep_all = host(full.contacts["sep"])
    want = np.flatnonzero(~(sep_all > 0.02))
    assert sc.num_contacts == len(want) and sc.num_contacts < 0.6 * sf.num_contacts
    np.testing.assert_array_equal(host(cut.contact_pairs), host(full.links.pairs)[want])
    assert np.array_equal(host(cut.contacts["sep"]), sep_all[want])
    assert np.array_equal(host(cut.contacts["normal"]), host(full.contacts["normal"])[want])
    gf = host(full.op.apply(full.lam) + full.contacts["sep"])
    gc = host(cut.op.apply(cut.lam) + cut.contacts["sep"])
    np.testing.assert_allclose(gc, gf[want], atol=20 * tol)
    dropped = np.setdiff1d(np.arange(len(sep_all)), want)
    assert gf[dropped].min() >= -10 * tol                       # the pairs left out are inactive in the full solution
    np.testing.assert_allclose(host(cut.op.body_velocity()), host(full.op.body_velocity()), atol=1e-3 * np.abs(host(full.op.body_velocity())).max())
    # a cutoff so tight that overlapping pairs are dropped is caught by the check and the full list decides
    bad = mk(contact_cutoff=-0.2)
    sb = bad.step(integrate=False)
    assert bad.cutoff_fallbacks == 1 and sb.num_contacts == sf.num_contacts and sb.converged


def test_periodic_rods_10k_step_matches_oracle(mods, oracle):
    # spherocylinders in an orthorhombic periodic box -- periodic neighbour search, contacts against the
    # nearest image of the partner (PeriodicScaledMetric::sep of the centres, rigid translation), LCP, Euler update and
    # wrap_rigid of the centres (periodicity.hpp:812-823, :1094-1113), each stage against the CPU oracle
    ops, pipeline, synth = mods
    from gpu_util import assert_bits_equal, dev, host
    b = synth.spherocylinders(10_000, volume_fraction=0.3, seed=77)
    box = [b["box"]] * 3
    c0, q0, r, L = b["center"], b["quat"], b["radius"], b["length"]
    tol, dt, buf = 1e-6, 5e-3, 0.2
    st = pipeline.ContactStepper("spherocylinder", dev(c0), dev(r), dev(q0), dev(L), dt=dt, search_buffer=buf,
                                 periodic_box=box, cfg=ops.PGDConfig(max_iters=20000, tol=tol))
    s = st.step(integrate=True, force_rebuild=True)
    aabb = oracle.compute_aabb_spherocylinders(c0, q0, r, L)
    brad = oracle.bounding_radius_spherocylinders(r, L)
    lo, hi, R = oracle.grow(aabb, brad, buf)
    pairs = oracle.search(oracle.SEARCH_AABB, lo, hi, c0, R, box=box)
    np.testing.assert_array_equal(host(st.links.pairs), pairs)
    free = oracle.search(oracle.SEARCH_AABB, lo, hi, c0, R)
    assert len(pairs) > len(free) + 100                  # the box faces do contribute pairs
    seg = oracle.spherocylinder_segments(c0, q0, r, L)
    con = oracle.contact_spherocylinders(pairs, seg, c0, box=box)
    for k in ("sep", "normal", "s", "t"):
        assert_bits_equal(host(st.contacts[k]), con[k], k)
    # a pair across a face really is evaluated against the image: its separation is small, not a box length
    far = np.abs(c0[pairs[:, 0]] - c0[pairs[:, 1]]).max(axis=1) > 0.5 * box[0]
    assert far.sum() > 100 and con["sep"][far].max() < 3.5
    mt, mr = synth.dry_mobility(r, bounding_radius=brad)
    with oracle.compensated_sums():
        x, g, ro = oracle.solve_cqpp_contact(pairs, con["normal"], None, None, mt, mr, dt, con["sep"], np.zeros(len(pairs)),
                                             max_iters=20000, tol=tol, rod=(con["s"], con["t"], seg))
        _, vel = oracle.contact_op_apply(pairs, con["normal"], None, None, mt, mr, dt, x, len(r),
                                         rod=(con["s"], con["t"], seg), body_velocity=True)
    assert s.converged and ro["converged"] and abs(s.num_iters - ro["num_iters"]) <= 2
    c1, q1 = oracle.integrate_euler(dt, vel, c0, q0)
    c1 = oracle.periodic_wrap(box, c1)
    got = host(st.center)
    assert got.min() >= 0.0 and got.max() < box[0]
    d = got - c1
    d -= np.round(d / box[0]) * box[0]                   # a centre within rounding of a face may wrap either way
    assert np.abs(d).max() <= 1e-9
    assert np.abs(np.abs(np.sum(host(st.quat) * q1, axis=1)) - 1.0).max() <= 1e-12
    # images of the same system give the same step: shift every body by whole box vectors
    shift = np.random.default_rng(0).integers(-2, 3, c0.shape) * box[0]
    st2 = pipeline.ContactStepper("spherocylinder", dev(c0 + shift), dev(r), dev(q0), dev(L), dt=dt, search_buffer=buf,
                                  periodic_box=box, cfg=ops.PGDConfig(max_iters=20000, tol=tol))
    s2 = st2.step(integrate=False, force_rebuild=True)
    np.testing.assert_array_equal(host(st2.links.pairs), pairs)
    np.testing.assert_allclose(host(st2.contacts["sep"]), con["sep"], atol=1e-11)


def test_periodic_mixed_shapes_contacts_match_oracle(mods, oracle):
    # the same for the mixed sphere / spherocylinder / ellipsoid system: partner bodies at the nearest image of their
    # centre; the cheap classes bit for bit, the L-BFGS classes at the reference's 1e-4 (see test_gpu_mixed.py)
    ops, pipeline, synth = mods
    from gpu_util import dev, host
    b = synth.mixed_bodies(9000, volume_fraction=0.25, seed=5)
    box = [b["box"]] * 3
    st = pipeline.ContactStepper("mixed", dev(b["center"]), None, dev(b["quat"]), search_buffer=0.15, periodic_box=box,
                                 cfg=ops.PGDConfig(max_iters=20000, tol=1e-5), kinds=dev(b["kind"]), shape=dev(b["shape"]))
    s = st.step(integrate=True, force_rebuild=True)
    assert s.converged
    aabb, brad = oracle.aabb_mixed(b["kind"], b["center"], b["quat"], b["shape"])
    lo, hi, R = oracle.grow(aabb, brad, 0.15)
    pairs = oracle.search(oracle.SEARCH_AABB, lo, hi, b["center"], R, box=box)
    np.testing.assert_array_equal(host(st.links.pairs), pairs)
    con = oracle.contact_mixed(pairs, b["kind"], b["center"], b["quat"], b["shape"], box=box)
    ka, kb = b["kind"][pairs[:, 0]], b["kind"][pairs[:, 1]]
    cheap = (np.maximum(ka, kb) < 2)                      # S-S, S-R, R-R
    far = np.abs(b["center"][pairs[:, 0]] - b["center"][pairs[:, 1]]).max(axis=1) > 0.5 * box[0]
    assert (far & cheap).sum() > 30 and (far & ~cheap).sum() > 30
    got = {k: host(st.contacts[k]) for k in ("sep", "normal", "ra", "rb")}
    for k in got:
        assert np.array_equal(got[k][cheap], con[k][cheap]), k
    ok = np.abs(got["sep"][~cheap] - con["sep"][~cheap]) <= 1e-4
    assert ok.mean() >= 0.995
    assert got["sep"][far].max() < 3.0
    c = host(st.center)
    assert c.min() >= 0.0 and c.max() < box[0]
