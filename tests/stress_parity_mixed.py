"""Randomised parity sweep of the mixed-shape path (not part of the test-suite; run on the GPU box): systems of spheres,
rods and ellipsoids of varied size / density / shape through AABBs, neighbour list, class-binned narrow phase and the
LCP -- every output of every class bit for bit against the oracle (shared sincos), BBPGD iteration counts equal."""
import os
import sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from mundy_amd import ops, synth
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
host = lambda t: t.detach().cpu().numpy()
rng = np.random.default_rng(2027)
bad = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    n = int(rng.choice([600, 3000, 9000, 24000]))
    phi = float(rng.uniform(0.1, 0.45))
    buf = float(rng.uniform(0.02, 0.2))
    rs = float(rng.uniform(0.3, 0.8))
    rod = (float(rng.uniform(0.2, 0.6)), float(rng.uniform(0.0, 3.0)))
    ell = tuple(float(v) for v in rng.uniform(0.25, 1.1, 3))
    if case % 4 == 3:
        ell = (ell[0], ell[0], ell[2])   # a spheroid: equal semi-axes take the special branches of the closed forms
    b = synth.mixed_bodies(n, volume_fraction=phi, seed=500 + case, sphere_radius=rs, rod=rod, ellipsoid=ell)
    kind, c, q, shape = b["kind"], b["center"], b["quat"], b["shape"]
    dk, dc, dq, ds = dev(kind), dev(c), dev(q), dev(shape)
    aabb, brad = ops.compute_aabb_mixed(dk, dc, dq, ds)
    oaabb, obrad = oracle.aabb_mixed(kind, c, q, shape)
    links = ops.GenNeighborLinks().set_search_kind(ops.SEARCH_AABB).set_search_buffer(buf).concretize()
    links.generate(aabb, dc, brad)
    lo, hi, R = oracle.grow(oaabb, obrad, buf)
    pairs = oracle.search(oracle.SEARCH_AABB, lo, hi, c, R)
    same_pairs = np.array_equal(host(links.pairs), pairs)
    out = ops.contact_mixed(links.pairs, dk, dc, dq, ds, want_counts=True)
    with oracle.shared_trig():
        exp = oracle.contact_mixed(pairs, kind, c, q, shape)
    same = same_pairs and all(np.array_equal(host(out[k]).view(np.uint64), exp[k].view(np.uint64))
                              for k in ("sep", "normal", "cp1", "cp2", "ra", "rb"))
    mt, mr = synth.dry_mobility(obrad)
    tol = 1e-6
    op = ops.ContactOperator(links.pairs, out["normal"], dev(mt), 5e-3, ra=out["ra"], rb=out["rb"], mob_rot=dev(mr))
    op.set_tiering(3)   # the cold tier whatever the size: the harder path
    x, g, res = ops.solve_lcp(op, out["sep"], torch.zeros_like(out["sep"]), ops.PGDConfig(max_iters=50000, tol=tol))
    with oracle.compensated_sums():   # the device's definition of the sums
        ox, og, ores = oracle.solve_cqpp_contact(pairs, exp["normal"], exp["ra"], exp["rb"], mt, mr, 5e-3, exp["sep"],
                                                 np.zeros(len(pairs)), max_iters=50000, tol=tol)
    dg = float(np.abs(host(g) - og).max()) if len(pairs) else 0.0
    ok = same and res.converged and res.num_iters == ores["num_iters"] and dg <= 20 * tol
    bad += 0 if ok else 1
    print("%s case %2d: n=%5d phi=%.2f buf=%.2f sphere %.2f rod (%.2f, %.2f) ellipsoid (%.2f, %.2f, %.2f) contacts=%7d "
          "classes %s iters gpu/oracle %5d/%5d |dg|=%.2e pairs %s contacts bitwise %s"
          % ("ok  " if ok else "FAIL", case, n, phi, buf, rs, rod[0], rod[1], ell[0], ell[1], ell[2], len(pairs),
             [out["class_counts"][k] for k in ("SS", "SR", "SE", "RR", "RE", "EE")], res.num_iters, ores["num_iters"], dg,
             same_pairs, same), flush=True)
    op.close()
    links.close()
print("STRESS PASS (mixed)" if bad == 0 else "STRESS FAIL (mixed): %d" % bad)
sys.exit(0 if bad == 0 else 1)
