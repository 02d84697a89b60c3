"""Randomised parity sweep for sphere systems (free space and periodic boxes, bounding-sphere and AABB searches):
every stage of the stepper against the oracle.  Not part of the test-suite; run on the GPU box."""
import os
import sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from mundy_amd import ops, pipeline, synth
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rng = np.random.default_rng(77)
bad = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    n = int(rng.choice([500, 2000, 10000, 40000]))
    phi = float(rng.uniform(0.1, 0.35))
    buf = float(rng.uniform(0.05, 0.4))
    periodic = bool(rng.integers(0, 2))
    kind = int(rng.integers(0, 2))  # 0 = bounding spheres, 1 = AABB
    tol = 1e-6
    s = synth.spheres(n, volume_fraction=phi, seed=500 + case)
    c, r = s["center"], s["radius"] * rng.uniform(0.7, 1.0, n)      # polydisperse
    box = [s["box"]] * 3 if periodic else None
    st = pipeline.ContactStepper("sphere", dev(c), dev(r), search_buffer=buf, search_kind=kind, periodic_box=box,
                                 cfg=ops.PGDConfig(max_iters=50000, tol=tol))
    st.tiering = 3      # the cold tier whatever the size (by default only from 1.5M contacts on): the harder path
    res = st.step(integrate=False)
    lo, hi, R = oracle.grow(oracle.compute_aabb_spheres(c, r), r, buf)
    pairs = oracle.search(kind, lo, hi, c, R, box=box)
    sep, nrm = oracle.contact_spheres(pairs, c, r, box=box)
    mt, _ = synth.dry_mobility(r)
    with oracle.compensated_sums():
        xo, go, ro = oracle.solve_cqpp_contact(pairs, nrm, None, None, mt, None, 5e-3, sep, np.zeros(len(pairs)),
                                               max_iters=50000, tol=tol)
    ok_pairs = np.array_equal(st.links.pairs.cpu().numpy(), pairs)
    ok_sep = ok_pairs and np.array_equal(st.contacts["sep"].cpu().numpy(), sep) and \
        np.array_equal(st.contacts["normal"].cpu().numpy(), nrm)
    g = (st.op.apply(st.lam) + st.contacts["sep"]).cpu().numpy() if ok_pairs else None
    dg = float(np.abs(g - go).max()) if ok_pairs and len(pairs) else 0.0
    ok = ok_pairs and ok_sep and res.converged and ro["converged"] and dg <= 20 * tol and \
        abs(res.num_iters - ro["num_iters"]) <= 2
    bad += not ok
    print("%s case %2d: n=%5d phi=%.2f buf=%.2f %s %s contacts=%7d iters gpu/oracle %5d/%5d |dg|=%.2e pairs %s sep/normal %s"
          % ("ok  " if ok else "FAIL", case, n, phi, buf, "periodic" if periodic else "free    ",
             "AABB   " if kind else "spheres", len(pairs), res.num_iters, ro["num_iters"], dg, ok_pairs, ok_sep), flush=True)
print("STRESS", "PASS" if bad == 0 else "FAIL (%d)" % bad)
