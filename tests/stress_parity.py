"""Randomised parity sweep (not part of the test-suite; run on the GPU box): rod systems of varied size / density / buffer /
dt through the stepper, every stage against the oracle -- neighbour list, separations, normals bit for bit; LCP
gradient to 20 tol, BBPGD iteration count within 2 of the oracle's (compensated sums on both sides)."""
import os
import sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from mundy_amd import ops, pipeline, synth
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rng = np.random.default_rng(2026)
bad = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    n = int(rng.choice([300, 1000, 3000, 8000, 20000]))
    phi = float(rng.uniform(0.08, 0.45))
    buf = float(rng.uniform(0.03, 0.3))
    dt = float(rng.choice([1e-3, 5e-3, 2e-2]))
    L = float(rng.uniform(0.5, 3.0))
    tol = 1e-6
    b = synth.spherocylinders(n, seed=1000 + case, volume_fraction=phi, length=L)
    st = pipeline.ContactStepper("spherocylinder", dev(b["center"]), dev(b["radius"]), dev(b["quat"]), dev(b["length"]),
                                 dt=dt, search_buffer=buf, cfg=ops.PGDConfig(max_iters=50000, tol=tol))
    st.tiering = 3      # the cold tier whatever the size (by default only from 1.5M contacts on): the harder path
    s = st.step(integrate=False)
    aabb = oracle.compute_aabb_spherocylinders(b["center"], b["quat"], b["radius"], b["length"])
    brad = oracle.bounding_radius_spherocylinders(b["radius"], b["length"])
    lo, hi, R = oracle.grow(aabb, brad, buf)
    pairs = oracle.search(oracle.SEARCH_AABB, lo, hi, b["center"], R)
    seg = oracle.spherocylinder_segments(b["center"], b["quat"], b["radius"], b["length"])
    out = oracle.contact_spherocylinders(pairs, seg, b["center"])
    mt, mr = synth.dry_mobility(b["radius"], bounding_radius=brad)
    with oracle.compensated_sums():   # the device's definition of the sums, rod-axis form of the operator, serial
        xo, go, ro = oracle.solve_cqpp_contact(pairs, out["normal"], None, None, mt, mr, dt, out["sep"],
                                               np.zeros(len(pairs)), max_iters=50000, tol=tol,
                                               rod=(out["s"], out["t"], seg))
    ok_pairs = np.array_equal(st.links.pairs.cpu().numpy(), pairs)
    ok_sep = ok_pairs and np.array_equal(st.contacts["sep"].cpu().numpy(), out["sep"]) and \
        np.array_equal(st.contacts["normal"].cpu().numpy(), out["normal"])
    g = (st.op.apply(st.lam) + st.contacts["sep"]).cpu().numpy() if ok_pairs else None
    dg = float(np.abs(g - go).max()) if ok_pairs and len(pairs) else 0.0
    ok = ok_pairs and ok_sep and s.converged and ro["converged"] and dg <= 20 * tol and \
        abs(s.num_iters - ro["num_iters"]) <= 2
    bad += not ok
    print("%s case %2d: n=%5d phi=%.2f buf=%.2f dt=%.0e L=%.1f contacts=%7d iters gpu/oracle %5d/%5d  |dg|=%.2e  pairs %s sep/normal %s"
          % ("ok  " if ok else "FAIL", case, n, phi, buf, dt, L, len(pairs), s.num_iters, ro["num_iters"], dg, ok_pairs, ok_sep),
          flush=True)
print("STRESS", "PASS" if bad == 0 else "FAIL (%d)" % bad)
