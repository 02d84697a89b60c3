"""Sphere systems (BASELINE configs[1] and a 10^6 version): step time and per-sweep times of the 3-DOF operator."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mundy_amd import ops, pipeline, synth
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
# buffer 0.25 as in the parity test of configs[1].  (With buffer 1.0 this overlapping start does not converge within
# 10^4 BBPGD iterations at 10^5 spheres -- on the GPU and on the CPU oracle alike, residual 6e-3: gaps up to 2 make the
# step's LCP long-ranged.)
for n, buf in ((100_000, 0.25), (1_000_000, 0.25)):
    s = synth.spheres(n, volume_fraction=0.4)
    st = pipeline.ContactStepper("sphere", dev(s["center"]), dev(s["radius"]), search_buffer=buf,
                                 search_kind=ops.SEARCH_SPHERES, cfg=ops.PGDConfig(max_iters=10000, tol=1e-5))
    st.step(integrate=False)
    st.profile_next = True
    r = st.step(integrate=False, force_rebuild=True, timed=True)
    bm, cm, k = st.op.get_profile()
    print("spheres n=%d buffer=%.2f: contacts %d, iterations %d, solve %.1f ms, broadphase %.2f ms; per sweep k_body %.4f ms, "
          "k_constraint %.4f ms" % (n, buf, r.num_contacts, r.num_iters, r.timings_ms["solve"], r.timings_ms["broadphase"],
                                    bm / max(k, 1), cm / max(k, 1)), flush=True)
