"""One full step of a mixed sphere / rod / ellipsoid system (BASELINE configs[4] shapes) on one GPU, stage by stage."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mundy_amd import ops, pipeline, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
phi = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
b = synth.mixed_bodies(n, volume_fraction=phi)
st = pipeline.ContactStepper("mixed", dev(b["center"]), None, dev(b["quat"]), search_buffer=0.1,
                             cfg=ops.PGDConfig(max_iters=10000, tol=1e-5), kinds=dev(b["kind"]), shape=dev(b["shape"]))
snap = st.snapshot()
st.step()
st.restore(snap)
s = st.step(force_rebuild=True, timed=True)
print("mixed %d bodies phi %.2f: contacts %d, BBPGD iterations %d (converged %s), stages ms %s"
      % (n, phi, s.num_contacts, s.num_iters, s.converged, {k: round(v, 2) for k, v in s.timings_ms.items()}), flush=True)
