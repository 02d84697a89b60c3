"""Cold-tier statistics of the mixed-shape step (BASELINE configs[4] shapes, vector-arm operator)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mundy_amd import ops, pipeline, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
b = synth.mixed_bodies(n, volume_fraction=0.4)
st = pipeline.ContactStepper("mixed", dev(b["center"]), None, dev(b["quat"]), None, kinds=dev(b["kind"]), shape=dev(b["shape"]),
                             search_buffer=0.1, cfg=ops.PGDConfig(max_iters=10000, tol=1e-5))
for k in range(2):
    s = st.step(timed=True)
    print(k, s.num_iters, s.num_contacts, {a: round(v, 2) for a, v in s.timings_ms.items()}, st.op.tier_stats())
    for mode in (0,):
        pass
