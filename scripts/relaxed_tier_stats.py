import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from mundy_amd import ops, pipeline, synth
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
b = synth.spherocylinders(1_000_000)
st = pipeline.ContactStepper("spherocylinder", dev(b["center"]), dev(b["radius"]), dev(b["quat"]), dev(b["length"]), search_buffer=0.1, cfg=ops.PGDConfig(max_iters=10000, tol=1e-5))
st.reorder_bodies(cell_size=3.0, lo=[0.0, 0.0, 0.0])
for k in range(4):
    s = st.step()
    print(k, s.num_iters, s.num_contacts, st.op.tier_stats())
