"""Wall time of building (and destroying) the rod contact operator on the bench input: incidence index, half-edge
records, activity-mask slots, plus its device allocations."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mundy_amd import ops, pipeline, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
b = synth.spherocylinders(n)
st = pipeline.ContactStepper("spherocylinder", dev(b["center"]), dev(b["radius"]), dev(b["quat"]), dev(b["length"]),
                             search_buffer=0.1, cfg=ops.PGDConfig(max_iters=5, tol=1e-5))
st.reorder_bodies(cell_size=3.0, lo=[0.0, 0.0, 0.0])
st.step(integrate=False)
con, pairs = st.contacts, st.links.pairs
seg = ops.spherocylinder_segments(st.center, st.quat, st.radius, st.length)
con = ops.contact_spherocylinders(pairs, seg, st.center, want_points=False, arms="arclength")
torch.cuda.synchronize()
tb, td = [], []
for _ in range(8):
    t0 = time.perf_counter()
    op = ops.ContactOperator(pairs, con["normal"], st.mob_trans, 5e-3, mob_rot=st.mob_rot, rod=(con["s"], con["t"], seg),
                             priority=con["sep"])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    op.close()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tb.append(t1 - t0)
    td.append(t2 - t1)
print("operator build %.3f ms, destroy %.3f ms (median of 8; %d contacts)" % (1e3 * np.median(tb), 1e3 * np.median(td), pairs.shape[0]))
