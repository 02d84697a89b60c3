"""Body-order A/B on one GPU: per-sweep time of the fused BBPGD iteration for the same 10^6-rod packing stored in
generator (random) order, Morton order (lattice edge 3 / 1.5) and Hilbert order (levels 5..8)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from mundy_amd import distributed as D, ops, pipeline, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
b = synth.spherocylinders(n)
cfg = ops.PGDConfig(max_iters=200, tol=1e-5)


def run(label, order=None, morton_cell=None):
    c, q, r, L = (b[k] if order is None else b[k][order] for k in ("center", "quat", "radius", "length"))
    st = pipeline.ContactStepper("spherocylinder", dev(c), dev(r), dev(q), dev(L), search_buffer=0.1, cfg=cfg)
    if morton_cell:
        st.reorder_bodies(cell_size=morton_cell, lo=[0.0, 0.0, 0.0])
    st.profile_next = True
    st.step(integrate=False)
    bm, cm, k = st.op.get_profile()
    print("%-22s k_body %.4f ms  k_constraint %.4f ms  (%d timed iterations)" % (label, bm / k, cm / k, k), flush=True)


run("generator order")
run("morton cell 3.0", morton_cell=3.0)
run("morton cell 1.5", morton_cell=1.5)
run("morton cell 0.75", morton_cell=0.75)
for lvl in (5, 6, 7, 8, 9):
    run("hilbert level %d" % lvl, order=D.hilbert_order(b["center"], 0.0, b["box"], level=lvl))
