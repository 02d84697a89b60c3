"""Fraction of contacts that are 'active' (not x == 0 with g >= 0) along a BBPGD solve of the 10^6-rod step."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from mundy_amd import ops, pipeline, synth
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
b = synth.spherocylinders(1_000_000)
for iters in (1, 5, 20, 100, 400, 10000):
    st = pipeline.ContactStepper("spherocylinder", dev(b["center"]), dev(b["radius"]), dev(b["quat"]), dev(b["length"]),
                                 search_buffer=0.1, cfg=ops.PGDConfig(max_iters=iters, tol=1e-5))
    st.reorder_bodies(cell_size=3.0, lo=[0.0, 0.0, 0.0])
    s = st.step(integrate=False)
    lam = st.lam
    g = st.op.apply(lam) + st.contacts["sep"]
    inactive = (lam == 0) & (g >= 0)
    deg = torch.zeros(1_000_000, dtype=torch.int64, device="cuda")
    act_pairs = st.links.pairs[~inactive].long()
    deg.index_add_(0, act_pairs[:, 0], torch.ones_like(act_pairs[:, 0]))
    deg.index_add_(0, act_pairs[:, 1], torch.ones_like(act_pairs[:, 1]))
    print("after %5d iterations (converged %s): active %.3f of %d contacts; lam > 0: %.3f; active half-edges per body mean %.2f max %d; sep < 0: %.3f"
          % (s.num_iters, s.converged, 1 - inactive.double().mean().item(), lam.numel(), (lam > 0).double().mean().item(),
             deg.double().mean().item(), int(deg.max()), (st.contacts["sep"] < 0).double().mean().item()), flush=True)
