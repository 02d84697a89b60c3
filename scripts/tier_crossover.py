"""Where the cold tier starts to pay: the same rod step with the tier on and off, fused and staged (world 1 over nccl),
at a ladder of sizes.  Median of 5 solves each."""
import sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, ".")
from mundy_amd import distributed as D, ops, pipeline, synth
sizes = [int(a) for a in sys.argv[1:]] or [30000, 60000, 125000, 250000, 500000]
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29735", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
import os
cfg = ops.PGDConfig(max_iters=10000, tol=float(os.environ.get("TOL", "1e-5")))
comm = D.Comm()
def med(fn):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        t = time.perf_counter(); out = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts)), out
for n in sizes:
    b = synth.spherocylinders(n)
    order = D.hilbert_order(b["center"], 0.0, b["box"], level=7)
    c, q, r, L = (b[k][order] for k in ("center", "quat", "radius", "length"))
    row = []
    for mode in (1, 3, 0):
        st = D.DistributedContactStepper(dev(c), dev(q), dev(r), dev(L), 0, comm=comm, cfg=cfg, poll_every=32)
        st.tiering = mode
        st.profile = True
        ms, out = med(lambda: st.step(integrate=False))
        row.append("staged[%d] %.2f ms (%d it, C %d)" % (mode, st.phase_ms["solve"], out["num_iters"], out["local_contacts"]))
        st.op.close()
        ref = pipeline.ContactStepper("spherocylinder", dev(c), dev(r), dev(q), dev(L), search_buffer=0.1, cfg=cfg)
        ref.tiering = mode
        ms, out = med(lambda: ref.step(integrate=False, timed=True))
        row.append("fused[%d] %.2f ms" % (mode, out.timings_ms["solve"]))
    print(n, "; ".join(row), flush=True)
dist.destroy_process_group()
