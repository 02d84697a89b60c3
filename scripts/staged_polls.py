import sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, ".")
from mundy_amd import distributed as D, ops, pipeline, synth
import os
n = int(os.environ.get("N_RODS", "1000000"))
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29737", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
b = synth.spherocylinders(n)
order = D.hilbert_order(b["center"], 0.0, b["box"], level=7)
c, q, r, L = (b[k][order] for k in ("center", "quat", "radius", "length"))
cfg = ops.PGDConfig(max_iters=10000, tol=1e-5)
comm = D.Comm()
for poll in [int(a) for a in sys.argv[1:]] or (16, 32, 64, 128):
    st = D.DistributedContactStepper(dev(c), dev(q), dev(r), dev(L), 0, comm=comm, cfg=cfg, poll_every=poll)
    st.step(integrate=False); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        t = time.perf_counter(); out = st.step(integrate=False); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    dt = float(np.median(ts))
    print("poll_every %d: %.2f ms/step, %d iterations, phases %s, tier %s" % (poll, 1e3 * dt, out["num_iters"], {k: round(v, 2) for k, v in st.phase_ms.items()}, st.op.tier_stats()), flush=True)
    st.op.close()
ref = pipeline.ContactStepper("spherocylinder", dev(c), dev(r), dev(q), dev(L), search_buffer=0.1, cfg=cfg)
ref.step(integrate=False); torch.cuda.synchronize()
t = time.perf_counter(); out = ref.step(integrate=False); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("fused: %.1f ms/step, tier %s" % (1e3 * dt, ref.op.tier_stats()))
dist.destroy_process_group()
