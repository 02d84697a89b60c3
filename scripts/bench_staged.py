"""Host overhead of the staged (multi-rank) solver path, measured at world size 1 over nccl on one GPU: the same 10^6-rod
step through DistributedContactStepper (Python-driven stages + all-gather per iteration) vs the fused C++ driver."""
import sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, ".")
from mundy_amd import distributed as D, ops, pipeline, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29733", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
b = synth.spherocylinders(n)
order = D.hilbert_order(b["center"], 0.0, b["box"], level=7)
c, q, r, L = (b[k][order] for k in ("center", "quat", "radius", "length"))
cfg = ops.PGDConfig(max_iters=10000, tol=1e-5)
st = D.DistributedContactStepper(dev(c), dev(q), dev(r), dev(L), 0, comm=D.Comm(), cfg=cfg, poll_every=32)
ref = pipeline.ContactStepper("spherocylinder", dev(c), dev(r), dev(q), dev(L), search_buffer=0.1, cfg=cfg)
st0 = D.DistributedContactStepper(dev(c), dev(q), dev(r), dev(L), 0, comm=D.Comm(), cfg=cfg, poll_every=32)
st0.tiering = 0
st1 = D.DistributedContactStepper(dev(c), dev(q), dev(r), dev(L), 0, comm=D.Comm(mailbox=False), cfg=cfg, poll_every=32)
assert st.comm.mailbox and not st1.comm.mailbox
for name, fn in (("staged (records through the mailbox)", lambda: st.step(integrate=False)),
                 ("staged, records through ncclAllGather", lambda: st1.step(integrate=False)),
                 ("staged, cold tier off", lambda: st0.step(integrate=False)),
                 ("fused", lambda: ref.step(integrate=False))):
    fn(); fn(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        t = time.perf_counter(); out = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    dt = float(np.median(ts))
    it = out["num_iters"] if isinstance(out, dict) else out.num_iters
    print("%s: %.1f ms/step, %d iterations, %.3f ms/iteration" % (name, 1e3 * dt, it, 1e3 * dt / max(it, 1)))
print("staged cold tier:", st.op.tier_stats())
print("fused stage ms:", {k: round(v, 2) for k, v in ref.step(integrate=False, timed=True).timings_ms.items()})
st.profile = True
st.prof.update(body_ms=0.0, con_ms=0.0, iters=0)
out = st.step(integrate=False)
print("staged sampled sweeps: k_body %.4f ms, k_constraint+local3 %.4f ms over %d iterations; stats %s"
      % (st.prof["body_ms"] / max(1, st.prof["iters"]), st.prof["con_ms"] / max(1, st.prof["iters"]), st.prof["iters"], out))
print("staged phases (ms, with syncs):", {k: round(v, 2) for k, v in st.phase_ms.items()})
dist.destroy_process_group()
