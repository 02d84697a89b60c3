"""An EXPECTATION for the strong-scaling curve of configs[3] (10^6 rods over N GPUs), from what one GPU can measure:
for N = 1, 2, 4, 8 the share of a MIDDLE rank of the real Hilbert partition of the bench system -- its owned range plus
the ghost bodies it would hold (every body that has a neighbour pair with an owned one) -- runs as a system of its own
through the staged / distributed driver at world size 1 over RCCL (the code path of a rank, reductions through the
mailbox).  Measured: the set-up stages of a step and the cost of one BBPGD iteration at that size.  The global solve
needs the SAME 770 iterations whatever N (every sum that feeds the BB step is rounded once: the partition does not reach
the iterates), so
    predicted ms/step (N) = set-up (N) + 770 x (t_iteration (N) + t_halo_launches + 2 x t_exchange)
with t_halo_launches = the push + collect launches of the inbox halo a single rank does not run (measured here as two
empty launches) and t_exchange = the latency of one posted-write exchange between GPUs -- the one number a single GPU
cannot measure: the table carries 0, 3 and 8 us.  What the model leaves out: load imbalance between ranks (the bench
re-cuts the curve by work first: max / mean contacts 1.00-1.02 in the 2-rank runs here), and the wait for the slowest
rank inside every exchange (OS jitter), both of which only lengthen the iteration.

python scripts/scaling_model.py  >  profiles/r04_scaling_model.txt   (on a GPU box; ~1 min)"""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mundy_amd import distributed as D, ops, pipeline, synth

ITERS = 770
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29741", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
n_total = 1_000_000
b = synth.spherocylinders(n_total, seed=1234)
order = D.hilbert_order(b["center"], 0.0, b["box"], level=7)
c, q, r, L = (b[k][order] for k in ("center", "quat", "radius", "length"))
cfg = ops.PGDConfig(max_iters=10000, tol=1e-5)
# the whole system's neighbour list once (who is whose neighbour decides the ghosts)
ref = pipeline.ContactStepper("spherocylinder", dev(c), dev(r), dev(q), dev(L), search_buffer=0.1, cfg=cfg)
ref.compute_aabb()
ref.generate_neighbor_links(force=True)
pairs = ref.links.pairs.cpu().numpy()
comm = D.Comm()
# two empty launches on the stream, as the halo's push + collect cost a rank that has nothing else to wait for
x = torch.zeros(64, device="cuda")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    x.zero_()
    x.zero_()
e1.record()
torch.cuda.synchronize()
t_launch2 = 1e3 * e0.elapsed_time(e1) / 200      # us for two back-to-back tiny launches
print("# scripts/scaling_model.py: expectation for `bench.py --gpus N` (configs[3], strong scaling of the 10^6-rod system)")
print("# a middle rank's share (owned + ghosts) through the staged driver at world 1 over RCCL; %d iterations; two tiny "
      "launches back to back: %.1f us" % (ITERS, t_launch2))
print("%2s %9s %8s %10s %9s %10s %11s | %s" % ("N", "owned", "ghosts", "contacts", "setup ms", "iters run", "us/iter", "predicted ms/step -> timesteps/s at t_exchange = 0 / 3 / 8 us"))
for N in (1, 2, 4, 8):
    starts = D.partition_ranges(n_total, N)
    k = N // 2 if N > 1 else 0
    a, e = int(starts[k]), int(starts[k + 1])
    own = np.zeros(n_total, bool)
    own[a:e] = True
    touch = own[pairs[:, 0]] | own[pairs[:, 1]]
    local = own.copy()
    local[pairs[touch].reshape(-1)] = True
    idx = np.nonzero(local)[0]                 # Hilbert order kept
    ghosts = int(local.sum() - own.sum())
    st = D.DistributedContactStepper(dev(c[idx]), dev(q[idx]), dev(r[idx]), dev(L[idx]), 0, comm=comm, cfg=cfg, poll_every=64)
    st.step(integrate=False)
    st.profile = True
    ts, its = [], []
    for _ in range(3):
        s = st.step(integrate=False)
        ts.append(dict(st.phase_ms))
        its.append(s["num_iters"])
    ph = {k2: float(np.median([t[k2] for t in ts])) for k2 in ts[0]}
    setup = sum(v for k2, v in ph.items() if k2 not in ("solve", "start"))
    per_it = 1e3 * ph["solve"] / max(1, its[-1])
    halo = t_launch2 if N > 1 else 0.0
    pred = [setup + ITERS * (per_it + halo + 2 * tx * (N > 1)) / 1e3 for tx in (0.0, 3.0, 8.0)]
    print("%2d %9d %8d %10d %9.2f %10d %11.1f | %s" % (N, e - a, ghosts, s["local_contacts"], setup, its[-1], per_it,
          "   ".join("%.1f -> %.2f" % (p, 1e3 / p) for p in pred)), flush=True)
    st.op.close()
print("# (N = 1 through the staged driver: the fused single-GPU solve of the same system is ~8 % faster; the bench's N = 1 "
      "line uses the fused one)")
comm.close()
dist.destroy_process_group()
