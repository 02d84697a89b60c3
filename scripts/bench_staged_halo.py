"""Per-iteration cost of the distributed (staged) solve with the velocity halo through the IPC-mapped inboxes against the
transport's send / recv, W ranks SHARING the one GPU of a test box (gloo: the transport is the host-callback one, so this
prices the inbox path against a host-staged halo, not against RCCL over xGMI -- that needs the N > 1 node).
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29791 \
      scripts/bench_staged_halo.py [bodies_total]"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, ".")
from mundy_amd import distributed as D, ops, synth  # noqa: E402

n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
b = synth.spherocylinders(n_total, seed=1234)
order = D.hilbert_order(b["center"], 0.0, b["box"], level=6)
starts = D.partition_ranges(n_total, world)
a, e = int(starts[rank]), int(starts[rank + 1])
g = {k: b[k][order] for k in ("center", "quat", "radius", "length")}
cfg = ops.PGDConfig(max_iters=10000, tol=1e-5)
for label, kw in (("halo through the inboxes, records through the mailbox", dict()),
                  ("halo through the transport (host-staged here), records through the mailbox", dict(halo_ipc=False)),
                  ("halo and records through the transport (host-staged here)", dict(halo_ipc=False, mailbox=False))):
    comm = D.Comm(**kw)
    st = D.DistributedContactStepper(dev(g["center"][a:e]), dev(g["quat"][a:e]), dev(g["radius"][a:e]),
                                     dev(g["length"][a:e]), a, comm=comm, search_buffer=0.1, cfg=cfg, poll_every=32,
                                     domain=(0.0, b["box"]), curve_level=5)
    st.step(integrate=False)
    ts = []
    for _ in range(3):
        dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = st.step(integrate=False)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    if rank == 0:
        it = out["num_iters"]
        print("%d ranks on one GPU, %d rods in all, %d ghosts on rank 0 | %s: inboxes active %s | %.1f ms per step, %d "
              "iterations, %.1f us per iteration (neighbour list and contacts included)"
              % (world, n_total, out["ghosts"], label, comm.halo_ipc_active(), 1e3 * float(np.median(ts)), it,
                 1e6 * float(np.median(ts)) / max(1, it)), flush=True)
    st.op.close()
    comm.close()
dist.destroy_process_group()
