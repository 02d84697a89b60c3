"""The steady-state regime of the headline system (what a running simulation sees; bench.py's relaxed_packing.trajectory):
10^6 rods relaxed by two full steps, then K consecutive steps with the rebuild rule deciding about the neighbour list.
Prints ms per step, the stage split of rebuild / reuse steps, and -- when run under `rocprofv3 --kernel-trace` -- brackets
the trajectory with two marker launches (a cumsum: nothing else in this process runs one) so that
scripts/summarize_trajectory.py can cut the trace to it.   python scripts/trajectory.py [K] [--force-rebuild]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mundy_amd import ops, pipeline, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
force = "--force-rebuild" in sys.argv
no_tier = "--no-tier" in sys.argv
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
b = synth.spherocylinders(1_000_000, seed=1234)
st = pipeline.ContactStepper("spherocylinder", dev(b["center"]), dev(b["radius"]), dev(b["quat"]), dev(b["length"]),
                             dt=5e-3, viscosity=1e-3, search_buffer=0.1, search_kind=ops.SEARCH_AABB,
                             cfg=ops.PGDConfig(max_iters=10000, tol=1e-5))
st.reorder_bodies(cell_size=3.0, lo=[0.0, 0.0, 0.0])
if no_tier:
    st.tiering = 0
for _ in range(2):
    st.step(integrate=True, force_rebuild=True)
st.links.invalidate()
st.step(integrate=True)                       # (a rebuild: the list of the relaxed packing)
marker = torch.arange(5, device="cuda")
torch.cuda.synchronize()
marker.cumsum(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
stats = [st.step(integrate=True, force_rebuild=force) for _ in range(K)]
torch.cuda.synchronize()
el = time.perf_counter() - t0
marker.cumsum(0)
torch.cuda.synchronize()
print("trajectory%s%s: %d steps, %.3f ms per step, %d rebuilds, iterations %s" % (" --force-rebuild" if force else "", " --no-tier" if no_tier else "", K, 1e3 * el / K, sum(1 for s in stats if s.rebuilt), [s.num_iters for s in stats]))
# stage split (HIP events) of steps of either kind, outside the timed loop
acc = {True: {}, False: {}}
cnt = {True: 0, False: 0}
for _ in range(K):
    s = st.step(integrate=True, force_rebuild=force, timed=True)
    cnt[bool(s.rebuilt)] += 1
    for k, v in s.timings_ms.items():
        acc[bool(s.rebuilt)][k] = acc[bool(s.rebuilt)].get(k, 0.0) + v
for r in (True, False):
    if cnt[r]:
        print("%s steps (%d): %s" % ("rebuild" if r else "reuse", cnt[r], {k: round(v / cnt[r], 3) for k, v in acc[r].items()}))
