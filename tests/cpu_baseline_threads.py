"""CPU baseline (the oracle's OpenMP BBPGD) at different thread counts on this box: the bench reports the count it
used; this shows whether another count would have been a fairer (faster) baseline."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from mundy_amd import ops, pipeline, synth
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "omp max", oracle.num_threads())
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError as e:
    print("cgroup cpu.max unavailable:", e)
b = synth.spherocylinders(1_000_000)
st = pipeline.ContactStepper("spherocylinder", dev(b["center"]), dev(b["radius"]), dev(b["quat"]), dev(b["length"]),
                             search_buffer=0.1, cfg=ops.PGDConfig(max_iters=1, tol=1e-5), rod_kinematics=False)
st.step(integrate=False)
pairs = st.links.pairs.cpu().numpy()
c = {k: v.cpu().numpy() for k, v in st.contacts.items() if v is not None}
mt, mr = st.mob_trans.cpu().numpy(), st.mob_rot.cpu().numpy()
for th in (8, 16, 32, 64, 128):
    oracle.set_num_threads(th)
    k = 6
    t = time.perf_counter()
    oracle.solve_cqpp_contact(pairs, c["normal"], c["ra"], c["rb"], mt, mr, 5e-3, c["sep"], np.zeros(len(pairs)),
                              max_iters=k, tol=1e-5, threads=True, fast=True)
    dt = time.perf_counter() - t
    print("threads %3d: %.3f s per BBPGD iteration" % (th, dt / (k + 1)), flush=True)
