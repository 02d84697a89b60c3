"""DIAGNOSTIC (library built with -DMHIP_EXP_COUNT_MM): of the body sweep's waves behind a snapshot, how many have a lane
whose body got a newly active entry since (the whole wave then runs the per-body chains: two more dependent round trips)"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mundy_amd import capi, ops, pipeline, synth
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
b = synth.spherocylinders(1_000_000, seed=1234)
st = pipeline.ContactStepper("spherocylinder", dev(b["center"]), dev(b["radius"]), dev(b["quat"]), dev(b["length"]),
                             search_buffer=0.1, cfg=ops.PGDConfig(max_iters=10000, tol=1e-5))
st.reorder_bodies(cell_size=3.0, lo=[0.0, 0.0, 0.0])
lib = C.CDLL(capi.load()._name)
out = (C.c_ulonglong * 4)()
lib.mhip_debug_counters(out)
s = st.step(integrate=False, force_rebuild=True)
lib.mhip_debug_counters(out)
waves, with_mm, lanes, entries = out[2], out[0], out[1], out[3]
print("iterations %d; flat-sweep waves %d, of which %d (%.1f %%) had a lane with newly active entries; such lanes %d (%.2f per such wave), entries %d"
      % (s.num_iters, waves, with_mm, 100.0 * with_mm / max(1, waves), lanes, lanes / max(1, with_mm), entries))
