"""BUILD EXTENSION (friction, parity unpinned): iterations of the frictional solve on the bench input -- the raw packing
and the packing relaxed by two frictionless steps (what bench.py calls `relaxed_packing`).  Usage: [bodies] [mu] [method]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from mundy_amd import ops, pipeline, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
mu = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
method = sys.argv[3] if len(sys.argv) > 3 else None
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
b = synth.spherocylinders(n, seed=1234)
cfg = ops.PGDConfig(max_iters=40000, tol=1e-5)
kw = {} if method is None else {"friction_method": method}


def stepper(center, quat, friction):
    st = pipeline.ContactStepper("spherocylinder", center, dev(b["radius"]), quat, dev(b["length"]), search_buffer=0.1,
                                 cfg=cfg, friction=friction, **(kw if friction is not None else {}))
    st.reorder_bodies(cell_size=3.0, lo=[0.0, 0.0, 0.0])
    return st


free = stepper(dev(b["center"]), dev(b["quat"]), None)
for label, relax in (("raw", 0), ("relaxed", 2)):
    for _ in range(relax):
        free.step(integrate=True, force_rebuild=True)
    fr = stepper(free.center.clone(), free.quat.clone(), mu)
    fr.radius, fr.length = free.radius.clone(), free.length.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = fr.step(integrate=False, force_rebuild=True)
    torch.cuda.synchronize()
    print("%s packing, mu %.2f%s: %d contacts, %d iterations, converged %s, residual %.3g, %.1f ms" % (
        label, mu, "" if method is None else " (" + method + ")", s.num_contacts, s.num_iters, s.converged, s.residual,
        1e3 * (time.perf_counter() - t0)), flush=True)
