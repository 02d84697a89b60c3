"""Writes the bench's 10^6-rod input in Hilbert order and runs the C++ DISTRIBUTED host program (tests/cpp/rod_dist_app,
mech::DistributedSpherocylinderStepper over the RCCL transport) on it with one rank: ghost plan, partitioned operator
and the domain-decomposed BBPGD loop with its all-gather, no Python / torch / MPI in the timed process."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mundy_amd import distributed as D, synth
from test_adapter_cpp import _build_dist_app
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
b = synth.spherocylinders(n, seed=1234)
order = D.hilbert_order(b["center"], 0.0, b["box"], level=7)
c, q, r, ln = (np.ascontiguousarray(b[k][order]) for k in ("center", "quat", "radius", "length"))
mt, mr = synth.dry_mobility(0.5 * ln + r)
d = tempfile.mkdtemp()
path = os.path.join(d, "rods.bin")
with open(path, "wb") as f:
    f.write(np.uint64(n).tobytes())
    for a in (c, q, r, ln, mt, mr):
        f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
p = subprocess.run([_build_dist_app(), path, "3", "0", "1", d], capture_output=True, text=True)
print(p.stdout, p.stderr[-2000:])
