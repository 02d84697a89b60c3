"""Writes the bench's 10^6-rod input to a file and runs the C++ host program (tests/cpp/rod_step_app, built on
include/mundy_hip/stepper.hpp) on it: the same hot path with no Python / torch in the timed process."""
import os, subprocess, sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from mundy_amd import synth
from test_adapter_cpp import _build_rod_app
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
b = synth.spherocylinders(n, seed=1234)
mt, mr = synth.dry_mobility(0.5 * b["length"] + b["radius"])
path = "/tmp/rods_%d.bin" % n
with open(path, "wb") as f:
    f.write(np.uint64(n).tobytes())
    for a in (b["center"], b["quat"], b["radius"], b["length"], mt, mr):
        f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
print(subprocess.run([_build_rod_app(), path, "3", "3.0"], capture_output=True, text=True).stdout)
