"""The randomised parity sweeps (tests/stress_parity*.py: rods, spheres, mixed shapes of varied size / density /
buffer / dt / shape through the whole step, every stage against the oracle, cold tier forced on) inside the GPU test run.
Their output is also committed under profiles/ (r0N_stress_parity.txt)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,cases,verdict", [("stress_parity.py", 16, "STRESS PASS"),
                                                  ("stress_parity_spheres.py", 12, "STRESS PASS"),
                                                  ("stress_parity_mixed.py", 10, "STRESS PASS (mixed)")])
def test_randomised_sweep(script, cases, verdict):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", script), str(cases)], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    print(p.stdout[-4000:], p.stderr[-1500:])
    assert p.stdout.count("\nok  ") + p.stdout.startswith("ok  ") == cases, p.stdout[-3000:]
    assert verdict in p.stdout and "FAIL" not in p.stdout
