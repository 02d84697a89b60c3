"""bench.py's output contract on a small workload: one JSON line on stdout with the driver's keys, the roofline and
cpu_baseline objects, and sane values."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bodies", "30000", "--steps", "2", "--warmup",
                        "1", "--cpu-iters", "2"] + extra, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_default_line_has_the_contract_keys():
    d = _run([])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "timesteps/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-3)
    assert all(d["config"]["converged"]) and d["config"]["contacts_per_gpu"] > 100_000
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0.0 < r["frac"] < 1.0 and r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=1e-3)
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] >= 1 and 0.0 < c["value"] < d["value"]


def test_friction_extension_line_is_labelled():
    d = _run(["--friction", "0.3", "--no-cpu-baseline"])
    assert "EXTENSION" in d["config"]["workload"] and "build extension" in d["metric"]
    assert d["cpu_baseline"] is None


def test_mixed_line_names_configs4():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mixed", "--bodies", "9000", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert "configs[4]" in d["config"]["workload"] and "mixed" in d["metric"]
    assert d["config"]["converged"] == [True] and d["stage_ms"]["narrowphase"] > 0 and d["roofline"]["bound"] == "hbm"
