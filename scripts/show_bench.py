"""one-line digest of a bench.py JSON log"""
import json
import sys

for path in sys.argv[1:]:
    line = [l for l in open(path).read().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    r = d.get("roofline") or {}
    print(path, "| ms/step", d["ms_per_step"], "| value", d["value"], "| iters", d["config"].get("bbpgd_iters_per_step"),
          "| k_constraint ms", r.get("avg_launch_ms"), "GB/s", r.get("achieved"), "| k_body", d.get("k_body"),
          "| stages", d.get("stage_ms"))
