"""Per-kernel time of the trajectory window of a `rocprofv3 --kernel-trace` run of scripts/trajectory.py (the window
between the two marker launches whose kernel name contains `cumsum` / `scan`): python scripts/summarize_trajectory.py <dir> <K>"""
import csv, glob, os, sys
from collections import defaultdict
out, K = sys.argv[1], int(sys.argv[2])
f = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "single_scan_kernel" in r["Kernel_Name"]]
a, b = marks[0], marks[-1]
win = rows[a + 1:b]
t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["Start_Timestamp"])
tot = defaultdict(lambda: [0, 0])
for r in win:
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("mhip::", "").split("(")[0]
    k = k.split("<")[0] if not k.startswith("at::") else k[:60]
    tot[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot[k][1] += 1
busy = sum(v[0] for v in tot.values())
print("# trajectory window: %d steps, wall %.3f ms per step, kernels busy %.3f ms per step (%d launches per step)"
      % (K, (t1 - t0) / 1e6 / K, busy / 1e6 / K, len(win) // K))
print("%-44s %10s %9s %8s" % ("kernel", "ms/step", "launches", "share"))
for k, (ns, n) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%-44s %10.4f %9.1f %7.1f%%" % (k[:44], ns / 1e6 / K, n / K, 100.0 * ns / busy))
