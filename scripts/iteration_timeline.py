"""Per-iteration time budget of the fused solve from a rocprofv3 kernel trace (kernel_trace.csv): median duration of each
solver kernel over the tiered part of the solve and the median idle gap in front of it.
Usage: iteration_timeline.py <dir with *_kernel_trace.csv> [first_launch last_launch]"""
import csv, glob, os, sys
import numpy as np
root = sys.argv[1]
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (250, 750)
path = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("mhip::", "")))
rows.sort()
names = ("k_body", "k_constraint", "k_constraint_listed", "k_fold_partials", "k_finalize")
seen = {n: 0 for n in names}
dur = {n: [] for n in names}
gap = {n: [] for n in names}
prev_end = None
for s, e, n in rows:
    if n in seen:
        seen[n] += 1
        if lo <= seen[n] < hi:
            dur[n].append(e - s)
            if prev_end is not None:
                gap[n].append(s - prev_end)
    prev_end = e
tot = 0.0
for n in names:
    if dur[n]:
        d, g = np.median(dur[n]) / 1e3, np.median(gap[n]) / 1e3
        tot += d + g
        print("%-22s launches %5d  median %7.2f us  (p10 %7.2f, p90 %7.2f)  median gap before %5.2f us"
              % (n, seen[n], d, np.percentile(dur[n], 10) / 1e3, np.percentile(dur[n], 90) / 1e3, g))
print("sum of medians (kernels + gaps): %.2f us per iteration, launches %d..%d" % (tot, lo, hi))
