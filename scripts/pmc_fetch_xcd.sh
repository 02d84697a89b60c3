#!/bin/bash
# FETCH_SIZE per launch of the two solver sweeps with and without the XCD-contiguous tile mapping (bench.py --xcd-tile T).
# Usage (on the GPU box): scripts/pmc_fetch_xcd.sh  -> gpurun_out/pmc_xcd_{0,32}/...
export TMPDIR=/tmp
R=$PWD
for T in 0 32; do
  OUT=$R/gpurun_out/pmc_xcd_$T
  rm -rf "$OUT"; mkdir -p "$OUT"
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -T --kernel-include-regex "k_constraint|k_body" -d "$OUT" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --relaxed-steps 0 --max-iters 40 --xcd-tile $T > "$OUT/bench.json" 2> "$OUT/err.txt" < /dev/null)
  python3 - "$OUT" "$T" <<'PY'
import csv, glob, sys, collections
out, T = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"].split("<")[0].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v = [x for x in v if x > 0.5 * max(v)]
    print("xcd_tile=%s %-14s launches %3d  FETCH_SIZE %.1f MB/launch (x2 gfx950 correction applied)" % (T, k, len(v), 2 * 1024 * sum(v) / len(v) / 1e6))
PY
done
