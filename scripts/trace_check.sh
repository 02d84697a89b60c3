#!/bin/bash
# rocprofv3 marker trace of a small bench run with MHIP_TRACE=1: lists the roctx ranges the library emits
export TMPDIR=/tmp MHIP_TRACE=1
OUT=$PWD/gpurun_out/prof_trace
rm -rf "$OUT"; mkdir -p "$OUT"
B="$PWD/bench.py"
cd /tmp
rocprofv3 --marker-trace --output-format csv -d "$OUT" -- python3 "$B" --bodies 50000 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/run.log" 2> "$OUT/err.log" || { tail -5 "$OUT/err.log"; exit 1; }
f=$(find "$OUT" -name "*marker_api_trace.csv" | head -1)
echo "marker file: $f"
python3 - "$f" <<'PY'
import csv, sys, collections
c = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    c[r.get("Function", r.get("Message", "?"))] += 1
for k, v in c.most_common():
    print("%5d  %s" % (v, k))
PY
