#!/bin/bash
# Diagnostic PMC passes over the two sweeps and the stream-ceiling copy of ONE bench.py run each (few counters per pass:
# the TCC has two slots).  What the L2's memory side saw: requests, requests in flight (-> mean latency), credit stalls.
# Usage (GPU box): scripts/pmc_diag.sh <tag> [bench flags]  -> gpurun_out/pmcdiag_<tag>/summary.txt
set -e
TAG=${1:-r04}
shift || true
EXTRA="$@"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcdiag_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="$PWD/bench.py"
cd /tmp
i=0
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" "TCC_CYCLE_sum TCC_BUSY_sum" "TCC_REQ_sum TCC_STREAMING_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -T --kernel-include-regex "k_constraint|k_body|k_copy" -d "$OUT/p$i" -- python3 "$BENCH" --steps 1 --warmup 0 --no-cpu-baseline --relaxed-steps 0 --max-iters 400 $EXTRA > "$OUT/p$i.json" 2> "$OUT/p$i.err" || { echo "pass $i ($SET) failed"; tail -3 "$OUT/p$i.err"; }
  echo "pass $i done: $SET"
done
cd - > /dev/null
python3 - "$OUT" <<'PY' > "$OUT/summary.txt"
import csv, glob, os, sys, statistics
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0]
        acc[k][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print("# mean per EFFECTIVE launch (duration >= half the median of that kernel in that pass); durations under the counters in us")
for k in sorted(acc):
    print("\n## " + k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        med = statistics.median(d for _, d in v)
        e = [(x, d) for x, d in v if d >= 0.5 * med]
        print("%-44s %14.4g   launches %5d  mean duration %.1f us" % (c, sum(x for x, _ in e) / len(e), len(e), sum(d for _, d in e) / len(e) / 1e3))
PY
cat "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +5M -delete
