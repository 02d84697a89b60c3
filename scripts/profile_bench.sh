#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py, then two separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Usage: scripts/profile_bench.sh <tag> [extra bench.py flags, e.g. --mixed]     -> gpurun_out/prof_<tag>/...
set -e
TAG=${1:-r01}
shift || true
EXTRA="$@"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="$PWD/bench.py"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -T -d "$OUT/trace" -- python3 "$BENCH" --steps 2 --warmup 1 --no-cpu-baseline --relaxed-steps 0 $EXTRA > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || { tail -20 "$OUT/trace.err"; exit 1; }
echo "trace pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -T --kernel-include-regex "k_constraint|k_body" -d "$OUT/pmc_fetch" -- python3 "$BENCH" --steps 1 --warmup 0 --no-cpu-baseline --relaxed-steps 0 $EXTRA > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.err" || { tail -20 "$OUT/pmc_fetch.err"; exit 1; }
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -T --kernel-include-regex "k_constraint|k_body" -d "$OUT/pmc_write" -- python3 "$BENCH" --steps 1 --warmup 0 --no-cpu-baseline --relaxed-steps 0 $EXTRA > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.err" || { tail -20 "$OUT/pmc_write.err"; exit 1; }
echo "write pass done"
# L2 hit rate and wave stall split of the two sweeps (diagnostics; TCC 2 slots + SQ 4 slots)
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -T --kernel-include-regex "k_constraint|k_body" -d "$OUT/pmc_l2" -- python3 "$BENCH" --steps 1 --warmup 0 --no-cpu-baseline --relaxed-steps 0 --max-iters 200 $EXTRA > "$OUT/bench_pmc_l2.json" 2> "$OUT/pmc_l2.err" || { tail -20 "$OUT/pmc_l2.err"; echo "l2 pass failed (diagnostic only)"; }
echo "l2 pass done"
cd - > /dev/null
python3 scripts/summarize_profile.py "$OUT" $EXTRA > "$OUT/summary.txt"
cat "$OUT/summary.txt"
# keep the merge small: drop the per-dispatch CSVs of the trace pass, keep stats + PMC summaries
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
