#!/bin/bash
# A/B of k_body's lanes-per-body x chains-per-lane layouts and of the occupancy hints of the two sweeps (run on the GPU
# box; A/B builds only -- the shipped library takes its defaults from convex.hip).
out=gpurun_out/ab_body; mkdir -p $out
for flags in "" "-DMHIP_CONSTRAINT_WAVES=8" "-DMHIP_BODY_WAVES=6 -DMHIP_CONSTRAINT_WAVES=8"; do
  tag=$(echo "$flags" | tr -d ' =-' ); tag=${tag:-default}
  MHIP_EXTRA_HIPCC_FLAGS="$flags" python mundy_amd/build.py --force > $out/build_$tag.log 2>&1 || { echo "build failed $tag"; continue; }
  for lay in "4 4" "4 2" "8 2" "8 4" "2 4" "4 8"; do
    set -- $lay
    MHIP_LANES_PER_BODY=$1 MHIP_BODY_UNROLL=$2 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/b.json 2> $out/b.err
    python - "$tag" $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/ab_body/b.json").read().strip().splitlines()[-1])
    kb = d["roofline"] if d["roofline"]["kernel"].startswith("k_body") else d["k_body"]
    kc = d["roofline"] if d["roofline"]["kernel"].startswith("k_constraint") else d["k_constraint"]
    print("%-40s G=%s U=%s  k_body %.4f ms  k_constraint %.4f ms  step %.2f ms  iters %s" % (
        sys.argv[1], sys.argv[2], sys.argv[3], kb["avg_launch_ms"], kc["avg_launch_ms"], d["ms_per_step"],
        d["config"]["bbpgd_iters_per_step"][0]), flush=True)
except Exception as e:
    print(sys.argv[1:], "failed", e, open("gpurun_out/ab_body/b.err").read()[-500:])
PY
  done
done 2>&1 | tee $out/summary.txt
python mundy_amd/build.py --force > $out/build_final.log 2>&1
