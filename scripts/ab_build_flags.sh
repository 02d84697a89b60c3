#!/bin/bash
# A/B of tuning macros in place: every argument is one set of extra hipcc flags ("" = the defaults); the library is
# rebuilt for each (mundy_amd/build.py keys the build on the flags) and bench.py run REPS times.  Runs on the GPU box.
REPS=${REPS:-2}
for flags in "$@"; do
  export MHIP_EXTRA_HIPCC_FLAGS="$flags"
  python3 -m mundy_amd.build > /dev/null 2>&1 || { echo "build failed for [$flags]"; continue; }
  for rep in $(seq $REPS); do
    python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --relaxed-steps 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
o = d.get('k_body') or d.get('k_constraint')
print('[$flags]', 'ms/step %.2f' % d['ms_per_step'], 'iters', d['config']['bbpgd_iters_per_step'][0], 'relaxed %.2f' % d['relaxed_packing']['ms_per_step'], d['roofline']['kernel'], '%.4f' % d['roofline']['avg_launch_ms'], 'other %.4f' % o['avg_launch_ms'], d.get('cold_tier'))"
  done
done
unset MHIP_EXTRA_HIPCC_FLAGS
python3 -m mundy_amd.build > /dev/null 2>&1
