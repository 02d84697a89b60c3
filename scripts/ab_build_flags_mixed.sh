#!/bin/bash
# as ab_build_flags.sh, on the mixed line (configs[4])
REPS=${REPS:-2}
for flags in "$@"; do
  export MHIP_EXTRA_HIPCC_FLAGS="$flags"
  python3 -m mundy_amd.build > /dev/null 2>&1 || { echo "build failed for [$flags]"; continue; }
  for rep in $(seq $REPS); do
    python3 bench.py --mixed --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
o = d.get('k_body') or d.get('k_constraint')
print('[$flags]', 'ms/step %.2f' % d['ms_per_step'], 'iters', d['config']['bbpgd_iters_per_step'][0], d['roofline']['kernel'], '%.4f' % d['roofline']['avg_launch_ms'], 'frac %.3f' % d['roofline']['frac'], 'other %.4f' % o['avg_launch_ms'], 'narrow %.2f' % d['stage_ms']['narrowphase'])"
  done
done
unset MHIP_EXTRA_HIPCC_FLAGS
python3 -m mundy_amd.build > /dev/null 2>&1
