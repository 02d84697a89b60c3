#!/bin/bash
# A/B where _ab_old carries extra hipcc flags: $1 = the flags of _ab_old
for rep in 1 2 3; do
  for t in _ab_old .; do
    ( cd $t && if [ "$t" = "_ab_old" ]; then export MHIP_EXTRA_HIPCC_FLAGS="$1"; fi; python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --relaxed-steps 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$t', 'ms/step %.2f' % d['ms_per_step'], 'relaxed ms/step %.2f' % d['relaxed_packing']['ms_per_step'], 'k_body %.4f' % d['roofline']['avg_launch_ms'], 'k_constraint %.4f' % d['k_constraint']['avg_launch_ms'], d['cold_tier'])" )
  done
done
