#!/bin/bash
# the headline workload at 4 x and 16 x its size on one GPU (same density, same tolerance)
echo "# \`python3 bench.py --bodies N --steps 1 --warmup 0 --no-cpu-baseline --relaxed-steps 0\` on one MI355X: the headline workload at 4 x and 16 x its size (same density, same tolerance)"
for n in 4000000 16000000; do
  timeout -k 10 500 python3 bench.py --bodies $n --steps 1 --warmup 0 --no-cpu-baseline --relaxed-steps 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
r, o = d['roofline'], d.get('k_constraint') or d.get('k_body')
print('bodies %d  contacts %d  BBPGD iterations %s converged %s  ms/step %.1f  %s %.4f ms (frac %.3f)  %s %.4f ms (frac %.3f)  stages %s  cold tier %s' % (d['config']['bodies_total'], d['config']['contacts_total'], d['config']['bbpgd_iters_per_step'], d['config']['converged'], d['ms_per_step'], r['kernel'].split('<')[0], r['avg_launch_ms'], r['frac'], o['kernel'].split('<')[0], o['avg_launch_ms'], o['frac'], d['stage_ms'], d.get('cold_tier')))"
done
