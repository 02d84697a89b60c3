#!/bin/bash
# A/B runs of bench.py under different environment settings (on the GPU box).
# Usage: scripts/ab_env.sh "A=1 B=2" "A=3" ...     prints ms/step and the per-kernel averages per setting
for v in "$@"; do
  env $v python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; o=[d[k] for k in ('k_body','k_constraint') if k in d]
print('$v', 'ms/step', d['ms_per_step'], 'iters', d['config']['bbpgd_iters_per_step'][0], r['kernel'], r['avg_launch_ms'], *[(x['kernel'], x['avg_launch_ms']) for x in o])
" || exit 1
done
