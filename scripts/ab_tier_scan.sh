#!/bin/bash
# On the GPU box: workgroups that scan the cold tail underneath the hot sweep (convex.hip, kTierScanBlocks), swept as a
# compile-time constant; prints the bench's step time and the two sweeps' launch times.  Restores the default build.
for b in "${@:-256 512 1024 2048}"; do
  export MHIP_EXTRA_HIPCC_FLAGS="-DMHIP_TIER_SCAN_BLOCKS=$b"
  python3 -c "from mundy_amd import build; build.build()" > /dev/null 2>&1 || { echo "build failed for $b"; exit 1; }
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --relaxed-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; k=d.get('k_constraint') or d.get('k_body')
print('scan blocks $b: step %.2f ms, %s %.4f ms, %s %.4f ms, iterations %s' % (d['ms_per_step'], r['kernel'][:12], r['avg_launch_ms'], k['kernel'][:12], k['avg_launch_ms'], d['config']['bbpgd_iters_per_step']))"
done
unset MHIP_EXTRA_HIPCC_FLAGS
python3 -c "from mundy_amd import build; build.build()" > /dev/null 2>&1
