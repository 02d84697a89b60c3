#!/bin/bash
# A/B of the lockstep kernels' unit of work (starts of a pair per unit: 9 = the whole pair, 3, 1), at pair counts from
# an eighth of the mixed bench's E-E class to four times it.  Runs on the GPU box; rebuilds the library per setting.
for spu in 9 3 1 0; do
  export MHIP_EXTRA_HIPCC_FLAGS="-DMHIP_LOCKSTEP_STARTS_PER_UNIT=$spu"
  python3 -m mundy_amd.build > /dev/null 2>&1 || { echo "build failed for $spu"; continue; }
  for n in 30000 100000 250000 1000000; do
    echo -n "starts per unit $spu: "; python3 scripts/time_ellipsoids.py $n 2>/dev/null | tail -1
  done
done
unset MHIP_EXTRA_HIPCC_FLAGS
python3 -m mundy_amd.build > /dev/null 2>&1
