#!/bin/bash
# On the GPU box: the lockstep minimiser's schedule (line-search tests every L-th round, loop head every H-th round;
# ellipsoid_lockstep.hpp) swept as compile-time constants, arguments "L:H ...".  Prints E-E pairs/s and the mixed
# narrow-phase time; restores the default build at the end.
for cfg in "${@:-1:1 2:4 3:6}"; do
  L=${cfg%%:*}; H=${cfg##*:}
  export MHIP_EXTRA_HIPCC_FLAGS="-DMHIP_LOCKSTEP_LS_EVERY=$L -DMHIP_LOCKSTEP_HEAD_EVERY=$H"
  python3 -c "from mundy_amd import build; build.build()" > /dev/null 2>&1 || { echo "build failed for $cfg"; exit 1; }
  echo "== line-search tests every $L, head every $H"
  python3 scripts/time_ellipsoids.py 400000 2>/dev/null | tail -1
  python3 scripts/time_mixed.py 2>/dev/null | tail -1 | sed "s/.*narrow phase/mixed 300k narrow phase/;s/;.*//"
done
unset MHIP_EXTRA_HIPCC_FLAGS
python3 -c "from mundy_amd import build; build.build()" > /dev/null 2>&1
