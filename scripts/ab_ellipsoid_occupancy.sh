#!/bin/bash
# On the GPU box: rebuild the library with different occupancy targets for the ellipsoid kernels and time them.
for w in "$@"; do
  MHIP_EXTRA_HIPCC_FLAGS="-DELL_WAVES=$w" python3 mundy_amd/build.py --force > /dev/null 2>&1 || { echo "build failed for $w"; exit 1; }
  echo -n "ELL_WAVES=$w  "; python3 scripts/time_ellipsoids.py 1000000 2>/dev/null | tail -1
done
