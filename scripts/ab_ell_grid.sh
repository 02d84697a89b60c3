for flags in "" "-DMHIP_ELL_GRID=1024" "-DMHIP_ELL_GRID=1536"; do
  export MHIP_EXTRA_HIPCC_FLAGS="$flags"
  python3 -m mundy_amd.build > /dev/null 2>&1
  echo -n "[$flags] "; python3 scripts/time_ellipsoids.py 250000 2>/dev/null | tail -1
done
unset MHIP_EXTRA_HIPCC_FLAGS; python3 -m mundy_amd.build > /dev/null 2>&1
