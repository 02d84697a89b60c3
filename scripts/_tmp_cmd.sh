python3 scripts/stream_ceiling.py > /dev/null 2>&1
for flags in "" "-DMHIP_STREAM_GRID=1048576" "-DMHIP_STREAM_GRID=1048576 -DMHIP_STREAM_UNROLL=2" "-DMHIP_STREAM_GRID=1048576 -DMHIP_STREAM_UNROLL=8" "-DMHIP_STREAM_GRID=2048" ""; do
  export MHIP_EXTRA_HIPCC_FLAGS="$flags"
  python3 -m mundy_amd.build > /dev/null 2>&1
  echo "[$flags]"; python3 scripts/stream_ceiling.py 2>/dev/null | head -5
done
unset MHIP_EXTRA_HIPCC_FLAGS; python3 -m mundy_amd.build > /dev/null 2>&1
