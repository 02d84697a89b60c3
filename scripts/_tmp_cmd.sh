export MHIP_EXTRA_HIPCC_FLAGS="-DMHIP_TIER_DEBUG"
python3 -m mundy_amd.build > /dev/null 2>&1
python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --relaxed-steps 0 2>&1 | grep -E "tier_update|compact active" | head -40
unset MHIP_EXTRA_HIPCC_FLAGS; python3 -m mundy_amd.build > /dev/null 2>&1
