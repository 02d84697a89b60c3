#!/bin/bash
# Runs the oracle's CPU tests against an AddressSanitizer + UndefinedBehaviorSanitizer build of the CPU restatement
# (sanitizers are available on the CPU build only; the GPU pool has no ASan).  Usage: tests/run_oracle_sanitizers.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
mkdir -p "$W/oracle"
cp "$ROOT"/oracle/*.py "$ROOT"/oracle/*.cpp "$ROOT"/oracle/*.hpp "$ROOT"/oracle/Makefile "$W/oracle/"
(cd "$W/oracle" && g++ -std=c++17 -fPIC -shared -fopenmp -O1 -g -ffp-contract=off -fsanitize=address,undefined \
   -fno-sanitize-recover=undefined -fno-omit-frame-pointer -o liboracle.so oracle_capi.cpp && cp liboracle.so liboracle_fast.so)
export LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0 MUNDY_ORACLE_PATH="$W"
cd "$ROOT"
python -m pytest tests/test_oracle_geom_kat.py tests/test_oracle_convex_kat.py tests/test_oracle_zmorton_hilbert_kat.py \
  tests/test_oracle_search.py tests/test_oracle_ellipsoid_kat.py tests/test_oracle_friction_ext.py \
  tests/test_oracle_aligned_rods.py tests/test_oracle_sums.py -x -q -p no:cacheprovider
