scripts/profile_bench.sh r03 > gpurun_out/prof_r03.log 2>&1; tail -3 gpurun_out/prof_r03.log
python3 bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err
python3 bench.py --mixed --steps 2 --warmup 1 > gpurun_out/r03_mixed_bench_line.json 2> gpurun_out/r03_mixed.err
