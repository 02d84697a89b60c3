for r in 1 2 3; do python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --relaxed-steps 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
o = d.get('k_body') or d.get('k_constraint')
print('ms/step %.2f' % d['ms_per_step'], 'iters', d['config']['bbpgd_iters_per_step'][0], 'relaxed %.2f' % d['relaxed_packing']['ms_per_step'], d['roofline']['kernel'], '%.4f' % d['roofline']['avg_launch_ms'], 'other %.4f' % o['avg_launch_ms'], d['stage_ms'])"; done
python -m pytest tests/test_gpu_convex.py -m gpu -x -q 2>&1 | tail -2
