set -e
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in 1 2 3; do for hw in 0 1; do
HLX_HALF_WAVES=$hw python bench.py --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('half=$hw us/step', round(d['ms_per_step']*1000,2), 'fused', round(d['fused_rollout']['ms_per_step']*1000,2))"
done; done
