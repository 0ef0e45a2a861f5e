mkdir -p gpurun_out/r3z
for spec in "65536 400 50 2000" "131072 300 40 1000" "262144 200 30 500" "1048576 80 10 100" "4194304 40 5 40"; do
  set -- $spec
  python bench.py --envs-per-gpu $1 --steps $2 --warmup $3 --preroll $4 --no-cpu-baseline --no-extra-points --no-selfcheck 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
f=d.get('fused_rollout') or {}
print('$1', 'contract', round(r['kernel_us'],2), round(r['frac'],3), 'no_outputs', round(r['single_pass']['kernel_us'],2), round(r['single_pass']['frac'],3), 'term_obs', round(r['terminal_obs_only']['kernel_us'],2), round(r['terminal_obs_only']['frac'],3), 'fused_us_per_step', round(f.get('device_us_per_step',0),2), 'sched', d['config']['episode_pool']['load_schedule'], 'value %.3g' % d['value'])
"
done
