#!/bin/bash
# tools/ab_bench.sh NAME... : interleaved steady-state bench runs of hlynr_intercept_amd/libhlx_NAME.so on ONE box
# (run-to-run spread between boxes is +-4 %: an A/B only means something inside one gpurun call).  3 rounds each;
# prints the event-clocked launch duration (median of the windows) of each form of bench.py's default workload:
# contract (the headline: terminal observations + info planes + done list), single_pass, terminal_obs_only.
ARGS=${AB_ARGS:---steps 400 --warmup 50 --no-cpu-baseline --no-extra-points --no-selfcheck --fused 0}
for rep in 1 2 3; do for v in "$@"; do
  HLX_LIBRARY=$PWD/hlynr_intercept_amd/libhlx_$v.so python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; g=lambda k: round(r[k]['kernel_us'],3) if k in r else None; print('$v', r['form'], round(r['kernel_us'],3), 'single_pass', g('single_pass'), 'terminal_obs_only', g('terminal_obs_only'), 'wall_us', round(d['ms_per_step']*1000,3), flush=True)"
done; done
