#!/bin/bash
# tools/ab_env.sh "VAR=1" ... : interleaved steady-state bench runs of the SAME library with / without an environment switch
# (e.g. HLX_NO_BAKED=1), 3 rounds, one box.  Prints roofline.kernel_us.
ARGS=${AB_ARGS:---steps 400 --warmup 50 --no-cpu-baseline --no-extra-points --no-selfcheck --fused 0}
for rep in 1 2 3; do for v in "" "$@"; do
  env $v python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; g=lambda k: round(r[k]['kernel_us'],3) if k in r else None; print('[$v]', r['form'], round(r['kernel_us'],3), 'single_pass', g('single_pass'), 'terminal_obs_only', g('terminal_obs_only'), 'wall_us', round(d['ms_per_step']*1000,3), flush=True)"
done; done
