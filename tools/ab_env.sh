#!/bin/bash
# tools/ab_env.sh "VAR=1" ... : interleaved steady-state bench runs of the SAME library with / without an environment switch
# (e.g. HLX_NO_BAKED=1), 3 rounds, one box.  Prints roofline.kernel_us.
ARGS=${AB_ARGS:---steps 2000 --warmup 200 --no-cpu-baseline --no-extra-points --no-selfcheck --fused 0}
for rep in 1 2 3; do for v in "" "$@"; do
  env $v python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', 'kernel_us', round(d['roofline']['kernel_us'],3), 'wall_us', round(d['ms_per_step']*1000,3), flush=True)"
done; done
