#!/bin/bash
# tools/ab_bench.sh NAME... : interleaved steady-state bench runs of hlynr_intercept_amd/libhlx_NAME.so on ONE box
# (run-to-run spread between boxes is +-4 %: an A/B only means something inside one gpurun call).  3 rounds each;
# prints the event-clocked mean launch duration (roofline.kernel_us) of bench.py's default workload.
ARGS=${AB_ARGS:---steps 2000 --warmup 200 --no-cpu-baseline --no-extra-points --no-selfcheck --fused 0}
for rep in 1 2 3; do for v in "$@"; do
  HLX_LIBRARY=$PWD/hlynr_intercept_amd/libhlx_$v.so python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['roofline'].get('with_terminal_observations'); print('$v', 'kernel_us', round(d['roofline']['kernel_us'],3), 'wall_us', round(d['ms_per_step']*1000,3), 'with_terminal_obs_us', round(t['kernel_us'],3) if t else None, flush=True)"
done; done
