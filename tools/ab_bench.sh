#!/bin/bash
# tools/ab_bench.sh NAME... : interleaved bench runs of hlynr_intercept_amd/libhlx_NAME.so (3 rounds)
for rep in 1 2 3; do for v in "$@"; do
  HLX_LIBRARY=$PWD/hlynr_intercept_amd/libhlx_$v.so python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extra-points 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'us/step', round(d['ms_per_step']*1000,2))"
done; done
