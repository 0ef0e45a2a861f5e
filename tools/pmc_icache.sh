#!/bin/bash
# tools/pmc_icache.sh TAG : instruction-fetch counters of the step kernel (separate --pmc passes, --kernel-trace only)
set -e -o pipefail
TAG=${1:-x}
OUT=$PWD/gpurun_out/pmc_icache_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQC_TC_REQ" \
           "SQ_INST_LEVEL_VMEM SQ_ACCUM_PREV_HIRES SQ_WAIT_INST_LDS SQ_INSTS_BRANCH" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --fused 0 --no-extra-points --no-selfcheck --forms contract > /dev/null 2> $OUT/err_$i.txt || { echo "set $i failed: $set"; tail -3 $OUT/err_$i.txt; continue; }
  f=$(find $OUT/pmc_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py $f | tee -a $OUT/summary.txt
done
