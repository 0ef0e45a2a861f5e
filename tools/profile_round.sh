#!/bin/bash
# tools/profile_round.sh TAG : the evidence set of one round, written to gpurun_out/prof_TAG/ (run through gpurun).
#   1. bench.py (full line, CPU baseline included)            -> bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command    -> stats/
#   3. rocprofv3 --pmc passes (instruction mix, cycles, HBM)   -> pmc_*/   (counters only with --kernel-trace)
set -e -o pipefail
TAG=${1:-x}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 2000 --warmup 200 > $OUT/bench.json
cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extra-points --no-selfcheck --no-terminal-obs-point > $OUT/bench_under_rocprof.json
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --fused 0 --no-extra-points --no-selfcheck --no-terminal-obs-point > /dev/null
  f=$(find $OUT/pmc_$i -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py $f | tee -a $OUT/pmc_summary.txt
done
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
head -5 $OUT/kernel_stats.csv
