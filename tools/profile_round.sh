#!/bin/bash
# tools/profile_round.sh TAG : the evidence set of one round, written to gpurun_out/prof_TAG/ (run through gpurun).
#   1. bench.py (full line, CPU baseline included) and the driver's standing command (--steps 20 --warmup 5)   -> bench.json, bench_k20.json
#   2. rocprofv3 --kernel-trace --stats of ONE form of the step per run (contract = the headline, then single_pass)  -> stats_*/
#   3. rocprofv3 --pmc passes on the contract form (instruction mix, cycles, HBM)                               -> pmc_*/   (counters only with --kernel-trace)
set -e -o pipefail
TAG=${1:-x}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# BENCH_EXTRA: extra bench.py arguments for every run (e.g. "--config 3" = v2dr physics, "--envs-per-gpu 4194304");
# SKIP_BENCH=1: no plain bench lines (profiling passes only); FORMS: which forms get a --stats pass
BENCH_EXTRA=${BENCH_EXTRA:-}
if [ -z "$SKIP_BENCH" ]; then
python3 bench.py --steps 2000 --warmup 200 $BENCH_EXTRA > $OUT/bench.json
cut -c1-1500 $OUT/bench.json
python3 bench.py --steps 20 --warmup 5 $BENCH_EXTRA > $OUT/bench_k20.json
fi
QUIET="--no-cpu-baseline --no-extra-points --no-selfcheck --fused 0 $BENCH_EXTRA"
STEPS=${PROF_STEPS:-1000}
for form in ${FORMS:-contract single_pass}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$form -- python3 bench.py --steps $STEPS --warmup 100 $QUIET --forms $form > $OUT/bench_under_rocprof_$form.json
  find $OUT/stats_$form -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_$form.csv \;
  head -4 $OUT/kernel_stats_$form.csv
done
if [ -n "$SKIP_PMC" ]; then exit 0; fi      # SKIP_PMC=1: no counter passes
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 bench.py --steps 40 --warmup 5 --preroll 200 $QUIET --forms contract > /dev/null
  f=$(find $OUT/pmc_$i -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py $f "${PMC_KERNEL:-hlx_env_kernel<608u, 0, false, false}" | tee -a $OUT/pmc_summary.txt
done
