set -e
timeout -k 10 400 python -m pytest tests/test_wrappers_gpu.py -m gpu -x -q 2>&1 | tail -3
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_w -- python3 tools/time_wrappers.py > gpurun_out/prof_w.log 2>&1
tail -3 gpurun_out/prof_w.log
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_w/**/*kernel_stats.csv',recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
