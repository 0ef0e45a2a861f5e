for rep in 1 2; do
python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --fused 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product us/step', round(d['ms_per_step']*1000,2))"
HLX_LIBRARY=$PWD/hlynr_intercept_amd/libhlx_probe.so python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --fused 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('role-split probe us/step', round(d['ms_per_step']*1000,2))"
done
