for rep in 1 2; do for v in g13 cur; do
L=$PWD/hlynr_intercept_amd/libhlx_$v.so; [ $v = cur ] && L=$PWD/hlynr_intercept_amd/libhlx.so
for n in 1048576 4194304; do
HLX_LIBRARY=$L python bench.py --envs-per-gpu $n --steps 100 --warmup 30 --no-cpu-baseline --fused 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v n=$n', 'us/step', round(d['ms_per_step']*1000,2), 'frac', round(d['roofline']['frac'],3))"
done; done; done
