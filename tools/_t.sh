set -e
HLX_LIBRARY=$PWD/hlynr_intercept_amd/libhlx_wpb4.so timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
tools/ab_bench.sh wpb1 wpb2 wpb4
