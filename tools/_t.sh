set -e
timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
tools/ab_bench.sh lds1 lds0
