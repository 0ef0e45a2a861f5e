set -e
timeout -k 10 400 python -m pytest tests/test_wrappers_gpu.py -m gpu -x -q 2>&1 | tail -30
