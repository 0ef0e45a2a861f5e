timeout -k 10 600 python -m pytest tests/test_hrl_gpu.py -m gpu -x -q 2>&1 | tail -25
