timeout -k 10 300 python -m pytest tests/test_backward_gpu.py -q -m gpu -x 2>&1 | tail -3
