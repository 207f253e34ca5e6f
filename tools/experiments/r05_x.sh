timeout 600 python -m pytest tests/test_gpu_parity_c2_c5.py -m gpu -q -x -k "cnn84_32k" 2>&1 | tail -30
