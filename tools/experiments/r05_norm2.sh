O=gpurun_out/r05d; mkdir -p $O
timeout 800 python -m pytest tests/test_gpu_nn.py tests/test_gpu_parity_c2_c5.py -m gpu -q -k "loader_fused or normalize_input" > $O/pytest_norm.log 2>&1; tail -40 $O/pytest_norm.log
