O=gpurun_out/r06t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_runner.py -m gpu -q 2>&1 | tail -5 | tee $O/r06_t_pytest_runner.log
