# round 6, call e: the whole GPU suite on the current tree
O=gpurun_out/r06e; mkdir -p $O
timeout 2400 python -m pytest tests/ -m gpu -q -x 2>&1 | tail -15 > $O/r06_e_pytest_gpu.log; cat $O/r06_e_pytest_gpu.log
