# the whole GPU suite, failures listed
O=gpurun_out/r06s; mkdir -p $O
timeout 2400 python -m pytest tests/ -m gpu -q 2>&1 | tail -80 > $O/r06_s_pytest_gpu.log; cp $O/r06_s_pytest_gpu.log $O/r06_z_pytest_gpu.log; tail -5 $O/r06_s_pytest_gpu.log
