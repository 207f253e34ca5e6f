# round 6, call c: Tuple(Discrete, Box) tests + the RL-kernel / learner golden tests around them
O=gpurun_out/r06c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_rl_kernels.py tests/test_gpu_rollout_golden.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest_rl.log; cat $O/pytest_rl.log
timeout 1200 python -m pytest tests/test_gpu_nn.py -m gpu -x -q -k "tuple or learner_prepare or continuous or masked" 2>&1 | tail -8 > $O/pytest_nn_tuple.log; cat $O/pytest_nn_tuple.log
