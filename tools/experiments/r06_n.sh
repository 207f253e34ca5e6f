# native separate actor / critic weights: the reference replays (native towers and torch path), model forward goldens, rollout
O=gpurun_out/r06n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_nn.py -m gpu -q -x -k "separate or sep_ or rnn or recurrent or stacked" 2>&1 | tail -25 > $O/r06_n_pytest_sep.log; cat $O/r06_n_pytest_sep.log
