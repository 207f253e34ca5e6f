# native stacked recurrent layers: the recurrent tests (1 and 2 layers), the rollout goldens, configs[4] parity and the c5 line
O=gpurun_out/r06m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_nn.py tests/test_gpu_rollout_golden.py tests/test_gpu_parity_c2_c5.py tests/test_gpu_dp.py -m gpu -q -x -k "rnn or recurrent or lstm or gru or stacked or c5 or sequence or rollout or replicas" 2>&1 | tail -15 > $O/r06_m_pytest_rnn.log; cat $O/r06_m_pytest_rnn.log
python bench.py --workload c5 --steps 16 --warmup 3 --no_cpu_baseline --no_secondary 2> $O/c5.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', d['value'], d['ms_per_step'])"
