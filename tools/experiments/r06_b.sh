# round 6, call b: the new data-parallel tests (bucket order, per-epoch moment exchange), host facts, the bench line with the new
# secondary points and the cpu_baseline leg
O=gpurun_out/r06b; mkdir -p $O
(free -g; nproc; cat /sys/fs/cgroup/memory.max 2>/dev/null; lscpu | grep -i "numa\|model name\|socket") > $O/host.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_dp.py -m gpu -x -q -s 2>&1 | tail -15 > $O/pytest_dp.log; cat $O/pytest_dp.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06_b_bench.json 2> $O/r06_b_bench.err; tail -c 400 $O/r06_b_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06b/r06_b_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for s in d.get('secondary',[]): print(s.get('workload'), s.get('value'), s.get('ms_per_step'), s.get('steps'), s.get('wall_s'), s.get('error'), (s.get('roofline') or {}).get('kernel'), (s.get('roofline') or {}).get('frac'))
print(json.dumps(d['cpu_baseline'])[:1200])
PY
