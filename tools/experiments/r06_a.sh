set -x
bash tools/experiments/r06_x9_gate.sh
python bench.py > gpurun_out/r06a/r06_a_bench.json 2> gpurun_out/r06a/r06_a_bench.err; tail -c 600 gpurun_out/r06a/r06_a_bench.err
python -c "
import json;d=json.load(open('gpurun_out/r06a/r06_a_bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'])
for s in d.get('secondary',[]): print(s.get('name'), s.get('value'), s.get('ms_per_step'))
"
