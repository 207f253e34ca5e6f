#!/bin/bash
# launch programs for the rollout step: parity tests, then c5 / c2 with programs off / on, alternating on one box
set -u
O=gpurun_out/r06_q; mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_launch_programs.py -x -q -m gpu 2>&1 | tail -25 > $O/r06_q_pytest_launch_programs.log
cat $O/r06_q_pytest_launch_programs.log
for rep in 1 2; do
  for on in 0 1; do
    SF_LAUNCH_PROGRAMS=$on python bench.py --workload c5 --steps 16 --warmup 4 --no_cpu_baseline --no_secondary 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('c5 programs=$on rep=$rep', d['ms_per_step'], d['value'], d['roofline']['frac'])" | tee -a $O/r06_q_ab.log
  done
done
for on in 0 1; do
  SF_LAUNCH_PROGRAMS=$on python bench.py --steps 6 --warmup 3 --no_cpu_baseline --no_secondary 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('c2 programs=$on', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel'])" | tee -a $O/r06_q_ab.log
done
