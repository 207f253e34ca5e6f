#!/bin/bash
# c3 (minibatches of 8192): the fc weight gradient on the LDS-DMA kernel instead of the register-staged one (row threshold)
cd /root/repo
for rep in 1 2; do
for v in "" "SF_WGRAD_GLDS_MIN=8192"; do
  env $v python bench.py --workload c3 --steps 12 --warmup 3 --no_cpu_baseline --no_secondary --no_kernel_events 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('c3 [$v] rep=$rep', d['ms_per_step'], d['value'])
except Exception as e: print('c3 [$v] failed', e)"
done; done
