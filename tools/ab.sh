#!/bin/bash
# same-box A/B of an environment switch:  bash tools/ab.sh SF_LINEAR_SMALL_N 0 1 [repeats]
VAR=$1; A=$2; B=$3; N=${4:-2}
for i in $(seq $N); do for v in $A $B; do
  env $VAR=$v timeout 200 python bench.py --no_cpu_baseline --no_secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done; done
