#!/bin/bash
# k_fwd_glds2 (one recurrent inference step, 2048 x 2048 x (64 + 512)): tile shapes 128x64 (production) / 64x64 / 128x128
set -u
O=gpurun_out/r06_w; mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
  for v in base g2_64 g2_128; do
    L=""; [ $v != base ] && L="SF_HIP_LIB=$PWD/build/variants/libsf_hip_$v.so"
    env $L python bench.py --workload c5 --steps 16 --warmup 4 --no_cpu_baseline --no_secondary --no_kernel_events 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('c5 $v rep=$rep', d['ms_per_step'], d['value'])" | tee -a $O/r06_w_ab.log
  done
done
for v in base g2_64 g2_128; do
  L=""; [ $v != base ] && L="SF_HIP_LIB=$PWD/build/variants/libsf_hip_$v.so"
  env $L rocprofv3 --kernel-trace --stats -d $O/prof_$v -o c5 --output-format csv -- python bench.py --workload c5 --steps 8 --warmup 2 --no_cpu_baseline --no_secondary --no_kernel_events > /dev/null 2>&1
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "$v: $(grep k_fwd_glds2 $f | head -1 | cut -c1-200)" | tee -a $O/r06_w_ab.log
  rm -rf $O/prof_$v
done
