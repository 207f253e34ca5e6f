O=gpurun_out/r05i; mkdir -p $O
L=$O/r05_i_fastcell_ab.log; : > $L
V=$PWD/build/variants/libsf_hip_fastcell.so
for i in 1 2 3; do
  for lib in default fastcell; do
    if [ $lib = fastcell ]; then export SF_HIP_LIB=$V; else unset SF_HIP_LIB; fi
    python bench.py --workload c5 --steps 12 --warmup 3 --no_cpu_baseline --no_secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'])" >> $L
  done
done
export SF_HIP_LIB=$V
timeout 600 python -m pytest tests/test_gpu_parity_c2_c5.py tests/test_gpu_rl_kernels.py -m gpu -q -k "config5 or seq or rnn or lstm or gru" 2>&1 | tail -5 >> $L
cat $L
