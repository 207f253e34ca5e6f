O=gpurun_out/r05n; mkdir -p $O
L=$O/r05_n_wgrad_zl_ab.log
export KBENCH_LAYERS=fc KBENCH_NS=4096,4096,32768,32768
echo "## fc weight gradient n=32768: k_wgrad_glds (SF_WGRAD_ZL=0) vs k_wgrad_glds_z (=1), tools/kbench.py wgrad" > $L
for r in 1 2; do for v in 0 1; do echo "SF_WGRAD_ZL=$v" >> $L; SF_WGRAD_ZL=$v python tools/kbench.py wgrad 2>/dev/null | grep 32768 >> $L; done; done
cat $L
timeout 900 python -m pytest tests/test_gpu_nn.py tests/test_gpu_parity_c2_c5.py tests/test_gpu_headline_sizes.py -m gpu -q -x 2>&1 | tail -4 | tee -a $L
python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
for k in d['network_kernels']['top'][:16]: print('    ', k['name'], k['kernel'], k['ms_total'], k.get('tflops'))" | tee -a $L
