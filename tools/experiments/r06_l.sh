# ReLU sign bits of conv3 for the fc data gradient (SF_RELU_BITS=1, default) vs the activation re-read (SF_RELU_BITS=0)
O=gpurun_out/r06l; mkdir -p $O
L=$O/r06_l_relu_bits.log
timeout 900 python -m pytest tests/test_gpu_nn.py -m gpu -q -x -k "relu_sign_bits or conv3 or lds_image" 2>&1 | tail -3 | tee $L
echo "## bench.py --steps 20 --warmup 5 --no_secondary --no_cpu_baseline, SF_RELU_BITS 0 / 1 alternating: ms per step, fc dgrad / conv3 fwd rows of the breakdown" >> $L
for r in 1 2; do for v in 0 1; do
  SF_RELU_BITS=$v timeout 600 python bench.py --steps 20 --warmup 5 --no_secondary --no_cpu_baseline > $O/b.json 2> $O/b.err
  python -c "
import json;d=json.load(open('$O/b.json'));print('SF_RELU_BITS=$v', d['ms_per_step'], d['value'], [(k['kernel'],k['ms_total'],k['tflops']) for k in d['network_kernels']['top'] if k['kernel'].startswith('dgrad:3136') or k['kernel'].startswith('fwd_t:64x9->64 n=32768')])" >> $L
done; done
cat $L
timeout 900 python -m pytest tests/test_gpu_headline_sizes.py tests/test_gpu_parity_c2_c5.py -m gpu -q -x 2>&1 | tail -3 | tee -a $L
