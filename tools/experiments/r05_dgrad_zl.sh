O=gpurun_out/r05k; mkdir -p $O
L=$O/r05_k_dgrad_zl_ab.log
export KBENCH_LAYERS=conv2,conv3 KBENCH_NS=4096,4096,32768,32768
echo "## data gradients n=32768: k_dgrad_quadrow / k_dgrad_pix (SF_DGRAD_ZL=0) vs the _z forms (=1), tools/kbench.py dgrad" > $L
for r in 1 2; do for v in 0 1; do echo "SF_DGRAD_ZL=$v" >> $L; SF_DGRAD_ZL=$v python tools/kbench.py dgrad 2>/dev/null | grep 32768 >> $L; done; done
cat $L
SF_DGRAD_ZL=1 timeout 600 python -m pytest tests/test_gpu_nn.py -m gpu -q -x -k "dgrad or fuzz or large_grids" 2>&1 | tail -4 | tee -a $L
