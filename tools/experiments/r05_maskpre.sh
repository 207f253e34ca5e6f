O=gpurun_out/r05o; mkdir -p $O
L=$O/r05_o_mask_prefetch.log
export KBENCH_LAYERS=conv2,fc KBENCH_NS=4096,4096,32768,32768
echo "## after: mask prefetch in the linear data gradient + __launch_bounds__(256, 2) on k_fwd_glds_z (no AGPR moves)" > $L
for r in 1 2; do python tools/kbench.py fwd dgrad 2>/dev/null | grep -v "^$" >> $L; done
cat $L
timeout 600 python -m pytest tests/test_gpu_nn.py -m gpu -q -x -k "dgrad or fuzz or large_grids or glds or small_inference" 2>&1 | tail -3 | tee -a $L
