O=gpurun_out/r05n; mkdir -p $O
L=$O/r05_n_dgrad_pix_lite_ab.log
export KBENCH_LAYERS=conv3 KBENCH_NS=4096,4096,32768,32768
echo "## conv3 data gradient n=32768: k_dgrad_pix (SF_DGRAD_ZL=1) vs k_dgrad_pix_z with SADDR-form DMA only (=3)" > $L
for r in 1 2 3; do for v in 1 3; do echo "SF_DGRAD_ZL=$v" >> $L; SF_DGRAD_ZL=$v python tools/kbench.py dgrad 2>/dev/null | grep 32768 >> $L; done; done
cat $L
SF_DGRAD_ZL=3 timeout 600 python -m pytest tests/test_gpu_nn.py -m gpu -q -x -k "dgrad or fuzz or large_grids" 2>&1 | tail -3 | tee -a $L
