O=gpurun_out/r05k; mkdir -p $O
L=$O/r05_k_zl_ab.log
export KBENCH_LAYERS=conv2,fc KBENCH_NS=4096,4096,32768
echo "## k_fwd_glds vs k_fwd_glds_z (zero-VALU k-loop: SADDR-form DMA, fragment reads = VGPR + immediate), tools/kbench.py fwd_t column" > $L
for r in 1 2; do for v in 0 1; do echo "SF_GLDS_ZL=$v" >> $L; SF_GLDS_ZL=$v python tools/kbench.py fwd 2>/dev/null | grep -v "n=  4096 conv2.*\n" >> $L; done; done
cat $L
SF_GLDS_ZL=1 timeout 600 python -m pytest tests/test_gpu_nn.py -m gpu -q -x -k "glds or small_inference or fuzz or 64x64" 2>&1 | tail -4 | tee -a $L
