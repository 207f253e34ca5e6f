# conv2 data gradient: DMA addresses of a chunk prepared one chunk ahead vs computed between the barrier and the DMA
O=gpurun_out/r05ac; mkdir -p $O
L=$O/r05_ac_quadrow_prep.log
V=$PWD/build/variants
export KBENCH_NS=4096,4096,32768,32768 KBENCH_LAYERS=conv2
echo "## k_dgrad_quadrow_z n=32768: -DSF_QUADROW_PREP=0 (noprep: DMA addresses computed between barrier and DMA) vs 1 (tree: one chunk ahead); dgrad = with activation read, dgrad_noact = the step's launch" > $L
for r in 1 2 3; do for v in noprep tree; do echo "lib=$v" >> $L; lib=$V/libsf_hip_$v.so; [ $v == tree ] && lib=$PWD/sample_factory_amd/libsf_hip.so; SF_HIP_LIB=$lib python tools/kbench.py dgrad dgrad_noact 2>/dev/null | grep 32768 | tail -1 >> $L; done; done
cat $L
timeout 900 python -m pytest tests/test_gpu_nn.py -m gpu -q -x -k "dgrad or fuzz or large_grids" 2>&1 | tail -4 | tee -a $L
