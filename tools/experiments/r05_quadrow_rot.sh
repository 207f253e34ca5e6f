# conv2 data gradient: half of the work-groups walk their group rows rotated by one (out of lock-step) vs all in step
O=gpurun_out/r05ab; mkdir -p $O
L=$O/r05_ab_quadrow_rotation.log
V=$PWD/build/variants
export KBENCH_NS=4096,4096,32768,32768 KBENCH_LAYERS=conv2
echo "## k_dgrad_quadrow_z n=32768: -DSF_QUADROW_ROT=0 (norot) vs 1 (tree); dgrad = with activation read, dgrad_noact = the step's launch" > $L
for r in 1 2 3; do for v in norot tree; do echo "lib=$v" >> $L; lib=$V/libsf_hip_$v.so; [ $v == tree ] && lib=$PWD/sample_factory_amd/libsf_hip.so; SF_HIP_LIB=$lib python tools/kbench.py dgrad dgrad_noact 2>/dev/null | grep 32768 | tail -1 >> $L; done; done
cat $L
timeout 900 python -m pytest tests/test_gpu_nn.py -m gpu -q -x -k "dgrad or fuzz or large_grids" 2>&1 | tail -4 | tee -a $L
