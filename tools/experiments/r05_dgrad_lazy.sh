# data gradients: counted vmcnt waits across the store phases (tree) vs vmcnt(0) (-DSF_DGRAD_LAZY=0)
O=gpurun_out/r05z; mkdir -p $O
L=$O/r05_z_dgrad_lazy_wait.log
V=$PWD/build/variants
export KBENCH_NS=4096,4096,32768,32768 KBENCH_LAYERS=conv2,conv3
echo "## conv2 / conv3 data gradients n=32768: vmcnt(0) at every chunk (nolazy) vs counted waits behind the output stores (tree)" > $L
for r in 1 2 3; do for v in nolazy tree; do echo "lib=$v" >> $L; lib=$V/libsf_hip_$v.so; [ $v == tree ] && lib=$PWD/sample_factory_amd/libsf_hip.so; SF_HIP_LIB=$lib python tools/kbench.py dgrad 2>/dev/null | grep 32768 >> $L; done; done
cat $L
timeout 900 python -m pytest tests/test_gpu_nn.py -m gpu -q -x -k "dgrad or fuzz or large_grids" 2>&1 | tail -4 | tee -a $L
