# conv2 data gradient (k_dgrad_quadrow_z, n = 32768, no activation read = the step's launch): where does the non-MFMA time go?
O=gpurun_out/r05q2; mkdir -p $O
L=$O/r05_aa_quadrow_ablation.log
V=$PWD/build/variants
export KBENCH_NS=4096,4096,32768,32768 KBENCH_LAYERS=conv2
echo "## k_dgrad_quadrow_z ablation (-DSF_GLDS_ABLATE bits: 16 no DMA, 32 no output stores, 64 no wait/barrier per chunk, 128 no MFMAs; 48 = 16+32), tools/kbench.py dgrad_noact" > $L
for r in 1 2; do for v in tree qabl16 qabl32 qabl48 qabl64 qabl128; do echo "lib=$v" >> $L; lib=$V/libsf_hip_$v.so; [ $v == tree ] && lib=$PWD/sample_factory_amd/libsf_hip.so; SF_HIP_LIB=$lib python tools/kbench.py dgrad_noact 2>/dev/null | grep 32768 | tail -1 >> $L; done; done
cat $L
