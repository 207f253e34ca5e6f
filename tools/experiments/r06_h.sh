# conv3 data gradient (k_dgrad_pix_z): LDS destinations of the DMA instructions as SGPR base + immediate (tree, SF_PIX_LDSIMM=1)
# vs one loop-invariant destination per instruction parked in VGPR lanes (noimm = -DSF_PIX_LDSIMM=0); digests must be equal
O=gpurun_out/r06h; mkdir -p $O
L=$O/r06_h_pix_ldsimm.log
export KBENCH_NS=32768,32768,32768 KBENCH_LAYERS=conv3 KBENCH_HASH=1
echo "## tree = SF_PIX_LDSIMM=1; noimm = -DSF_PIX_LDSIMM=0" > $L
for r in 1 2 3; do for v in tree noimm; do
  echo "lib=$v" >> $L; lib=$PWD/build/variants/libsf_hip_$v.so; [ $v == tree ] && lib=$PWD/sample_factory_amd/libsf_hip.so
  SF_HIP_LIB=$lib timeout 300 python tools/kbench.py dgrad 2>&1 | grep "^n=" >> $L
done; done
cat $L
timeout 600 python -m pytest tests/test_gpu_nn.py -m gpu -q -x -k "dgrad or fuzz or large_grids" 2>&1 | tail -3 | tee -a $L
