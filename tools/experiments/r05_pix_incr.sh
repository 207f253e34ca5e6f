# conv3 data gradient (k_dgrad_pix_z): the next chunk's (filter row, filter column, channel chunk) stepped (tree, SF_PIX_INCR=1)
# vs decoded from the chunk index by three run-time divisions (piold = -DSF_PIX_INCR=0); digests must be equal
#   bash tools/experiments/r05_pix_incr.sh
O=gpurun_out/r05aj; mkdir -p $O
L=$O/r05_aj_pix_incr.log
export KBENCH_NS=32768,32768 KBENCH_LAYERS=conv3,conv2 KBENCH_HASH=1
echo "## tree = SF_PIX_INCR=1; piold = -DSF_PIX_INCR=0" > $L
for r in 1 2 3; do for v in tree piold; do
  echo "lib=$v" >> $L; lib=$PWD/build/variants/libsf_hip_$v.so; [ $v == tree ] && lib=$PWD/sample_factory_amd/libsf_hip.so
  SF_HIP_LIB=$lib timeout 300 python tools/kbench.py dgrad dgrad_noact 2>&1 | grep "^n=" >> $L
done; done
cat $L
timeout 600 python -m pytest tests/test_gpu_nn.py -m gpu -q -x -k "dgrad or fuzz or large_grids" 2>&1 | tail -4 | tee -a $L
