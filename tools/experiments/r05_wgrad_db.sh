# fc weight gradient (k_wgrad_glds_z<128, 128>): fragments two k-steps ahead through a ring of four register slots
# (-DSF_WGRAD_FRAG_DB=1, wdb) vs the shipped single-buffered reads (tree); digests must be equal
#   bash tools/experiments/r05_wgrad_db.sh
O=gpurun_out/r05ai; mkdir -p $O
L=$O/r05_ai_wgrad_db.log
export KBENCH_NS=32768,32768,32768 KBENCH_LAYERS=fc KBENCH_HASH=1
echo "## tree = shipped; wdb = -DSF_WGRAD_FRAG_DB=1" > $L
for r in 1 2 3; do for v in tree wdb; do
  echo "lib=$v" >> $L; lib=$PWD/build/variants/libsf_hip_$v.so; [ $v == tree ] && lib=$PWD/sample_factory_amd/libsf_hip.so
  SF_HIP_LIB=$lib timeout 300 python tools/kbench.py wgrad 2>&1 | grep "^n=" >> $L
done; done
cat $L
