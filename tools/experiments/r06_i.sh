# conv3 forward (k_fwd_img): static priority for the work-group in the odd hardware wave slots (prio1 / prio3 = -DSF_IMG_PRIO=1 / 3)
# vs none (tree): do the two co-resident work-groups of a CU run their per-step overhead in phase?
O=gpurun_out/r06i; mkdir -p $O
L=$O/r06_i_img_prio.log
export KBENCH_LAYERS=conv3 KBENCH_HASH=1
echo "## tree = no priority; prio1 / prio3 = s_setprio 1 / 3 for the odd-slot work-group" > $L
for r in 1 2 3; do for v in tree prio1 prio3; do
  echo "lib=$v" >> $L; lib=$PWD/build/variants/libsf_hip_$v.so; [ $v == tree ] && lib=$PWD/sample_factory_amd/libsf_hip.so
  SF_HIP_LIB=$lib KBENCH_NS=4096,4096,32768,32768 timeout 300 python tools/kbench.py fwd 2>&1 | grep "^n=" | sed 's/.*| fwd_t/fwd_t/' >> $L
done; done
cat $L
