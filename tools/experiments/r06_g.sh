# conv3 forward (k_fwd_img): a work-group's last step with ONE live fragment runs the one-fragment k-loop (tree, SF_IMG_HALFSTEP=1)
# vs multiplying a duplicate fragment (nohalf = -DSF_IMG_HALFSTEP=0); digests must be equal
O=gpurun_out/r06g; mkdir -p $O
L=$O/r06_g_img_halfstep.log
export KBENCH_LAYERS=conv3 KBENCH_HASH=1
echo "## tree = SF_IMG_HALFSTEP=1; nohalf = -DSF_IMG_HALFSTEP=0" > $L
for r in 1 2 3; do for v in tree nohalf; do
  echo "lib=$v" >> $L; lib=$PWD/build/variants/libsf_hip_$v.so; [ $v == tree ] && lib=$PWD/sample_factory_amd/libsf_hip.so
  SF_HIP_LIB=$lib KBENCH_NS=4096,4096,4096,32768 timeout 300 python tools/kbench.py fwd 2>&1 | grep "^n=" | sed 's/.*| fwd_t/fwd_t/' >> $L
done; done
cat $L
timeout 900 python -m pytest tests/test_gpu_nn.py tests/test_gpu_headline_sizes.py -m gpu -q -x -k "conv3 or lds_image or headline or fwd" 2>&1 | tail -3 | tee -a $L
