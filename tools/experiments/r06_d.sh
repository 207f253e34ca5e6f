# conv3 forward (k_fwd_img): LDS image rows bottom-up + planes padded to 16 chunks (tree, SF_IMG_FLIP=1: conflict-free ds_read_b128
# groups) vs the round-5 layout (noflip = -DSF_IMG_FLIP=0); digests must be equal
#   bash tools/experiments/r06_d.sh
O=gpurun_out/r06d; mkdir -p $O
L=$O/r06_d_img_flip.log
export KBENCH_LAYERS=conv3 KBENCH_HASH=1
echo "## tree = SF_IMG_FLIP=1; noflip = -DSF_IMG_FLIP=0" > $L
for r in 1 2 3; do for v in tree noflip; do
  echo "lib=$v" >> $L; lib=$PWD/build/variants/libsf_hip_$v.so; [ $v == tree ] && lib=$PWD/sample_factory_amd/libsf_hip.so
  SF_HIP_LIB=$lib KBENCH_NS=4096,4096,32768,32768 timeout 300 python tools/kbench.py fwd 2>&1 | grep "^n=" >> $L
done; done
cat $L
timeout 900 python -m pytest tests/test_gpu_nn.py tests/test_gpu_headline_sizes.py -m gpu -q -x -k "conv3 or lds_image or headline or fwd" 2>&1 | tail -4 | tee -a $L
