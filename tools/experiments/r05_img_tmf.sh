O=gpurun_out/r05q; mkdir -p $O
L=$O/r05_q_fwd_img_tmf.log
export KBENCH_LAYERS=conv3 KBENCH_NS=4096,4096,32768,32768
echo "## conv3 forward k_fwd_img: fragments per wave and block step (TMF) 2 (default) / 3 / 4" > $L
for r in 1 2; do for t in 2 3 4; do
  if [ $t = 2 ]; then unset SF_HIP_LIB; else export SF_HIP_LIB=$PWD/build/variants/libsf_hip_tmf$t.so; fi
  echo "TMF=$t" >> $L; python tools/kbench.py fwd 2>/dev/null | tail -3 >> $L
done; done
cat $L
