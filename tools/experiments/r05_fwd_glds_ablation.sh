O=gpurun_out/r05j; mkdir -p $O
L=$O/r05_j_fwd_glds_ablation.log
export KBENCH_LAYERS=conv2 KBENCH_NS=4096,4096,32768,32768
echo "## k_fwd_glds<128,64> conv2 forward (tools/kbench.py fwd_t column; n = 4096 twice first as clock warm-up), SF_GLDS_ABLATE bits: 1 no DMA in the k-loop, 2 no wait/barrier per chunk, 4 no epilogue stores, 8 one k-chunk pair per tile" > $L
for r in 1 2; do
for b in 0 1 2 3 4 7 8 15; do
  if [ $b = 0 ]; then unset SF_HIP_LIB; else export SF_HIP_LIB=$PWD/build/variants/libsf_hip_abl$b.so; fi
  echo "ablate=$b" >> $L; python tools/kbench.py fwd 2> $O/err_$b.txt | grep "32768 conv2" >> $L; tail -2 $O/err_$b.txt | grep -i "error\|assert" >> $L
done; done
cat $L
