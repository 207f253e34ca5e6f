O=gpurun_out/r05q; mkdir -p $O
L=$O/r05_q_conv3_img_vs_glds_z.log
export KBENCH_LAYERS=conv3 KBENCH_NS=1024,2048,4096,4096,32768,32768
echo "## conv3 forward: k_fwd_img (SF_FWD_IMG=1) vs k_fwd_glds_z<128,64> (SF_FWD_IMG=0)" > $L
for r in 1 2; do for v in 1 0; do echo "SF_FWD_IMG=$v" >> $L; SF_FWD_IMG=$v python tools/kbench.py fwd 2>/dev/null >> $L; done; done
cat $L
