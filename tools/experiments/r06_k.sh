# fc data gradient (k_fwd_glds_z<128,128> + ReLU-mask epilogue): how much does the mask read cost?  dgrad with the activation
# read vs dgrad(no act read) = the upper bound of a sign-bit mask
O=gpurun_out/r06k; mkdir -p $O
L=$O/r06_k_fc_dgrad_mask.log
export KBENCH_LAYERS=fc,conv3 KBENCH_NS=32768,32768,32768
echo "## tools/kbench.py fwd dgrad dgrad_noact, fc and conv3, n = 32768" > $L
timeout 600 python tools/kbench.py fwd dgrad dgrad_noact 2>&1 | grep "^n=" >> $L
cat $L
