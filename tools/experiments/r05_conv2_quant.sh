O=gpurun_out/r05e; mkdir -p $O
L=$O/r05_e_conv2_quantisation.log
export KBENCH_LAYERS=conv2
echo "## conv2 forward k_fwd_glds<128,64>: tiles per CU = n*81/128/256; 3640 -> 9.0, 4045 -> 10.0, 4096 -> 10.125, 4449 -> 11.0" > $L
KBENCH_NS=3640,4045,4096,4449,4854 python tools/kbench.py fwd >> $L 2>&1
cat $L
