O=gpurun_out/r05e; mkdir -p $O
L=$O/r05_e_conv2_64x64_tiles.log
export KBENCH_NS=4096,8192 KBENCH_LAYERS=conv2
echo "## conv2 forward, 128x64 tiles (default)" > $L; python tools/kbench.py fwd >> $L 2>&1
echo "## conv2 forward, 64x64 tiles (SF_GLDS_FORCE64)" >> $L; SF_GLDS_FORCE64=100000 python tools/kbench.py fwd >> $L 2>&1
cat $L
