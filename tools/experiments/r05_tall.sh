O=gpurun_out/r05r; mkdir -p $O
L=$O/r05_r_conv2_tall_tiles.log
export KBENCH_LAYERS=conv2 KBENCH_NS=4096,4096,8192,32768,32768
echo "## conv2 forward: 128x64 tiles (default) vs 256x64 tiles, waves 4x1 (SF_GLDS_TALL=1)" > $L
for r in 1 2; do for v in 0 1; do echo "SF_GLDS_TALL=$v" >> $L; SF_GLDS_TALL=$v python tools/kbench.py fwd 2>/dev/null | tail -3 >> $L; done; done
cat $L
