O=gpurun_out/r05r; mkdir -p $O
L=$O/r05_r_conv2_persistent.log
export KBENCH_LAYERS=conv2 KBENCH_NS=4096,4096,4096,32768,32768
echo "## conv2 forward: k_fwd_glds_z (one tile per work-group) vs k_fwd_glds_zp (SF_GLDS_PERSIST=1: resident work-groups walk the tiles)" > $L
for r in 1 2; do for v in 0 1; do echo "SF_GLDS_PERSIST=$v" >> $L; SF_GLDS_PERSIST=$v python tools/kbench.py fwd 2>/dev/null | tail -4 >> $L; done; done
cat $L
