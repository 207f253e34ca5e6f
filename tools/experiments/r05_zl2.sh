O=gpurun_out/r05k; mkdir -p $O
L=$O/r05_k_zl128_ab.log
export KBENCH_LAYERS=fc KBENCH_NS=4096,4096,32768,32768
echo "## fc forward n=32768 on 128x128 tiles: k_fwd_glds (SF_GLDS_ZL=1) vs k_fwd_glds_z (SF_GLDS_ZL=2)" > $L
for r in 1 2; do for v in 1 2; do echo "SF_GLDS_ZL=$v" >> $L; SF_GLDS_ZL=$v python tools/kbench.py fwd 2>/dev/null | grep 32768 >> $L; done; done
cat $L
