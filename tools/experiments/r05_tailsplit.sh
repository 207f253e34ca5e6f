# conv2 forward: the last, thin round of 128-row tiles run as 64-row tiles (k_fwd_glds_zt) vs the plain launch
O=gpurun_out/r05ad; mkdir -p $O
L=$O/r05_ad_fwd_tail_split.log
export KBENCH_NS=4096,4096,4096,32768,32768 KBENCH_LAYERS=conv2
echo "## conv2 forward (sf_conv_fwd_t): SF_GLDS_TAILSPLIT=0 vs 1, tools/kbench.py fwd (second figure of a line = the LDS-DMA path)" > $L
for r in 1 2 3; do for v in 0 1; do echo "SF_GLDS_TAILSPLIT=$v" >> $L; SF_GLDS_TAILSPLIT=$v python tools/kbench.py fwd 2>/dev/null | tail -3 | cut -c1-140 >> $L; done; done
cat $L
timeout 900 python -m pytest tests/test_gpu_nn.py tests/test_gpu_headline_sizes.py -m gpu -q -x -k "glds or fwd_t or headline or fuzz or large_grids" 2>&1 | tail -4 | tee -a $L
