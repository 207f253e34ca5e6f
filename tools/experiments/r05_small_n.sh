set -u
O=gpurun_out/r05b; mkdir -p $O
export KBENCH_NS=512,1024,2048 KBENCH_LAYERS=conv2,conv3,fc
L=$O/r05_b_kbench_small_n.log
echo "## baseline dispatch (n = 512 / 1024 / 2048: per-split inference launches of host-env runs)" > $L
python tools/kbench.py fwd >> $L 2>&1
echo "## SF_GLDS_MIN_TILES=256 (k_fwd_glds<128,64> from one tile per CU)" >> $L
SF_GLDS_MIN_TILES=256 python tools/kbench.py fwd >> $L 2>&1
echo "## SF_GLDS_SMALL64=256 (k_fwd_glds<64,64> for N = 64 layers) + SF_GLDS_SPLIT64=32 (fc: 64x64 tiles split along K)" >> $L
SF_GLDS_SMALL64=256 SF_GLDS_SPLIT64=32 python tools/kbench.py fwd >> $L 2>&1
echo "## SF_GLDS_MIN_TILES=256 SF_GLDS_SPLIT64=32 SF_FWD_IMG=0" >> $L
SF_GLDS_MIN_TILES=256 SF_GLDS_SPLIT64=32 SF_FWD_IMG=0 python tools/kbench.py fwd >> $L 2>&1
cat $L
