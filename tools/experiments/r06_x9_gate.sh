# exact-product 3 x 3 bf16 split (9 MFMAs of 16x16x32 bf16 per K = 32, f32 accumulate) against TODAY's f32 kernels
# (k_fwd_glds_z<128,128> fc forward, k_fwd_glds_zt<128,64> conv2 forward) on the same box, with the shader clock inside the kernel.
#   bash tools/experiments/r06_x9_gate.sh
O=gpurun_out/r06a; mkdir -p $O
L=$O/r06_a_x9_gate.log
export KBENCH_NS=32768,32768,32768 KBENCH_LAYERS=fc,conv2
echo "## same-box f32 reference (tools/kbench.py fwd; fwd_t = the LDS-DMA kernels the step runs)" > $L
timeout 300 python tools/kbench.py fwd 2>&1 | grep "^n=" >> $L
fc=$(grep " fc " $L | sed -n 's/.*fwd_t *\([0-9.]*\)us.*/\1/p' | sort -n | head -1)
c2=$(grep " conv2 " $L | sed -n 's/.*fwd_t *\([0-9.]*\)us.*/\1/p' | sort -n | head -1)
fcms=$(python -c "print($fc/1000)"); c2ms=$(python -c "print($c2/1000)")
echo "## gemm_x9_dma against fc $fcms ms, conv2 $c2ms ms (best of the three kbench rounds)" >> $L
for r in 1 2; do timeout 300 tools/ubench/gemm_x9_dma $fcms $c2ms >> $L 2>&1; done
cat $L
