# conv1 forward: whole-line output stores through an LDS staging tile (k_conv1_u8_bf16_w) vs the dword-store form (r05_t: the DPP-transposition form of the same experiment)
O=gpurun_out/r05v; mkdir -p $O
L=$O/r05_v_conv1_staged_stores.log
V=$PWD/build/variants
export KBENCH_NS=4096,4096,32768,32768 KBENCH_LAYERS=conv1
echo "## k_conv1_u8_bf16_w phase trace (build -DSF_CONV1_TRACE=1)" > $L
SF_HIP_LIB=$V/libsf_hip_c1trace.so python tools/conv1_trace.py 32768 2>/dev/null >> $L
SF_HIP_LIB=$V/libsf_hip_c1trace.so python tools/conv1_trace.py 4096 2>/dev/null >> $L
echo "## conv1 forward: SF_CONV1_WIDE=0 (dword stores) vs 1 (whole-line stores), same library, LDS pitch 84" >> $L
for r in 1 2; do for v in 0 1; do echo "SF_CONV1_WIDE=$v" >> $L; SF_CONV1_WIDE=$v python tools/kbench.py fwd 2>/dev/null | tail -3 >> $L; done; done
cat $L
timeout 900 python -m pytest tests/test_gpu_nn.py tests/test_gpu_headline_sizes.py -m gpu -q -x -k "conv1 or relu_mask or headline or u8" 2>&1 | tail -6 | tee -a $L
