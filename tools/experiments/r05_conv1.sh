# conv1 forward: phase trace, LDS row pitch 84 vs 88; conv2 data gradient: byte-offset epilogue + zero-C first chunk vs HEAD
O=gpurun_out/r05s; mkdir -p $O
L=$O/r05_s_conv1_trace_quadrow.log
V=$PWD/build/variants
export KBENCH_NS=4096,4096,32768,32768
echo "## k_conv1_u8_bf16 phase trace (build -DSF_CONV1_TRACE=1)" > $L
SF_HIP_LIB=$V/libsf_hip_c1trace.so python tools/conv1_trace.py 32768 >> $L 2>&1
SF_HIP_LIB=$V/libsf_hip_c1trace.so python tools/conv1_trace.py 4096 >> $L 2>&1
echo "## conv1 forward, LDS row pitch: HEAD (88) vs 84" >> $L
for r in 1 2; do for v in head c1wp84; do echo "lib=$v" >> $L; SF_HIP_LIB=$V/libsf_hip_$v.so KBENCH_LAYERS=conv1 python tools/kbench.py fwd 2>/dev/null | tail -3 >> $L; done; done
echo "## conv2 data gradient n=32768: HEAD k_dgrad_quadrow_z vs byte-offset epilogue + zero-C first chunk (tree)" >> $L
for r in 1 2; do for v in head tree; do echo "lib=$v" >> $L; lib=$V/libsf_hip_$v.so; [ $v == tree ] && lib=$PWD/sample_factory_amd/libsf_hip.so; SF_HIP_LIB=$lib KBENCH_LAYERS=conv2 python tools/kbench.py dgrad 2>/dev/null | grep 32768 >> $L; done; done
cat $L
timeout 900 python -m pytest tests/test_gpu_nn.py -m gpu -q -x -k "dgrad or fuzz or large_grids or conv1" 2>&1 | tail -4 | tee -a $L
