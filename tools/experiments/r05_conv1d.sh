# conv1 forward phase trace with one / two resident work-groups per CU, both store forms
O=gpurun_out/r05x; mkdir -p $O
L=$O/r05_x_conv1_trace_occupancy.log
V=$PWD/build/variants
echo "## k_conv1_u8_bf16 / _w phase trace (build -DSF_CONV1_TRACE=1), n = 32768" > $L
for w in 0 1; do for g in 1 2; do echo "SF_CONV1_WIDE=$w SF_CONV1_WGS=$g" >> $L; SF_CONV1_WIDE=$w SF_CONV1_WGS=$g SF_HIP_LIB=$V/libsf_hip_c1trace.so python tools/conv1_trace.py 32768 2>/dev/null >> $L; done; done
cat $L
