# conv1 forward: how many work-groups per CU are resident, and what the persistent grid size does
O=gpurun_out/r05w; mkdir -p $O
L=$O/r05_w_conv1_occupancy.log
export KBENCH_NS=4096,4096,32768,32768 KBENCH_LAYERS=conv1
echo "## conv1 forward: persistent grid = SF_CONV1_WGS work-groups per CU, dword-store (WIDE=0) and whole-line (WIDE=1) forms" > $L
SF_DEBUG_OCC=1 KBENCH_NS=4096 python tools/kbench.py fwd 2>&1 | grep occupancy | head -1 >> $L
for w in 0 1; do for g in 1 2 3 4; do echo "SF_CONV1_WIDE=$w SF_CONV1_WGS=$g" >> $L; SF_CONV1_WIDE=$w SF_CONV1_WGS=$g python tools/kbench.py fwd 2>/dev/null | tail -3 >> $L; done; done
cat $L
