# round 6: the evidence set on the FINAL kernel sources, most important first (a round's GPU budget may end mid-script):
#   counter traffic of the headline workload (+ the per-kernel HBM file from the same two passes), the same for
#   normalize_input=True (k_conv_u8_img_norm: VERDICT r5 item 5d) and c5, the bench line with every secondary point, the whole GPU
#   suite, rocprofv3 kernel stats, one SQ pass.
#   bash tools/experiments/r06_final.sh <tag>
T=${1:-r06_z}
O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--steps 1 --warmup 1 --no_cpu_baseline --no_secondary --no_kernel_events"
traffic() {  # $1 = tag of the pass, $2 = output json, $3 = hbm json or "-", rest = bench arguments
  local tag=$1 out=$2 hbm=$3; shift 3
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${tag}fetch -o f -- python bench.py "$@" $Q > /dev/null 2> $O/${tag}fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${tag}write -o w -- python bench.py "$@" $Q > /dev/null 2> $O/${tag}write.err
  F=$(find $O/${tag}fetch -name "*counter_collection.csv" | head -1); W=$(find $O/${tag}write -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && [ -n "$W" ] && python tools/pmc_traffic.py $F $W > $O/$out 2> $O/${tag}_pmc_traffic.err
  [ "$hbm" != "-" ] && [ -n "$F" ] && [ -n "$W" ] && python tools/pmc_hbm.py $F $W $O/$hbm > /dev/null 2> $O/${tag}_pmc_hbm.err
}
traffic c2 ${T}_traffic.json ${T}_hbm_kernels.json --workload c2
cp $O/${T}_traffic.json profiles/${T}_traffic.json
traffic c2n ${T}_norm_traffic.json - --workload c2 --normalize_input
cp $O/${T}_norm_traffic.json profiles/${T}_norm_traffic.json
traffic c5 ${T}_c5_traffic.json ${T}_c5_hbm_kernels.json --workload c5
cp $O/${T}_c5_traffic.json profiles/${T}_c5_traffic.json
python bench.py --steps 20 --warmup 5 > $O/${T}_bench.json 2> $O/bench.err
tail -c 300 $O/${T}_bench.json
timeout 1500 python -m pytest tests/ -m gpu -q -x 2>&1 | tail -6 > $O/${T}_pytest_gpu.log
cat $O/${T}_pytest_gpu.log
B="python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > $O/${T}_bench_under_rocprof.json 2> $O/trace.err
K=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$K" ] && cp $K $O/${T}_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $O/sq -o p -- python bench.py $Q > /dev/null 2> $O/sq.err
S=$(find $O/sq -name "*counter_collection.csv" | head -1); [ -n "$S" ] && python tools/pmc_mfma.py $S $O/${T}_mfma_util.json > /dev/null 2> $O/pmc_mfma.err
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
ls $O
