# final stamp on the final kernel sources, most important first (the round's GPU budget may end mid-script): counter traffic
# of the headline workload, the bench line, the dispatch-sensitive tests, then c5 traffic, rocprofv3 kernel stats, one SQ pass
#   bash tools/experiments/r05_final3.sh <tag>
T=${1:-r05_g}
O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
Q="--steps 1 --warmup 1 --no_cpu_baseline --no_secondary --no_kernel_events"
traffic() {  # $1 = workload, $2 = output json
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$1fetch -o f -- python bench.py --workload $1 $Q > /dev/null 2> $O/$1fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/$1write -o w -- python bench.py --workload $1 $Q > /dev/null 2> $O/$1write.err
  F=$(find $O/$1fetch -name "*counter_collection.csv" | head -1); W=$(find $O/$1write -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && [ -n "$W" ] && python tools/pmc_traffic.py $F $W > $O/$2 2> $O/$1_pmc_traffic.err
}
traffic c2 ${T}_traffic.json
cp $O/${T}_traffic.json profiles/${T}_traffic.json
python bench.py --steps 20 > $O/${T}_bench.json 2> $O/bench.err
tail -c 200 $O/${T}_bench.json
timeout 300 python -m pytest tests/test_gpu_headline_sizes.py tests/test_gpu_parity_c2_c5.py tests/test_gpu_nn.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest_dispatch.log
cat $O/pytest_dispatch.log
traffic c5 ${T}_c5_traffic.json
B="python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > $O/${T}_bench_under_rocprof.json 2> $O/trace.err
K=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$K" ] && cp $K $O/${T}_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $O/sq -o p -- python bench.py $Q > /dev/null 2> $O/sq.err
S=$(find $O/sq -name "*counter_collection.csv" | head -1); [ -n "$S" ] && python tools/pmc_mfma.py $S $O/${T}_mfma_util.json > /dev/null 2> $O/pmc_mfma.err
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
ls $O
