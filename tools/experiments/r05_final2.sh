# final stamp of the round on the final kernel sources: counter traffic (both workloads), rocprofv3 kernel stats of the timed
# bench, one SQ pass (matrix-pipe busy), the bench line, the dispatch-sensitive tests
#   bash tools/experiments/r05_final2.sh <tag>
T=${1:-r05_f}
O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
bash tools/restamp.sh $T > $O/restamp.log 2>&1
cp $O/${T}_traffic.json profiles/${T}_traffic.json; cp $O/${T}_c5_traffic.json profiles/${T}_c5_traffic.json
B="python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_secondary"
B1="python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_secondary --no_kernel_events"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > $O/${T}_bench_under_rocprof.json 2> $O/trace.err
K=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$K" ] && cp $K $O/${T}_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $O/sq -o p -- $B1 > /dev/null 2> $O/sq.err
S=$(find $O/sq -name "*counter_collection.csv" | head -1); [ -n "$S" ] && python tools/pmc_mfma.py $S $O/${T}_mfma_util.json > /dev/null 2> $O/pmc_mfma.err
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
python bench.py --steps 20 > $O/${T}_bench.json 2> $O/bench.err
timeout 600 python -m pytest tests/test_gpu_headline_sizes.py tests/test_gpu_parity_c2_c5.py tests/test_gpu_nn.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest_dispatch.log
cat $O/pytest_dispatch.log; tail -c 300 $O/${T}_bench.json; ls $O
