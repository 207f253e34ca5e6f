#!/bin/bash
# One round of judged evidence (run on the GPU box from the repo root):  bash tools/profile_round.sh <tag>
#   kernel trace + stats of the timed bench, the two TCC passes (HBM traffic), one SQ pass (MFMA busy).
# Counter passes never share a run with the trace domains other than --kernel-trace.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
B="python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_secondary"
B1="python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_secondary --no_kernel_events"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > $O/trace_bench.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $B1 > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $B1 > /dev/null 2> $O/write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $O/sq -o p -- $B1 > /dev/null 2> $O/sq.err
find $O -name "*.csv" | head -20
ls -la $O/*/ 2>/dev/null | head -40
# keep the merge-back small: the per-dispatch traces are large, the stats and counter tables are what is judged
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
