#!/bin/bash
# Stall breakdown of the network kernels (run on the GPU box from the repo root):  bash tools/stall_counters.sh <tag> [c2|c5]
#   two SQ counter passes of one instrumented C2 step (counter passes never share a run with trace domains other than
#   --kernel-trace); per-kernel ratios are printed by tools/stall_counters.py
set -u
TAG=${1:-rXX}
WL=${2:-c2}   # bench workload: c2 | c5
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
B1="python bench.py --workload $WL --steps 1 --warmup 1 --no_cpu_baseline --no_secondary --no_kernel_events"
rocprofv3 --list-avail > $O/avail.txt 2>&1 || rocprofv3 -L > $O/avail.txt 2>&1
grep -o "SQ_[A-Z0-9_]*" $O/avail.txt | sort -u > $O/avail_sq.txt
pick() { for c in "$@"; do grep -qx "$c" $O/avail_sq.txt && echo -n "$c "; done; }
P1=$(pick SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES)
P2=$(pick SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU)
echo "pass 1: $P1" > $O/stall_passes.txt; echo "pass 2: $P2" >> $O/stall_passes.txt
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $O/st1_$WL -o a -- $B1 > /dev/null 2> $O/st1.err
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $O/st2_$WL -o b -- $B1 > /dev/null 2> $O/st2.err
A=$(find $O/st1_$WL -name "*counter_collection.csv" | head -1); Bc=$(find $O/st2_$WL -name "*counter_collection.csv" | head -1)
python tools/stall_counters.py "$A" "$Bc" > $O/${TAG}_stall_counters_$WL.txt 2> $O/stall_counters.err
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
cat $O/stall_passes.txt; cat $O/${TAG}_stall_counters_$WL.txt | head -40
