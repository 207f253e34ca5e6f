#!/bin/bash
# Counter traffic of BOTH bench workloads on the current kernel sources, nothing else (run on the GPU box from the repo
# root):  bash tools/restamp.sh <tag>     -> gpurun_out/<tag>/<tag>_traffic.json, <tag>_c5_traffic.json (stamped with the
# kernel-source hash: bench.py only quotes roofline.traffic from a file whose stamp matches the code that runs)
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
Q="--steps 1 --warmup 1 --no_cpu_baseline --no_secondary --no_kernel_events"
for WL in c2 c5; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${WL}fetch -o f -- python bench.py --workload $WL $Q > /dev/null 2> $O/${WL}fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${WL}write -o w -- python bench.py --workload $WL $Q > /dev/null 2> $O/${WL}write.err
done
F=$(find $O/c2fetch -name "*counter_collection.csv" | head -1); W=$(find $O/c2write -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_traffic.py $F $W > $O/${TAG}_traffic.json 2> $O/pmc_traffic.err
F=$(find $O/c5fetch -name "*counter_collection.csv" | head -1); W=$(find $O/c5write -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_traffic.py $F $W > $O/${TAG}_c5_traffic.json 2> $O/c5_pmc_traffic.err
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
ls -la $O/*.json
