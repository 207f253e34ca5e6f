#!/bin/bash
# Everything a round commits as judged evidence, in ONE GPU-box call (run from the repo root):
#   bash tools/evidence_round.sh <tag>        e.g. r03_d
#   1. the whole GPU test-suite;  2. (dropped: the driver-like line is step 5);  3. rocprofv3 kernel trace + stats of the timed
#   bench, the two TCC passes (HBM traffic) and one SQ pass (matrix-pipe busy) -- tools/profile_round.sh; counter passes never
#   share a run with other trace domains;  4. the post-processed summaries (stamped with the kernel-source hash);  5. the
#   bench line again, now with roofline.traffic from the pass just taken.
# Outputs under gpurun_out/<tag>/ ; the summaries a round keeps are copied into profiles/ by the builder.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
python -m pytest tests -q -m gpu --durations=30 2>&1 | tail -60 > $O/pytest.log
bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
F=$(find $O/fetch -name "*counter_collection.csv" | head -1)
W=$(find $O/write -name "*counter_collection.csv" | head -1)
S=$(find $O/sq -name "*counter_collection.csv" | head -1)
K=$(find $O/trace -name "*kernel_stats.csv" | head -1)
[ -n "$K" ] && cp $K $O/${TAG}_kernel_stats.csv
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_traffic.py $F $W > $O/${TAG}_traffic.json 2> $O/pmc_traffic.err
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_hbm.py $F $W $O/${TAG}_hbm_kernels.json > /dev/null 2> $O/pmc_hbm.err
[ -n "$S" ] && python tools/pmc_mfma.py $S $O/${TAG}_mfma_util.json > /dev/null 2> $O/pmc_mfma.err
cp $O/trace_bench.json $O/${TAG}_bench_under_rocprof.json 2>/dev/null
# the bench line with the traffic figure of the pass just taken (same kernel sources -> same hash)
if [ -s $O/${TAG}_traffic.json ]; then cp $O/${TAG}_traffic.json profiles/${TAG}_traffic.json; fi
python bench.py --steps 20 > $O/${TAG}_bench.json 2> $O/bench.err
# drop the bulky per-dispatch tables, keep stats + counter summaries
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
tail -3 $O/pytest.log
tail -c 400 $O/${TAG}_bench.json
du -sh $O
