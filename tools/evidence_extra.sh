#!/bin/bash
# Secondary evidence of a round (run on the GPU box from the repo root):  bash tools/evidence_extra.sh <tag>
#   async / Atari-preset / env-scaling / sustained C2 lines, the c5 line with its rocprofv3 kernel stats, the layer-by-layer
#   parity probe.  Outputs under gpurun_out/<tag>/.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
Q="--no_cpu_baseline --no_secondary"
python bench.py $Q --steps 8 --async_rl > $O/${TAG}_bench_async.json 2> $O/async.err
python bench.py $Q --steps 2 --warmup 1 --rollout 128 --num_batches 16 --num_epochs 4 > $O/${TAG}_bench_atari_preset.json 2> $O/atari.err
python bench.py $Q --steps 600 --no_kernel_events > $O/${TAG}_bench_600steps.json 2> $O/long.err
for E in 8192 32768; do python bench.py $Q --steps 4 --envs $E --no_kernel_events > $O/${TAG}_envs_$E.json 2> $O/envs_$E.err; done
python bench.py --workload c5 $Q --steps 8 > $O/${TAG}_c5_bench.json 2> $O/c5.err
python bench.py --workload c5 --rnn_type gru $Q --steps 8 > $O/${TAG}_c5_gru_bench.json 2> $O/c5g.err
python bench.py --workload c3 $Q --steps 4 > $O/${TAG}_c3_bench.json 2> $O/c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5trace -o t -- python bench.py --workload c5 $Q --steps 6 > $O/c5_trace_bench.json 2> $O/c5trace.err
K=$(find $O/c5trace -name "*kernel_stats.csv" | head -1); [ -n "$K" ] && cp $K $O/${TAG}_c5_kernel_stats.csv
find $O -name "*kernel_trace.csv" -size +2M -delete
python tools/parity_probe.py > $O/${TAG}_parity_probe.log 2>&1; cp gpurun_out/parity_probe.json $O/${TAG}_parity_probe.json 2>/dev/null
for f in $O/${TAG}_bench_async.json $O/${TAG}_bench_atari_preset.json $O/${TAG}_bench_600steps.json $O/${TAG}_envs_8192.json $O/${TAG}_envs_32768.json $O/${TAG}_c5_bench.json $O/${TAG}_c5_gru_bench.json $O/${TAG}_c3_bench.json; do
  python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])
except Exception as e: print('$f', 'ERR', e)
"; done
du -sh $O
