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
# config 5: counter passes first (HBM traffic + matrix-pipe busy of the fused sequence kernels), so that the c5 line below
# carries roofline.traffic from kernel sources with the same hash
C1="python bench.py --workload c5 --steps 1 --warmup 1 $Q --no_kernel_events"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c5fetch -o f -- $C1 > /dev/null 2> $O/c5fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c5write -o w -- $C1 > /dev/null 2> $O/c5write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $O/c5sq -o p -- $C1 > /dev/null 2> $O/c5sq.err
F=$(find $O/c5fetch -name "*counter_collection.csv" | head -1); W=$(find $O/c5write -name "*counter_collection.csv" | head -1)
S=$(find $O/c5sq -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_traffic.py $F $W > $O/${TAG}_c5_traffic.json 2> $O/c5_pmc_traffic.err && cp $O/${TAG}_c5_traffic.json profiles/${TAG}_c5_traffic.json
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_hbm.py $F $W $O/${TAG}_c5_hbm_kernels.json > /dev/null 2> $O/c5_pmc_hbm.err
[ -n "$S" ] && python tools/pmc_mfma.py $S $O/${TAG}_c5_mfma_util.json > /dev/null 2> $O/c5_pmc_mfma.err
find $O -name "*counter_collection.csv" -size +8M -delete
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
