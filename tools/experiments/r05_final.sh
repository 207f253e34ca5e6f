# final stamp of the round: counter traffic on the final kernel sources + the bench line + the dispatch-sensitive tests
#   bash tools/experiments/r05_final.sh <tag>
T=${1:-r05_e}
O=gpurun_out/$T; mkdir -p $O
bash tools/restamp.sh $T > $O/restamp.log 2>&1
cp $O/${T}_traffic.json profiles/${T}_traffic.json; cp $O/${T}_c5_traffic.json profiles/${T}_c5_traffic.json
python bench.py --steps 20 > $O/${T}_bench.json 2> $O/bench.err
timeout 900 python -m pytest tests/test_gpu_headline_sizes.py tests/test_gpu_parity_c2_c5.py tests/test_gpu_nn.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest_dispatch.log
cat $O/pytest_dispatch.log; tail -c 300 $O/${T}_bench.json
