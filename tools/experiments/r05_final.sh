# final stamp of the round: counter traffic on the final kernel sources + the bench line + the name-sensitive tests
O=gpurun_out/r05_c; mkdir -p $O
bash tools/restamp.sh r05_c > $O/restamp.log 2>&1
cp $O/r05_c_traffic.json profiles/r05_c_traffic.json; cp $O/r05_c_c5_traffic.json profiles/r05_c_c5_traffic.json
python bench.py --steps 20 > $O/r05_c_bench.json 2> $O/bench.err
timeout 900 python -m pytest tests/test_gpu_headline_sizes.py tests/test_gpu_parity_c2_c5.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest_names.log
cat $O/pytest_names.log; tail -c 600 $O/r05_c_bench.json
