export KBENCH_LAYERS=conv3 KBENCH_NS=4096,4096,32768,32768
for r in 1 2 3; do for v in 1 3; do echo -n "SF_DGRAD_ZL=$v "; SF_DGRAD_ZL=$v python tools/kbench.py dgrad 2>/dev/null | grep 32768 | tail -1; done; done
SF_DGRAD_ZL=3 timeout 600 python -m pytest tests/test_gpu_nn.py -m gpu -q -x -k "dgrad or fuzz or large_grids" 2>&1 | tail -2
