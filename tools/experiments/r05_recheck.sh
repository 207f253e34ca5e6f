export KBENCH_LAYERS=conv2 KBENCH_NS=32768
for r in 1 2; do for v in 0 1; do echo -n "SF_TAP_PERM=$v "; SF_TAP_PERM=$v python tools/kbench.py fwd 2>/dev/null | grep conv2; done; done
