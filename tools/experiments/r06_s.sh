#!/bin/bash
# native multi-key towers: the multi-key tests + launch programs + model tests
set -u
O=gpurun_out/${1:-r06_s_mk}; mkdir -p $O
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_nn.py tests/test_gpu_launch_programs.py -x -q -m gpu -k "multi or launch or replayed or recorder" 2>&1 | tail -40 > $O/r06_s_pytest_multikey.log
cat $O/r06_s_pytest_multikey.log
