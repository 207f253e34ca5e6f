# round 6, call f: one-shot exchange tests (2 ranks on one device), then the side-stream weight gradients (SF_BWD_STREAMS=1) A/B
O=gpurun_out/r06f; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_dp.py -m gpu -x -q -k "one_shot or oneshot or ring_path" 2>&1 | tail -12 > $O/pytest_oneshot.log; cat $O/pytest_oneshot.log
L=$O/r06_f_bwd_streams.log; echo "## bench.py --steps 20 --warmup 5 --no_secondary --no_cpu_baseline, SF_BWD_STREAMS 0 / 1 alternating" > $L
for r in 1 2; do for v in 0 1; do
  SF_BWD_STREAMS=$v timeout 600 python bench.py --steps 20 --warmup 5 --no_secondary --no_cpu_baseline > $O/b_$v_$r.json 2> $O/b.err
  python -c "
import json;d=json.load(open('$O/b_$v_$r.json'));print('SF_BWD_STREAMS=$v', d['ms_per_step'], d['value'], d['roofline']['frac'])" >> $L
done; done
cat $L
timeout 600 python -m pytest tests/test_gpu_parity_c2_c5.py -m gpu -x -q -k "c2_geometry" 2>&1 | tail -3
SF_BWD_STREAMS=1 timeout 600 python -m pytest tests/test_gpu_parity_c2_c5.py -m gpu -x -q -k "c2_geometry" 2>&1 | tail -3 | tee -a $L
